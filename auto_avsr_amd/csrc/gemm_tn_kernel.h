// gemm_tn_kernel.h -- the k-major (TN) weight-gradient tile kernel (see gemm_tn_fast.hip for the design notes) as a
// device-side struct shared by the plain launch (gemm_tn_fast.hip) and the paired launch (gemm_pair.hip).
#pragma once
#include "gemm_core.h"

namespace avsr_tn {

using avsr_gemm_impl::Params;

// TMS x TNS (round 5): 64-column sub-tiles per operand -- the block computes a (64 TMS) x (64 TNS) output tile, each of its 2 x 2
// waves a (32 TMS) x (32 TNS) grid of 32 x 32 accumulators.  1 x 1 is the original kernel.  2 x 2 (128 x 128) halves the operand
// bytes per output and the transpose reads per MFMA (8 reads feed 4 MFMAs instead of 4 feeding 1): for the weight gradients with
// >= 400 64 x 64 tiles (FFN, fused Q / K / V), whose launches are bound by the L2 -> LDS operand stream.
// DEPTH: fragment register sets = k-steps whose transpose reads are in flight together (2: k-step ks+1 requested before the MFMAs
// of ks; 4: all four k-steps of a staged tile requested up front, 16 reads per wait chain -- the LDS reaches its rate only
// with >= 16 DS operations in flight per wave, MI355X_MICROARCH.md "LDS").
template <int STAGES, int CV, int TMS = 1, int TNS = 1, int DEPTH = 2>
struct TnKernel {
    static_assert(DEPTH == 2 || DEPTH == 4, "fragment pipeline depth");
    static_assert(CV == 0 || (TMS == 1 && TNS == 1), "gathered B operand: 64 x 64 tiles only");
    static constexpr int BM = 64 * TMS, BN = 64 * TNS, BK = 64;
    static constexpr int OP_BYTES = BK * 128;          // one 64-column sub-tile of an operand stage: 64 k-rows x 64 columns bf16
    static constexpr int A_BYTES = TMS * OP_BYTES, STAGE_BYTES = (TMS + TNS) * OP_BYTES;
    static constexpr int LPT = 2 * (TMS + TNS);        // LDS-DMA ops per thread per tile
    static constexpr size_t RING_BYTES = (size_t)STAGES * STAGE_BYTES, EPI_BYTES = (size_t)BM * (BN + 4) * 4;
    static constexpr size_t LDS_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;

    static AVSR_DEV void issue(const Params& p, const bf16_t* A, const bf16_t* B, int m0, int n0, int k0, int kend,
                               char* stage, int wave, int lane) {
        const int ksub = lane >> 3, chunk = (lane & 7) * 8;
        const bf16_t* zero = reinterpret_cast<const bf16_t*>(p.gate);
        int kh = 0, kw = 0, cbase = 0;
        if (CV == 3) {  // the 64-column tile lies inside one filter tap (Cin % 64 == 0): wave-uniform decode
            const int tap = n0 / p.cC;
            cbase = n0 - tap * p.cC;
            kh = tap / p.cKW;
            kw = tap - kh * p.cKW;
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int kr = (wave * 2 + i) * 8 + ksub;  // k-row inside the tile
            const int k = k0 + kr;
            const bool kin = k < kend;
#pragma unroll
            for (int s = 0; s < TMS; s++) {
                const int mc = m0 + s * 64 + chunk;
                const bf16_t* sa = (kin && mc < p.M) ? A + (size_t)k * p.lda + mc : zero;
                glds16(sa, stage + s * OP_BYTES + (wave * 2 + i) * 1024);
            }
#pragma unroll
            for (int s = 0; s < TNS; s++) {
                const bf16_t* sb = zero;
                if (CV == 0) {
                    const int nc = n0 + s * 64 + chunk;
                    if (kin && nc < p.N) sb = B + (size_t)k * p.ldb + nc;
                } else if (kin) {
                    const int pix = p.cOH * p.cOW;
                    const int n = k / pix, r = k - n * pix;
                    const int oh = r / p.cOW, ow = r - oh * p.cOW;
                    const int ih = oh * p.cS + kh - p.cPH, iw = ow * p.cS + kw - p.cPW;
                    if (ih >= 0 && ih < p.cH && iw >= 0 && iw < p.cW)
                        sb = B + (((size_t)n * p.cH + ih) * p.cW + iw) * p.cC + cbase + chunk;
                }
                glds16(sb, stage + A_BYTES + s * OP_BYTES + (wave * 2 + i) * 1024);
            }
        }
    }

    // 32 (m or n) x 16 (k) MFMA fragment of the k-major 64-column sub-tile at `base`: columns c0..c0+31, k-step ks, as two
    // transpose reads.  Issued in the asm form (prims.h lds_tr16_async): the compiler would otherwise park every
    // transpose read behind a vmcnt(0) -- i.e. behind the LDS-DMA of the NEXT tiles -- and serialise the ring.
    static AVSR_DEV void frag_async(const char* base, int c0, int ks, int lane, bf16x4& lo, bf16x4& hi) {
        const int g = lane >> 4, i = lane & 15;
        const bf16_t* t = reinterpret_cast<const bf16_t*>(base);
        const int row = ks * 16 + 8 * (g >> 1) + (i >> 2);
        const int col = c0 + 16 * (g & 1) + 4 * (i & 3);
        lo = lds_tr16_async(t + row * 64 + col);
        hi = lds_tr16_async(t + (row + 4) * 64 + col);
    }

    static AVSR_DEV void run(const Params& p, char* smem) { run_at(p, smem, blockIdx.x, blockIdx.y, blockIdx.z); }
    // block (bx, by, bz): n-tile, m-tile, k-split -- arguments so that a launch can host several problems
    static AVSR_DEV void run_at(const Params& p, char* smem, int bx, int by, int bz) {
        const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
        const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
        const int lane = threadIdx.x & 63, wave = wave_id();
        const int wm = wave >> 1, wn = wave & 1;
        const int m0 = by * BM, n0 = bx * BN;
        const int zs = bz;
        const int kbeg = zs * p.k_chunk;
        const int kend = min(p.K, kbeg + p.k_chunk);
        const int nt = (kend - kbeg + BK - 1) / BK;
        f32x16 acc[TMS][TNS];
#pragma unroll
        for (int i = 0; i < TMS; i++)
#pragma unroll
            for (int j = 0; j < TNS; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        // Bias gradient on the side: the blocks of the first n-tile column also sum the A tiles they stage over k (each wave
        // 16 of the 64 k-rows, lane = column).  A is the output gradient of the Linear, so this IS colsum(dY) -- no separate
        // pass over dY.  Read through a generic pointer: an LDS-address-space load would be ordered behind the LDS-DMA of the
        // tiles still in flight (prims.h).
        const bool do_cs = CV == 0 && p.colsum_a != nullptr && bx == 0;
        float cs[TMS];
#pragma unroll
        for (int s = 0; s < TMS; s++) cs[s] = 0.f;
#pragma unroll
        for (int s = 0; s < STAGES - 1; s++)
            if (s < nt) issue(p, A, B, m0, n0, kbeg + s * BK, kend, smem + s * STAGE_BYTES, wave, lane);
        // this wave's fragments: m block i covers tile columns wm * 32 TMS + 32 i (sub-tile = that / 64), n block j likewise
        constexpr int NF = 2 * (TMS + TNS);  // transpose reads per k-step
        for (int t = 0; t < nt; t++) {
            const int later = min(STAGES - 2, nt - 1 - t);
            if (later >= 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
            block_barrier_raw();
            if (t + STAGES - 1 < nt)
                issue(p, A, B, m0, n0, kbeg + (t + STAGES - 1) * BK, kend, smem + ((t + STAGES - 1) % STAGES) * STAGE_BYTES,
                      wave, lane);
            const char* As = smem + (t % STAGES) * STAGE_BYTES;
            const char* Bs = As + A_BYTES;
            if (do_cs) {
#pragma unroll
                for (int s = 0; s < TMS; s++) {
                    const bf16_t* col = reinterpret_cast<const bf16_t*>(As + s * OP_BYTES) + (wave * 16) * 64 + lane;
#pragma unroll
                    for (int r = 0; r < 16; r++) cs[s] += bf2f(col[r * 64]);
                }
            }
            bf16x4 fa[DEPTH][TMS][2], fb[DEPTH][TNS][2];  // DEPTH register sets (see the template comment)
            auto request = [&](int set, int ks) {
#pragma unroll
                for (int i = 0; i < TMS; i++) {
                    const int col = wm * 32 * TMS + 32 * i;
                    frag_async(As + (col >> 6) * OP_BYTES, col & 63, ks, lane, fa[set][i][0], fa[set][i][1]);
                }
#pragma unroll
                for (int j = 0; j < TNS; j++) {
                    const int col = wn * 32 * TNS + 32 * j;
                    frag_async(Bs + (col >> 6) * OP_BYTES, col & 63, ks, lane, fb[set][j][0], fb[set][j][1]);
                }
            };
            request(0, 0);
            if (DEPTH == 4) {
#pragma unroll
                for (int ks = 1; ks < BK / 16; ks++) request(ks, ks);
            }
#pragma unroll
            for (int ks = 0; ks < BK / 16; ks++) {
                const int c = DEPTH == 4 ? ks : (ks & 1);
                if (DEPTH == 4) {
                    switch (ks) {  // (immediates; the reads of the later k-steps stay in flight)
                        case 0: lds_wait<3 * NF>(); break;
                        case 1: lds_wait<2 * NF>(); break;
                        case 2: lds_wait<NF>(); break;
                        default: lds_wait<0>(); break;
                    }
                } else if (ks + 1 < BK / 16) {
                    request(c ^ 1, ks + 1);
                    lds_wait<NF>();  // the reads just issued may stay in flight
                } else {
                    lds_wait<0>();
                }
                bf16x8 a[TMS], b[TNS];
#pragma unroll
                for (int i = 0; i < TMS; i++) {
                    lds_tie(fa[c][i][0]);
                    lds_tie(fa[c][i][1]);
                    a[i] = bf16x8{fa[c][i][0][0], fa[c][i][0][1], fa[c][i][0][2], fa[c][i][0][3],
                                  fa[c][i][1][0], fa[c][i][1][1], fa[c][i][1][2], fa[c][i][1][3]};
                }
#pragma unroll
                for (int j = 0; j < TNS; j++) {
                    lds_tie(fb[c][j][0]);
                    lds_tie(fb[c][j][1]);
                    b[j] = bf16x8{fb[c][j][0][0], fb[c][j][0][1], fb[c][j][0][2], fb[c][j][0][3],
                                  fb[c][j][1][0], fb[c][j][1][1], fb[c][j][1][2], fb[c][j][1][3]};
                }
#pragma unroll
                for (int i = 0; i < TMS; i++)
#pragma unroll
                    for (int j = 0; j < TNS; j++) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
                sched_fence();
            }
        }
        if (do_cs) {
#pragma unroll
            for (int s = 0; s < TMS; s++)
                if (m0 + s * 64 + lane < p.M) atomicAdd(p.colsum_a + m0 + s * 64 + lane, cs[s]);
        }
        Params q = p;
        q.gate = nullptr;  // the field carries the zero page
        avsr_gemm_impl::epilogue_lds<BM, BN, TMS, TNS>(acc, q, m0, n0, wm * 32 * TMS, wn * 32 * TNS, zs, 0, smem);
    }
};

}  // namespace avsr_tn
