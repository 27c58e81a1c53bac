// gemm_tn_kernel.h -- the k-major (TN) weight-gradient tile kernel (see gemm_tn_fast.hip for the design notes) as a
// device-side struct shared by the plain launch (gemm_tn_fast.hip) and the paired launch (gemm_pair.hip).
#pragma once
#include "gemm_core.h"

namespace avsr_tn {

using avsr_gemm_impl::Params;

template <int STAGES, int CV>
struct TnKernel {
    static constexpr int BM = 64, BN = 64, BK = 64;
    static constexpr int OP_BYTES = BK * 128;          // one operand stage: 64 k-rows x 64 columns bf16
    static constexpr int STAGE_BYTES = 2 * OP_BYTES;
    static constexpr int LPT = 4;                      // LDS-DMA ops per thread per tile (2 for A, 2 for B)
    static constexpr size_t LDS_BYTES = (size_t)STAGES * STAGE_BYTES;

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
            const bf16_t* sa = (kin && m0 + chunk < p.M) ? A + (size_t)k * p.lda + m0 + chunk : zero;
            glds16(sa, stage + (wave * 2 + i) * 1024);
            const bf16_t* sb = zero;
            if (CV == 0) {
                if (kin && n0 + chunk < p.N) sb = B + (size_t)k * p.ldb + n0 + chunk;
            } else if (kin) {
                const int pix = p.cOH * p.cOW;
                const int n = k / pix, r = k - n * pix;
                const int oh = r / p.cOW, ow = r - oh * p.cOW;
                const int ih = oh * p.cS + kh - p.cPH, iw = ow * p.cS + kw - p.cPW;
                if (ih >= 0 && ih < p.cH && iw >= 0 && iw < p.cW)
                    sb = B + (((size_t)n * p.cH + ih) * p.cW + iw) * p.cC + cbase + chunk;
            }
            glds16(sb, stage + OP_BYTES + (wave * 2 + i) * 1024);
        }
    }

    // 32 (m or n) x 16 (k) MFMA fragment of the k-major tile at `base`: columns c0..c0+31, k-step ks, as two
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
        f32x16 acc[1][1];
#pragma unroll
        for (int r = 0; r < 16; r++) acc[0][0][r] = 0.f;
        // Bias gradient on the side: the blocks of the first n-tile column also sum the A tiles they stage over k (each wave
        // 16 of the 64 k-rows, lane = column).  A is the output gradient of the Linear, so this IS colsum(dY) -- no separate
        // pass over dY.  Read through a generic pointer: an LDS-address-space load would be ordered behind the LDS-DMA of the
        // tiles still in flight (prims.h).
        const bool do_cs = CV == 0 && p.colsum_a != nullptr && bx == 0;
        float cs = 0.f;
#pragma unroll
        for (int s = 0; s < STAGES - 1; s++)
            if (s < nt) issue(p, A, B, m0, n0, kbeg + s * BK, kend, smem + s * STAGE_BYTES, wave, lane);
        for (int t = 0; t < nt; t++) {
            const int later = min(STAGES - 2, nt - 1 - t);
            if (later >= 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
            block_barrier_raw();
            if (t + STAGES - 1 < nt)
                issue(p, A, B, m0, n0, kbeg + (t + STAGES - 1) * BK, kend, smem + ((t + STAGES - 1) % STAGES) * STAGE_BYTES,
                      wave, lane);
            const char* As = smem + (t % STAGES) * STAGE_BYTES;
            const char* Bs = As + OP_BYTES;
            if (do_cs) {
                const bf16_t* col = reinterpret_cast<const bf16_t*>(As) + (wave * 16) * 64 + lane;
#pragma unroll
                for (int r = 0; r < 16; r++) cs += bf2f(col[r * 64]);
            }
            bf16x4 f[2][4];  // two register sets: k-step ks+1 is requested before the MFMA of k-step ks
            frag_async(As, wm * 32, 0, lane, f[0][0], f[0][1]);
            frag_async(Bs, wn * 32, 0, lane, f[0][2], f[0][3]);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ks++) {
                const int c = ks & 1;
                if (ks + 1 < BK / 16) {
                    frag_async(As, wm * 32, ks + 1, lane, f[c ^ 1][0], f[c ^ 1][1]);
                    frag_async(Bs, wn * 32, ks + 1, lane, f[c ^ 1][2], f[c ^ 1][3]);
                    lds_wait<4>();  // the four reads just issued may stay in flight
                } else {
                    lds_wait<0>();
                }
#pragma unroll
                for (int q = 0; q < 4; q++) lds_tie(f[c][q]);
                const bf16x8 a{f[c][0][0], f[c][0][1], f[c][0][2], f[c][0][3], f[c][1][0], f[c][1][1], f[c][1][2], f[c][1][3]};
                const bf16x8 b{f[c][2][0], f[c][2][1], f[c][2][2], f[c][2][3], f[c][3][0], f[c][3][1], f[c][3][2], f[c][3][3]};
                acc[0][0] = mfma32(a, b, acc[0][0]);
                sched_fence();
            }
        }
        if (do_cs && m0 + lane < p.M) atomicAdd(p.colsum_a + m0 + lane, cs);
        Params q = p;
        q.gate = nullptr;  // the field carries the zero page
        avsr_gemm_impl::epilogue_lds<64, 64, 1, 1>(acc, q, m0, n0, wm * 32, wn * 32, zs, 0, smem);
    }
};

}  // namespace avsr_tn
