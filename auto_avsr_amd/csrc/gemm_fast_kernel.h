// gemm_fast_kernel.h -- the LDS-DMA NT tile kernel (see gemm_fast.hip for the design notes) as a device-side struct, so
// that more than one __global__ entry can run it: the plain launch (gemm_fast.hip) and the paired data-gradient +
// weight-gradient launch (gemm_pair.hip).
#pragma once
#include <type_traits>
#include "gemm_core.h"

namespace avsr_fast {

using avsr_gemm_impl::Params;

// CV = 0: plain A[M][K].  CV = 1 / 2: A is the im2col view of a channels-last image tensor (forward / data gradient,
// see gemm_core.h Params); channels are a multiple of 64, so a 64-wide k-tile lies inside one filter tap and the
// tap decode is wave-uniform (scalar); out-of-image taps read a caller-provided page of zeros.
//
// Gather addressing.  Every A row a lane stages is decoded ONCE per block into (pointer to tap (0,0) of that row,
// bit mask of the taps that fall inside the image); inside the k loop a load costs one 64-bit add of a scalar tap
// offset and a select against the zero page.  The data gradient of a strided convolution is split into the s*s
// residue classes of (ih+ph, iw+pw) mod s: a class only ever touches the taps kh = py (mod s), kw = px (mod s), so
// each class is a dense implicit GEMM over its own (shorter) tap list and no MFMA is spent on structural zeros.
//
// WGM x WGN waves per block (64*WGM*WGN threads), each wave a (BM/WGM) x (BN/WGN) grid of 32x32x16 accumulators.
// ABL (benchmarks only): 1 = no LDS reads / MFMA, 2 = no operand loads in the steady state.
// KS > 1: KS groups of WGM x WGN waves work on the SAME output tile, group g multiplying k-steps [g*4/KS, (g+1)*4/KS) of
// every staged 64-wide k-tile; the partial tiles meet in LDS before the epilogue.  A skinny M = B*T GEMM gives every CU
// about one block: with 4 waves that is ONE wave per SIMD and nothing to cover the LDS-read -> MFMA latency of the
// dependent accumulator chain; 8 waves put two on every SIMD without a second pass over the output.
// F16 = 1: both operands are IEEE half instead of bf16 (the forward pass of the "mixed" numerical mode) -- the same bytes, the same
// staging, the same fragment layout; only the MFMA instruction differs.
// WP = 2 (round 5, F16 only): the B operand (a weight) arrives as TWO f16 planes -- hi = f16(w) and lo = f16((w - hi) * 2^11),
// Params::B and Params::B2, same pitch -- and every product is formed twice: acc += a * hi, acc_lo += a * lo, result =
// acc + acc_lo * 2^-11.  The weight is then exact to ~2^-22 and the only operand rounding left is the activation's 2^-12:
// the parity study (tools/precision_study.py, "f16a") puts the decoder-logit error of the mixed mode at 0.65x of the
// single-plane figure.  The lo plane is SCALED so that it stays in the normal f16 range (|w - hi| <= 2^-12 |w| would be
// subnormal for every weight below 0.25) and needs its own accumulators for that.  Cost: the B stream and the MFMA count double,
// the A stream does not.
template <int BM, int BN, int STAGES, int CV = 0, int WGM = 2, int WGN = 2, int ABL = 0, int KS = 1, int F16 = 0, int WP = 1>
struct FastKernel {
    static_assert(WP == 1 || (WP == 2 && F16 == 1 && KS == 1), "two weight planes: f16 operands only");
    static constexpr int BK = 64, NQ = WGM * WGN, NW = NQ * KS, NTHR = 64 * NW;
    static constexpr int KPG = (BK / 16) / KS;  // 16-wide k-steps per wave group and k-tile
    static_assert(KS == 1 || KS == 2 || KS == 4, "k-split of the 64-wide tile");
    static constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + WP * B_BYTES;
    static constexpr int A_LOADS = BM / (8 * NW), B_LOADS = BN / (8 * NW);  // wave-instructions per wave per stage (and plane)
    static constexpr int LPT = A_LOADS + WP * B_LOADS;                      // LDS-DMA ops per thread per tile
    static constexpr size_t RING_BYTES = (size_t)STAGES * STAGE_BYTES, EPI_BYTES = (size_t)BM * (BN + 4) * 4;
    static constexpr size_t MAP_OFF = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
    static constexpr size_t LDS_BYTES = MAP_OFF + (CV == 2 ? BM * 4 : 0);
    static_assert(A_LOADS >= 1 && B_LOADS >= 1 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile / wave-count mismatch");
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");

    struct Rows {
        const bf16_t* a[A_LOADS];  // CV 0: &A[row][8c] ; CV 1/2: &src[pixel of tap (0,0)][8c] (may lie outside the tensor)
        uint32_t mask[A_LOADS];    // CV 1/2: bit t set <=> tap t of the block's tap list reads inside the image
        const bf16_t* b[B_LOADS];  // &B[row][8c]
    };
    // wave-uniform description of the tap list this block walks
    struct Taps {
        int nkw, ntaps, cpt;  // taps per filter row, tap count, 64-wide k-tiles per tap
        int py, px;           // CV 2: residue class
    };

    // tile row r (of this block) -> (image n, grid coordinates a, b) ; false when the row is beyond the class
    static AVSR_DEV bool row_coords(const Params& p, int cls, int mloc, int& n, int& a, int& b) {
        const int hc = p.cls_h[cls], wc = p.cls_w[cls];
        const long mc = (long)p.cN * hc * wc;
        const bool ok = mloc < mc;
        const int m = ok ? mloc : (int)(mc - 1);
        const int pix = hc * wc;
        n = m / pix;
        const int r = m - n * pix;
        a = r / wc;
        b = r - a * wc;
        return ok;
    }

    static AVSR_DEV Rows decode_rows(const Params& p, const bf16_t* A, const bf16_t* B, int m0, int n0, int cls,
                                     const Taps& tp, int wave, int lane) {
        Rows ri;
        const int rsub = lane >> 3, pc = lane & 7;
#pragma unroll
        for (int i = 0; i < A_LOADS; i++) {
            const int r = (wave * A_LOADS + i) * 8 + rsub;  // row inside the tile
            const int c = pc ^ ((r >> 1) & 7);              // source chunk that lands in physical chunk pc
            if (CV == 0) {
                const int gr = min(m0 + r, p.M - 1);
                ri.a[i] = A + (size_t)gr * p.lda + c * 8;
                ri.mask[i] = 0;
            } else {
                int n, a, b;
                row_coords(p, cls, m0 + r, n, a, b);
                int ya, xa;
                if (CV == 1) {
                    ya = a * p.cS - p.cPH;
                    xa = b * p.cS - p.cPW;
                } else {
                    ya = (p.cls_y0[cls] + a * p.cS + p.cPH) / p.cS;
                    xa = (p.cls_x0[cls] + b * p.cS + p.cPW) / p.cS;
                }
                ri.a[i] = A + ((long)n * p.cH * p.cW + (long)ya * p.cW + xa) * p.cC + c * 8;
                uint32_t mk = 0;
                for (int t = 0; t < tp.ntaps; t++) {
                    const int ta = t / tp.nkw, tb = t - ta * tp.nkw;
                    const int y = CV == 1 ? ya + ta : ya - ta, x = CV == 1 ? xa + tb : xa - tb;
                    if (y >= 0 && y < p.cH && x >= 0 && x < p.cW) mk |= 1u << t;
                }
                ri.mask[i] = mk;
            }
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; i++) {
            const int r = (wave * B_LOADS + i) * 8 + rsub;
            const int c = pc ^ ((r >> 1) & 7);
            const int gr = min(n0 + r, p.N - 1);
            ri.b[i] = B + (size_t)gr * p.ldb + c * 8;
        }
        return ri;
    }

    // stage k-tile t of this block
    static AVSR_DEV void issue(const Params& p, const Rows& ri, const Taps& tp, int kbeg, int t, char* stage, int wave,
                               int rot = 0, int nt = 1) {
        long da, db;
        int tap = 0;
        if (CV == 0) {
            int tt = t + rot;
            if (tt >= nt) tt -= nt;
            da = db = kbeg + tt * BK;
        } else {
            tap = t / tp.cpt;
            const int cb = (t - tap * tp.cpt) * BK;
            const int ta = tap / tp.nkw, tb = tap - ta * tp.nkw;
            if (CV == 1) {
                da = (long)(ta * p.cW + tb) * p.cC + cb;
                db = (long)tap * p.cC + cb;
            } else {
                da = -(long)(ta * p.cW + tb) * p.cC + cb;
                db = (long)((tp.py + ta * p.cS) * p.cKW + tp.px + tb * p.cS) * p.cC + cb;
            }
        }
#pragma unroll
        for (int i = 0; i < A_LOADS; i++) {
            const bf16_t* src = ri.a[i] + da;
            if (CV != 0) src = ((ri.mask[i] >> tap) & 1u) ? src : reinterpret_cast<const bf16_t*>(p.gate);  // zero page
            glds16(src, stage + (wave * A_LOADS + i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; i++) glds16(ri.b[i] + db, stage + A_BYTES + (wave * B_LOADS + i) * 1024);
        if (WP == 2) {
            const long lo = reinterpret_cast<const bf16_t*>(p.B2) - reinterpret_cast<const bf16_t*>(p.B);  // (elements; same pitch)
#pragma unroll
            for (int i = 0; i < B_LOADS; i++) glds16(ri.b[i] + db + lo, stage + A_BYTES + B_BYTES + (wave * B_LOADS + i) * 1024);
        }
    }

    static AVSR_DEV bf16x8 frag(const char* base, int r, int chunk) {
        return *reinterpret_cast<const bf16x8*>(base + r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4));
    }

    static AVSR_DEV void run(const Params& p, char* smem) {
        run_at(p, smem, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y, gridDim.z);
    }
    // the block (bx, by, bz) of a (gx, gy, gz) grid -- coordinates are arguments so that a launch can host several problems
    static AVSR_DEV void run_at(const Params& p, char* smem, int bx, int by, int bz, int gx_, int gy_, int gz_) {
        const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
        const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
        const int lane = threadIdx.x & 63, wave = wave_id();
        const int kg = wave / NQ, wq = wave - kg * NQ;  // k group, wave inside the group
        const int wm = wq / WGN, wn = wq % WGN;
        if (p.xcd_order && gz_ == 1) {
            // block b runs on XCD b % 8 (observed): hand every XCD one contiguous run of tiles so that tiles sharing
            // A rows (the n-tiles of an m-tile, the halo rows of neighbouring m-tiles) meet in the same L2
            const int gx = gx_, total = gx * gy_;
            const int id = by * gx + bx;
            const int xcd = id & 7, slot = id >> 3, q = total >> 3, r = total & 7;
            const int nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
            by = nid / gx;
            bx = nid - by * gx;
        }
        const int n0 = bx * BN;
        const int zs = bz;

        // which tap list / row range does this block work on
        int cls = 0, m0 = by * BM, nt, kbeg = 0;
        Taps tp{1, 1, 1, 0, 0};
        if (CV == 0) {
            kbeg = zs * p.k_chunk;
            nt = (min(p.K, kbeg + p.k_chunk) - kbeg) / BK;
        } else {
            while (cls + 1 < p.ncls && by >= p.cls_tile0[cls + 1]) cls++;
            m0 = (by - p.cls_tile0[cls]) * BM;
            tp.nkw = p.cls_nkw[cls];
            tp.ntaps = p.cls_nkh[cls] * tp.nkw;
            tp.cpt = p.cC / BK;
            tp.py = p.cls_py[cls];
            tp.px = p.cls_px[cls];
            nt = tp.ntaps * tp.cpt;
        }
        int* rowmap = nullptr;
        if (CV == 2) {  // output pixel of every tile row (classes interleave in memory)
            rowmap = reinterpret_cast<int*>(smem + MAP_OFF);
            for (int r = threadIdx.x; r < BM; r += NTHR) {
                int n, a, b;
                const bool ok = row_coords(p, cls, m0 + r, n, a, b);
                rowmap[r] = ok ? (n * p.cOH + p.cls_y0[cls] + a * p.cS) * p.cOW + p.cls_x0[cls] + b * p.cS : -1;
            }
        }

        f32x16 acc[TM][TN];
        f32x16 acc_lo[WP == 2 ? TM : 1][WP == 2 ? TN : 1];  // products with the scaled lo plane of the weight
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    acc[i][j][r] = 0.f;
                    if (WP == 2) acc_lo[i][j][r] = 0.f;
                }

        const Rows ri = decode_rows(p, A, B, m0, n0, cls, tp, wave, lane);
        // blocks walk k in lockstep: with a row pitch that is a multiple of the channel interleave, every block's loads of
        // one k-tile land on the same few L2 channels.  A per-block rotation of the k order spreads them.
        const int rot = (CV == 0 && p.k_rot) ? (int)(((unsigned)(by * gx_ + bx) * (unsigned)p.k_rot) % (unsigned)nt) : 0;
        // prologue: tiles 0 .. STAGES-2 in flight
#pragma unroll
        for (int s = 0; s < STAGES - 1; s++)
            if (s < nt) issue(p, ri, tp, kbeg, s, smem + s * STAGE_BYTES, wave, rot, nt);

        // One k-tile: retire its loads, barrier, first fragments, (optionally) stage tile t+STAGES-1, multiply.
        // ISSUE is a compile-time flag -- the steady state (every iteration stages a tile) and the drain (none does)
        // are separate loops, so neither carries a branch between the fragment reads and the MFMAs that use them.
        auto step = [&](int t, auto issue_flag) {
            constexpr bool ISSUE = decltype(issue_flag)::value;
            // loads of at most STAGES-2 later tiles may stay in flight
            if (ISSUE) {
                wait_vmcnt<(STAGES - 2) * LPT>();
            } else {
                switch (nt - 1 - t) {  // wave-uniform; counts are immediates
                    case 0: wait_vmcnt<0>(); break;
                    default: wait_vmcnt<(STAGES > 2 ? LPT : 0)>(); break;
                }
            }
            block_barrier_raw();  // tile t is in LDS for every wave; everyone is done reading tile t-1's buffer
            const char* As = smem + (t % STAGES) * STAGE_BYTES;
            const char* Bs = As + A_BYTES;
            // Two fragment register sets.  The fragments of k-steps 0 and 1 are requested right after the barrier, ahead
            // of the address arithmetic of the next operand loads (which hides their LDS latency); k-step ks+2 is
            // requested as soon as the MFMAs of k-step ks have been issued.  The scheduling fences keep that order.
            // (The compiler makes the first LDS wait after an LDS-DMA instruction a full one; placed here it is free.)
            bf16x8 fa[2][TM], fb[2][TN], fl[2][WP == 2 ? TN : 1];
            const int arow = wm * WM + (lane & 31), brow = wn * WN + (lane & 31);
            auto load_frags = [&](int set, int ks) {
                const int chunk = ks * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < TM; i++) fa[set][i] = frag(As, arow + i * 32, chunk);
#pragma unroll
                for (int j = 0; j < TN; j++) fb[set][j] = frag(Bs, brow + j * 32, chunk);
                if (WP == 2) {
#pragma unroll
                    for (int j = 0; j < TN; j++) fl[set][j] = frag(Bs + B_BYTES, brow + j * 32, chunk);
                }
            };
            const int ks0 = kg * KPG;
            if (ABL != 1) {
                load_frags(0, ks0);
                if (KPG > 1) load_frags(1, ks0 + 1);
            }
            sched_fence();
            if (ISSUE && ABL != 2)
                issue(p, ri, tp, kbeg, t + STAGES - 1, smem + ((t + STAGES - 1) % STAGES) * STAGE_BYTES, wave, rot, nt);
            sched_fence();
            if (ABL == 1) return;
#pragma unroll
            for (int ks = 0; ks < KPG; ks++) {
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        acc[i][j] = mfma32x<F16>(fa[ks & 1][i], fb[ks & 1][j], acc[i][j]);
                        if (WP == 2) acc_lo[i][j] = mfma32x<F16>(fa[ks & 1][i], fl[ks & 1][j], acc_lo[i][j]);
                    }
                if (ks + 2 < KPG) load_frags(ks & 1, ks0 + ks + 2);
                sched_fence();
            }
        };
        int t = 0;
        for (; t + STAGES - 1 < nt; t++) step(t, std::true_type{});
        for (; t < nt; t++) step(t, std::false_type{});
        if (WP == 2) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[i][j][r] += acc_lo[i][j][r] * (1.0f / AVSR_H16_LO_SCALE);
        }
        if (CV != 0) {
            Params q = p;
            q.gate = nullptr;  // in conv mode the field carries the zero page, not an activation gate
            avsr_gemm_impl::epilogue_lds<BM, BN, TM, TN, NTHR, KS>(acc, q, m0, n0, wm * WM, wn * WN, zs, 0, smem, rowmap, kg);
        } else {
            avsr_gemm_impl::epilogue_lds<BM, BN, TM, TN, NTHR, KS>(acc, p, m0, n0, wm * WM, wn * WN, zs, 0, smem, nullptr, kg);
        }
    }
};

}  // namespace avsr_fast
