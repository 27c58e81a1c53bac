// gemm_core.h -- LDS-tiled MFMA GEMM for gfx950 (v_mfma_f32_32x32x16_bf16), templated on
// operand storage, operand memory layout and tile shape.
//
//   C[M,N] = epilogue( sum_k A[m,k] * B[n,k] )
//
// Operand layouts (what is contiguous in HBM):
//   LA = 0: A stored [M][K]  (k contiguous; activations x, dY in dgrad)
//   LA = 1: A stored [K][M]  (m contiguous; dY^T in wgrad)
//   LB = 0: B stored [N][K]  (k contiguous; torch Linear weight [out,in] in forward)
//   LB = 1: B stored [K][N]  (n contiguous; the same weight used for dgrad, x in wgrad)
//
// This one kernel therefore covers every dense contraction of the hot path:
// Linear forward (LA0,LB0), its data gradient (LA0,LB1) and weight gradient
// (LA1,LB1) -- the k=1 Conv1d "pointwise" layers of ConvolutionModule included
// (conformer_encoder.py:24,27), since (B,T,D) activations make them plain GEMMs.
//
// Structure: 256 threads = 4 waves in a 2x2 grid, each wave owns (BM/2)x(BN/2)
// of the tile as 32x32 MFMA accumulators; operands are staged HBM -> registers
// -> LDS ([rows][BK+8] bf16, 144-byte pitch: conflict-free ds_read_b128) with
// the next tile's global loads in flight while the current one is multiplied.
// NS = 2 keeps a hi and a lo bf16 plane per operand (f32-class accuracy).
#pragma once
#include <algorithm>
#include "prims.h"

namespace avsr_gemm_impl {

struct Params {
    const void* A;
    const void* B;
    const void* B2;  // tuned NT kernel with two f16 weight planes (gemm_fast_kernel.h WP = 2): the scaled lo plane, pitch ldb; else unused
    int lda, ldb;
    int M, N, K;
    int k_chunk;  // K range handled per blockIdx.z (split-K); == K when not split
    // epilogue, applied in this order:
    const float* bias;  // [N] or null
    int act;            // 0 none, 1 relu, 2 silu
    const void* gate;   // saved activation [M,ldg]: v = gate>0 ? v*gate_scale : 0 (ReLU/dropout backward)
    int gate_dtype, ldg;
    float gate_scale;
    float drop_p;  // dropout on v (forward); keep-mask from (seed, m*N+n)
    uint64_t seed;
    float alpha;         // v *= alpha (* *alpha_dev when given: device-side scalar, e.g. an upstream loss gradient)
    const float* alpha_dev;
    const uint64_t* seed_dev;  // added to seed when given (graph-replay safe dropout)
    const float* resid;  // v += resid[m,ldr]
    int ldr;
    void* C;
    void* C2;  // LDS epilogue, non-accumulating f32 / f16 outputs: bf16 twin of the stored values (same shape, pitch ldc2) or null
    int ldc2;
    int c_dtype, ldc;
    int accumulate;  // f32 C only: atomicAdd (needed for split-K; also "+=" semantics)
    // batching over (b, h): blockIdx.z = (b*batch_h + h)*nsplit + ksplit; strides in elements
    int nsplit, batch_h, nbatch;
    long sAb, sAh, sBb, sBh, sCb, sCh;
    // skewed A (LA = 1 only): A[m][k] = src[k*lda + m + k - skew_off], valid iff 0 <= m+k-skew_off < skew_lim.
    // This is the transpose of the reference's rel_shift (attention.py:131-151) applied to dS, so that the
    // gradient of the projected positions is an ordinary TN contraction.
    int a_skew, skew_off, skew_lim;
    int resid_dtype;  // 0: resid is f32 (default), 1: bf16
    // implicit-GEMM convolution on channels-last tensors (template parameter CV selects the gather):
    //   CV 1: A[m][k] = x[n, oh*s+kh-ph, ow*s+kw-pw, ci]          m=(n,oh,ow) k=(kh,kw,ci)   forward
    //   CV 2: A[m][k] = dy[n, (ih+ph-kh)/s, (iw+pw-kw)/s, co]     m=(n,ih,iw) k=(kh,kw,co)   data gradient
    //   CV 3: B[k][n'] = x[...] as CV 1 with k=(n,oh,ow) n'=(kh,kw,ci)                         weight gradient
    //   CV 4/5: CV 1/3 for a single input channel with temporal taps: k=(kt,kh,kw), clip length cT
    int cH, cW, cC;       // spatial size / channels of the gathered tensor
    int cOH, cOW;         // pixel grid the row index runs over
    int cKH, cKW, cS, cPH, cPW;
    int cT, cKT, cPT;
    // tuned bf16 kernel (gemm_fast.hip) only: image count, tile order, and the residue classes a strided data
    // gradient is split into (row grid cls_h x cls_w starting at pixel (cls_y0, cls_x0) with step cS; taps
    // kh = cls_py + cS*i, kw = cls_px + cS*j; m-tiles [cls_tile0[c], cls_tile0[c+1]))
    float* colsum;  // LDS epilogue, non-accumulating outputs: colsum[n] += sum_m of the stored value (bias gradients)
    // LDS epilogue, non-accumulating outputs: BatchNorm statistics of the stored values without a pass over them --
    // colstat [M tiles][2][N] (need not be initialised): row (m0 / BM) receives the tile's column sums and sums of squares by plain
    // stores (a first version accumulated into 64 slots with float atomics: 774 k L2 atomics per first-stage convolution cost
    // what the removed statistics pass had cost)
    float* colstat;
    int colstat_rows;  // rows the caller allocated (128-row tiles); kernels with taller tiles zero the rows they do not fill
    float* colsum_a;  // TN tile kernel only: colsum_a[m] += sum_k A[k][m] -- the bias gradient of the Linear whose weight
                      // gradient this contraction is (A = its output gradient), taken from the staged A tiles
    int cN, xcd_order, ncls;
    int k_rot;  // tuned NT kernel, plain GEMM: block b starts its k loop at tile (b * k_rot) % nt (0 = every block at k = 0)
    int cls_tile0[5], cls_py[4], cls_px[4], cls_y0[4], cls_x0[4], cls_h[4], cls_w[4], cls_nkh[4], cls_nkw[4];
};

template <class T> struct Raw8;
template <> struct Raw8<bf16_t> {
    bf16x8 v;
    AVSR_DEV void zero() { v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
    AVSR_DEV void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
    AVSR_DEV void set(int e, const bf16_t* p) { v[e] = (short)*p; }
    AVSR_DEV float get(int e) const { return bf2f((bf16_t)v[e]); }
};
template <> struct Raw8<float> {
    f32x4 a, b;
    AVSR_DEV void zero() { a = f32x4{0, 0, 0, 0}; b = a; }
    AVSR_DEV void load(const float* p) {
        a = *reinterpret_cast<const f32x4*>(p);
        b = *reinterpret_cast<const f32x4*>(p + 4);
    }
    AVSR_DEV void set(int e, const float* p) {
        if (e < 4) a[e] = *p; else b[e - 4] = *p;
    }
    AVSR_DEV float get(int e) const { return e < 4 ? a[e] : b[e - 4]; }
};

// 8 consecutive elements starting at (r, c) of a row-major matrix, zero outside [r_lim) x [c_lim)
template <class T>
AVSR_DEV Raw8<T> load_chunk(const T* base, int ld, int r, int c, int r_lim, int c_lim) {
    Raw8<T> out;
    if (r < r_lim && c + 8 <= c_lim) {
        out.load(base + (size_t)r * ld + c);
    } else {
        out.zero();
        if (r < r_lim) {
            for (int e = 0; e < 8; e++)
                if (c + e < c_lim) out.set(e, base + (size_t)r * ld + c + e);
        }
    }
    return out;
}

// skewed variant: element e of the chunk is src[r*ld + c + e + r - off] when m = c+e < m_lim and the shifted
// column lies in [0, lim)
template <class T>
AVSR_DEV Raw8<T> load_chunk_skew(const T* base, int ld, int r, int c, int r_lim, int m_lim, int off, int lim) {
    Raw8<T> out;
    out.zero();
    if (r < r_lim) {
        for (int e = 0; e < 8; e++) {
            const int j = c + e + r - off;
            if (c + e < m_lim && j >= 0 && j < lim) out.set(e, base + (size_t)r * ld + j);
        }
    }
    return out;
}

// Shared epilogue of the MFMA GEMM kernels: acc[i][j] is the 32x32 accumulator whose top-left element is
// (row0 + 32 i, col0 + 32 j); order: +bias -> act -> gate -> dropout -> *alpha -> +resid -> store / atomicAdd.
template <int TM, int TN>
AVSR_DEV void epilogue(f32x16 (&acc)[TM][TN], const Params& p, int row0, int col0, int lane, int zs, long c_off) {
    const float inv_keep = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    const float alpha = p.alpha * (p.alpha_dev ? *p.alpha_dev : 1.f);
    const uint64_t seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int col = col0 + j * 32 + (lane & 31);
            if (col >= p.N) continue;
            const float bias = (p.bias && zs == 0) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bias;
                if (p.act == 1) v = fmaxf(v, 0.f);
                else if (p.act == 2) v = avsr_silu(v);
                if (p.gate) {
                    const float g = p.gate_dtype == 0
                                        ? reinterpret_cast<const float*>(p.gate)[(size_t)row * p.ldg + col]
                                        : bf2f(reinterpret_cast<const bf16_t*>(p.gate)[(size_t)row * p.ldg + col]);
                    v = g > 0.f ? v * p.gate_scale : 0.f;
                }
                if (p.drop_p > 0.f)
                    v *= dropout_scale(seed, (uint64_t)row * (uint64_t)p.N + col, p.drop_p, inv_keep);
                v *= alpha;
                if (p.resid && zs == 0)
                    v += p.resid_dtype == 0 ? p.resid[(size_t)row * p.ldr + col]
                                            : bf2f(reinterpret_cast<const bf16_t*>(p.resid)[(size_t)row * p.ldr + col]);
                if (p.c_dtype == 0) {
                    float* c = reinterpret_cast<float*>(p.C) + c_off + (size_t)row * p.ldc + col;
                    if (p.accumulate) atomicAdd(c, v); else *c = v;
                } else {
                    reinterpret_cast<bf16_t*>(p.C)[c_off + (size_t)row * p.ldc + col] = f2bf(v);
                }
            }
        }
}

// Epilogue through LDS: the accumulators (MFMA C layout: a lane owns ONE column and 16 scattered rows per 32x32
// tile) are parked in LDS as an f32 [BM][BN+4] image, then every thread finishes 8 CONSECUTIVE columns of one row at
// a time -- vector loads of bias / gate / residual, one 16-byte (bf16) or two 16-byte (f32) coalesced stores.
// Measured on the short-K problems of this model (K = 576..768), the per-element epilogue above was ~40 % of the
// kernel time; `smem` is the (now idle) operand staging area and must hold BM*(BN+4) floats.
// rowmap (LDS, may be null): output row of every tile row, -1 = none; default is row m0 + r.
// KS > 1 (kgroup = this wave's group): KS wave groups hold partial sums of the same tile; group 0 stores, the others add
// into the LDS tile one after the other.
template <int BM, int BN, int TM, int TN, int NTHR = 256, int KS = 1>
AVSR_DEV void epilogue_lds(f32x16 (&acc)[TM][TN], const Params& p, int m0, int n0, int wrow, int wcol, int zs, long c_off,
                           char* smem, const int* rowmap = nullptr, int kgroup = 0) {
    constexpr int PITCH = BN + 4;
    float* tile = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x & 63;
    __syncthreads();  // every wave is done reading operands from LDS
#pragma unroll
    for (int g = 0; g < KS; g++) {
        if (g > 0) __syncthreads();
        if (kgroup != g) continue;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float* dst = tile + (wrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * PITCH + wcol + j * 32 + (lane & 31);
                    *dst = g == 0 ? acc[i][j][r] : *dst + acc[i][j][r];
                }
    }
    __syncthreads();
    const float inv_keep = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    const float alpha = p.alpha * (p.alpha_dev ? *p.alpha_dev : 1.f);
    const uint64_t seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
    const bool lead = zs == 0;
    if (p.accumulate) {
        // split-K / "+=" outputs: one f32 atomic per element with CONSECUTIVE LANES ON CONSECUTIVE COLUMNS, so that a
        // wave's atomic instruction covers whole cache lines (8 columns per lane would scatter it over 16 lines)
        for (int idx = threadIdx.x; idx < BM * BN; idx += NTHR) {
            const int r = idx / BN, c = idx - r * BN;
            const int row = rowmap ? rowmap[r] : m0 + r, col = n0 + c;
            if (row < 0 || row >= p.M || col >= p.N) continue;
            float v = tile[r * PITCH + c];
            if (p.bias && lead) v += p.bias[col];
            v *= alpha;
            atomicAdd(reinterpret_cast<float*>(p.C) + c_off + (size_t)row * p.ldc + col, v);
        }
        return;
    }
    constexpr int CPR = BN / 8;  // chunks per row
    static_assert(NTHR % CPR == 0 && 64 % CPR == 0, "a thread keeps one column chunk for all of its rows");
    float cs[8], cq[8];  // this thread's share of the column sums / sums of squares (its chunk, its rows)
#pragma unroll
    for (int e = 0; e < 8; e++) cs[e] = cq[e] = 0.f;
    for (int id = threadIdx.x; id < BM * CPR; id += NTHR) {
        const int r = id / CPR, c = (id % CPR) * 8;
        const int row = rowmap ? rowmap[r] : m0 + r, col = n0 + c;
        if (row < 0 || row >= p.M || col >= p.N) continue;
        float v[8];
        {
            const f32x4 a = *reinterpret_cast<const f32x4*>(tile + r * PITCH + c);
            const f32x4 b = *reinterpret_cast<const f32x4*>(tile + r * PITCH + c + 4);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v[e] = a[e];
                v[e + 4] = b[e];
            }
        }
        const bool full = col + 8 <= p.N;
        const int nv = full ? 8 : p.N - col;
        if (p.bias && lead) {
            if (full) {
                float bb[8];
                load8(p.bias + col, bb);
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] += bb[e];
            } else {
                for (int e = 0; e < nv; e++) v[e] += p.bias[col + e];
            }
        }
        if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == 2) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = avsr_silu(v[e]);
        }
        if (p.gate) {
            float g[8];
            const size_t go = (size_t)row * p.ldg + col;
            if (full && p.ldg % 8 == 0) {
                if (p.gate_dtype == 0) load8(reinterpret_cast<const float*>(p.gate) + go, g);
                else load8(reinterpret_cast<const bf16_t*>(p.gate) + go, g);
            } else {
                for (int e = 0; e < 8; e++)
                    g[e] = e < nv ? (p.gate_dtype == 0 ? reinterpret_cast<const float*>(p.gate)[go + e]
                                                       : bf2f(reinterpret_cast<const bf16_t*>(p.gate)[go + e]))
                                  : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = g[e] > 0.f ? v[e] * p.gate_scale : 0.f;
        }
        if (p.drop_p > 0.f) {
#pragma unroll
            for (int e = 0; e < 8; e++)
                v[e] *= dropout_scale(seed, (uint64_t)row * (uint64_t)p.N + col + e, p.drop_p, inv_keep);
        }
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] *= alpha;
        if (p.resid && lead) {
            float rr[8];
            const size_t ro = (size_t)row * p.ldr + col;
            if (full && p.ldr % 8 == 0) {
                if (p.resid_dtype == 0) load8(p.resid + ro, rr);
                else load8(reinterpret_cast<const bf16_t*>(p.resid) + ro, rr);
            } else {
                for (int e = 0; e < 8; e++)
                    rr[e] = e < nv ? (p.resid_dtype == 0 ? p.resid[ro + e] : bf2f(reinterpret_cast<const bf16_t*>(p.resid)[ro + e]))
                                   : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] += rr[e];
        }
        if (p.colsum || p.colstat) {
#pragma unroll
            for (int e = 0; e < 8; e++) cs[e] += e < nv ? v[e] : 0.f;
        }
        if (p.colstat) {
#pragma unroll
            for (int e = 0; e < 8; e++) cq[e] += e < nv ? v[e] * v[e] : 0.f;
        }
        const size_t co = c_off + (size_t)row * p.ldc + col;
        if (p.C2) {
            bf16_t* c2 = reinterpret_cast<bf16_t*>(p.C2) + c_off + (size_t)row * p.ldc2 + col;
            if (full && p.ldc2 % 8 == 0 && (c_off % 8) == 0) {
                store8(c2, v);
            } else {
                for (int e = 0; e < nv; e++) c2[e] = f2bf(v[e]);
            }
        }
        if (p.c_dtype == 0) {
            float* cp = reinterpret_cast<float*>(p.C) + co;
            if (p.accumulate) {
                for (int e = 0; e < nv; e++) atomicAdd(cp + e, v[e]);
            } else if (full && p.ldc % 4 == 0 && (c_off % 4) == 0) {
                store8(cp, v);
            } else {
                for (int e = 0; e < nv; e++) cp[e] = v[e];
            }
        } else if (p.c_dtype == 2) {  // f16 (forward activations of the mixed mode)
            f16_t* cp = reinterpret_cast<f16_t*>(p.C) + co;
            if (full && p.ldc % 8 == 0 && (c_off % 8) == 0) {
                store8(cp, v);
            } else {
                for (int e = 0; e < nv; e++) cp[e] = f2h(v[e]);
            }
        } else {
            bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + co;
            if (full && p.ldc % 8 == 0 && (c_off % 8) == 0) {
                store8(cp, v);
            } else {
                for (int e = 0; e < nv; e++) cp[e] = f2bf(v[e]);
            }
        }
    }
    if (p.colsum || p.colstat) {
        // lanes l, l + CPR, l + 2 CPR, ... of a wave hold the same column chunk: butterfly over those lane bits, then one
        // atomic per column from lanes 0 .. CPR-1 (a Linear's bias gradient without a second pass over its output gradient)
#pragma unroll
        for (int m = 32; m >= CPR; m >>= 1)
#pragma unroll
            for (int e = 0; e < 8; e++) cs[e] += __shfl_xor(cs[e], m);
        if (p.colsum && lane < CPR)
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (n0 + lane * 8 + e < p.N) atomicAdd(p.colsum + n0 + lane * 8 + e, cs[e]);
    }
    if (p.colstat) {
#pragma unroll
        for (int m = 32; m >= CPR; m >>= 1)
#pragma unroll
            for (int e = 0; e < 8; e++) cq[e] += __shfl_xor(cq[e], m);
        // the NTHR / 64 waves of the block hold different rows of the same column chunks: meet in LDS (the tile image is dead by now)
        __syncthreads();
        float* meet = tile;  // [NTHR / 64][2][BN]
        const int wv = threadIdx.x >> 6;
        if (lane < CPR)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                meet[(wv * 2 + 0) * BN + lane * 8 + e] = cs[e];
                meet[(wv * 2 + 1) * BN + lane * 8 + e] = cq[e];
            }
        __syncthreads();
        float* row = p.colstat + (size_t)(m0 / BM) * 2 * p.N;
        for (int i = threadIdx.x; i < 2 * BN; i += NTHR) {
            const int j = i / BN, c = i - j * BN;
            if (n0 + c < p.N) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NTHR / 64; w++) v += meet[(w * 2 + j) * BN + c];
                row[(size_t)j * p.N + n0 + c] = v;
            }
        }
    }
}

// 8 consecutive k (one tap, 8 channels) of pixel row m, or zeros
template <class T, int CV>
AVSR_DEV Raw8<T> gather_chunk(const T* base, const Params& p, int m, int k, int m_lim, int k_lim) {
    Raw8<T> out;
    out.zero();
    if (m >= m_lim || k >= k_lim) return out;
    if (CV == 1 || CV == 3) {
        const int tap = k / p.cC, ci = k - tap * p.cC;
        const int kh = tap / p.cKW, kw = tap - kh * p.cKW;
        const int pix = p.cOH * p.cOW;
        const int n = m / pix, r = m - n * pix;
        const int oh = r / p.cOW, ow = r - oh * p.cOW;
        const int ih = oh * p.cS + kh - p.cPH, iw = ow * p.cS + kw - p.cPW;
        if (ih >= 0 && ih < p.cH && iw >= 0 && iw < p.cW)
            out.load(base + (((size_t)n * p.cH + ih) * p.cW + iw) * p.cC + ci);
    } else if (CV == 2) {
        const int tap = k / p.cC, co = k - tap * p.cC;
        const int kh = tap / p.cKW, kw = tap - kh * p.cKW;
        const int pix = p.cOH * p.cOW;
        const int n = m / pix, r = m - n * pix;
        const int ih = r / p.cOW, iw = r - ih * p.cOW;
        const int th = ih + p.cPH - kh, tw = iw + p.cPW - kw;
        if (th >= 0 && tw >= 0) {
            const int oh = th / p.cS, ow = tw / p.cS;
            if (oh * p.cS == th && ow * p.cS == tw && oh < p.cH && ow < p.cW)
                out.load(base + (((size_t)n * p.cH + oh) * p.cW + ow) * p.cC + co);
        }
    } else {  // CV 4 / 5: one input channel, taps (kt,kh,kw): per-element gather
        const int pix = p.cOH * p.cOW;
        const int n = m / pix, r = m - n * pix;
        const int oh = r / p.cOW, ow = r - oh * p.cOW;
        const int b = n / p.cT, t = n - b * p.cT;
        for (int e = 0; e < 8; e++) {
            const int kk = k + e;
            if (kk >= k_lim) break;
            const int kt = kk / (p.cKH * p.cKW), r2 = kk - kt * (p.cKH * p.cKW);
            const int kh = r2 / p.cKW, kw = r2 - kh * p.cKW;
            const int tt = t + kt - p.cPT, ih = oh * p.cS + kh - p.cPH, iw = ow * p.cS + kw - p.cPW;
            if (tt >= 0 && tt < p.cT && ih >= 0 && ih < p.cH && iw >= 0 && iw < p.cW)
                out.set(e, base + (((size_t)b * p.cT + tt) * p.cH + ih) * p.cW + iw);
        }
    }
    return out;
}

template <class TA, class TB, int NS, int LA, int LB, int BM, int BN, int BK, int CV = 0>
struct Kernel {
    static constexpr int PITCH = BK + 8;  // bf16 elements per LDS row
    static constexpr int NT = 256;
    static constexpr int WM = BM / 2, WN = BN / 2;  // per-wave sub-tile
    static constexpr int TM = WM / 32, TN = WN / 32;
    // items staged per thread
    static constexpr int A_ITEMS = (LA == 0) ? (BM * (BK / 8) / NT) : ((BK / 2) * (BM / 8) / NT);
    static constexpr int B_ITEMS = (LB == 0) ? (BN * (BK / 8) / NT) : ((BK / 2) * (BN / 8) / NT);
    static constexpr int A_RAW = (LA == 0) ? A_ITEMS : 2 * A_ITEMS;
    static constexpr int B_RAW = (LB == 0) ? B_ITEMS : 2 * B_ITEMS;
    static constexpr size_t STAGE_ELEMS = (size_t)NS * (BM + BN) * PITCH;  // one LDS stage (A then B)
    static constexpr size_t LDS_BYTES = 2 * STAGE_ELEMS * sizeof(bf16_t);   // double buffered
    static constexpr bool GA = (CV == 1 || CV == 2 || CV == 4);  // A operand gathered
    static constexpr bool GB = (CV == 3 || CV == 5);             // B operand gathered
    static_assert(A_ITEMS >= 1 && B_ITEMS >= 1, "tile too small for 256 threads");

    // ---- HBM -> registers
    // GATHER: this operand is read through the convolution index map of Params (CV != 0)
    template <class T, int L, int ROWS, int ITEMS, int NRAW, bool GATHER>
    static AVSR_DEV void fetch(Raw8<T> (&raw)[NRAW], const T* base, int ld, int row0, int row_lim, int k0,
                        int k_lim, const Params& p, int skew = 0, int skew_off = 0, int skew_lim = 0) {
        const int tid = threadIdx.x;
        if (L == 0) {
            constexpr int CH = BK / 8;
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const int id = tid + NT * i;
                const int r = id / CH, kc = (id % CH) * 8;
                if (GATHER) raw[i] = gather_chunk<T, CV>(base, p, row0 + r, k0 + kc, row_lim, k_lim);
                else raw[i] = load_chunk<T>(base, ld, row0 + r, k0 + kc, row_lim, k_lim);
            }
        } else {
            constexpr int KP = BK / 2;
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const int id = tid + NT * i;
                const int kp = id % KP, mc = (id / KP) * 8;
                // matrix is [K][rows]: "row" index of the load is k, column is the m/n index
                if (GATHER) {
                    // matrix is [K = pixels][cols = taps*C]: gather_chunk(m = pixel row, k = column)
                    raw[2 * i] = gather_chunk<T, CV>(base, p, k0 + 2 * kp, row0 + mc, k_lim, row_lim);
                    raw[2 * i + 1] = gather_chunk<T, CV>(base, p, k0 + 2 * kp + 1, row0 + mc, k_lim, row_lim);
                } else if (skew) {
                    raw[2 * i] = load_chunk_skew<T>(base, ld, k0 + 2 * kp, row0 + mc, k_lim, row_lim, skew_off, skew_lim);
                    raw[2 * i + 1] = load_chunk_skew<T>(base, ld, k0 + 2 * kp + 1, row0 + mc, k_lim, row_lim, skew_off, skew_lim);
                } else {
                    raw[2 * i] = load_chunk<T>(base, ld, k0 + 2 * kp, row0 + mc, k_lim, row_lim);
                    raw[2 * i + 1] = load_chunk<T>(base, ld, k0 + 2 * kp + 1, row0 + mc, k_lim, row_lim);
                }
            }
        }
    }
    // ---- registers -> LDS ([NS][ROWS][PITCH] bf16)
    template <class T, int L, int ROWS, int ITEMS, int NRAW>
    static AVSR_DEV void stash(const Raw8<T> (&raw)[NRAW], bf16_t* lds) {
        const int tid = threadIdx.x;
        if (L == 0) {
            constexpr int CH = BK / 8;
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const int id = tid + NT * i;
                const int r = id / CH, kc = (id % CH) * 8;
                bf16x8 pl[NS];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    bf16_t s[NS];
                    split_bf16<NS>(raw[i].get(e), s);
#pragma unroll
                    for (int p = 0; p < NS; p++) pl[p][e] = (short)s[p];
                }
#pragma unroll
                for (int p = 0; p < NS; p++)
                    *reinterpret_cast<bf16x8*>(lds + (size_t)p * ROWS * PITCH + r * PITCH + kc) = pl[p];
            }
        } else {
            constexpr int KP = BK / 2;
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const int id = tid + NT * i;
                const int kp = id % KP, mc = (id / KP) * 8;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    bf16_t s0[NS], s1[NS];
                    split_bf16<NS>(raw[2 * i].get(e), s0);
                    split_bf16<NS>(raw[2 * i + 1].get(e), s1);
#pragma unroll
                    for (int p = 0; p < NS; p++) {
                        const uint32_t w = (uint32_t)s0[p] | ((uint32_t)s1[p] << 16);
                        *reinterpret_cast<uint32_t*>(lds + (size_t)p * ROWS * PITCH + (mc + e) * PITCH +
                                                     2 * kp) = w;
                    }
                }
            }
        }
    }

    static AVSR_DEV void run(const Params& p, char* smem) { run_at(p, smem, blockIdx.y, blockIdx.z); }

    // by, bz: the block's m-tile and (batch, k-split) index -- blockIdx.y / blockIdx.z for a single problem, remapped
    // when several problems share one launch (gemm_multi_kernel)
    static AVSR_DEV void run_at(const Params& p, char* smem, int by, int bz) {
        bf16_t* stage0 = reinterpret_cast<bf16_t*>(smem);
        const int zb = bz / p.nsplit, zs = bz % p.nsplit;
        const int zbb = zb / p.batch_h, zbh = zb % p.batch_h;
        const TA* A = reinterpret_cast<const TA*>(p.A) + zbb * p.sAb + zbh * p.sAh;
        const TB* B = reinterpret_cast<const TB*>(p.B) + zbb * p.sBb + zbh * p.sBh;
        const long c_off = zbb * p.sCb + zbh * p.sCh;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave >> 1, wn = wave & 1;
        const int m0 = by * BM, n0 = blockIdx.x * BN;
        const int kbeg = zs * p.k_chunk;
        const int kend = min(p.K, kbeg + p.k_chunk);

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

        // Software pipeline, one barrier per k-tile: while tile t is multiplied out of LDS stage (t&1), the global
        // loads of tile t+1 are in flight and are then written to the other stage.
        Raw8<TA> ra[A_RAW];
        Raw8<TB> rb[B_RAW];
        fetch<TA, LA, BM, A_ITEMS, A_RAW, GA>(ra, A, p.lda, m0, p.M, kbeg, kend, p, p.a_skew, p.skew_off, p.skew_lim);
        fetch<TB, LB, BN, B_ITEMS, B_RAW, GB>(rb, B, p.ldb, n0, p.N, kbeg, kend, p);
        stash<TA, LA, BM, A_ITEMS, A_RAW>(ra, stage0);
        stash<TB, LB, BN, B_ITEMS, B_RAW>(rb, stage0 + (size_t)NS * BM * PITCH);
        __syncthreads();

        int cur = 0;
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            const bool more = (k0 + BK) < kend;
            const bf16_t* As = stage0 + (size_t)cur * STAGE_ELEMS;
            const bf16_t* Bs = As + (size_t)NS * BM * PITCH;
            if (more) {
                fetch<TA, LA, BM, A_ITEMS, A_RAW, GA>(ra, A, p.lda, m0, p.M, k0 + BK, kend, p, p.a_skew, p.skew_off, p.skew_lim);
                fetch<TB, LB, BN, B_ITEMS, B_RAW, GB>(rb, B, p.ldb, n0, p.N, k0 + BK, kend, p);
            }
#pragma unroll
            for (int ks = 0; ks < BK / 16; ks++) {
                Frag<NS> fa[TM], fb[TN];
                const int koff = ks * 16 + 8 * (lane >> 5);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int pl = 0; pl < NS; pl++)
                        fa[i].p[pl] = *reinterpret_cast<const bf16x8*>(
                            As + (size_t)pl * BM * PITCH + (wm * WM + i * 32 + (lane & 31)) * PITCH + koff);
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int pl = 0; pl < NS; pl++)
                        fb[j].p[pl] = *reinterpret_cast<const bf16x8*>(
                            Bs + (size_t)pl * BN * PITCH + (wn * WN + j * 32 + (lane & 31)) * PITCH + koff);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acc[i][j] = mma32<NS>(fa[i], fb[j], acc[i][j]);
            }
            if (more) {
                bf16_t* An = stage0 + (size_t)(cur ^ 1) * STAGE_ELEMS;
                stash<TA, LA, BM, A_ITEMS, A_RAW>(ra, An);
                stash<TB, LB, BN, B_ITEMS, B_RAW>(rb, An + (size_t)NS * BM * PITCH);
            }
            __syncthreads();
            cur ^= 1;
        }

        epilogue_lds<BM, BN, TM, TN>(acc, p, m0, n0, wm * WM, wn * WN, zs, c_off, smem);
    }
};

template <class TA, class TB, int NS, int LA, int LB, int BM, int BN, int BK, int CV = 0>
__global__ __launch_bounds__(256) void gemm_kernel(Params p) {
    AVSR_DYN_SMEM(smem);
    Kernel<TA, TB, NS, LA, LB, BM, BN, BK, CV>::run(p, smem);
}

// Up to three independent problems of one (layout, dtype) family in ONE launch: blockIdx.z enumerates the problems'
// (batch, k-split) indices back to back, blockIdx.y covers the tallest problem (shorter ones exit).  Used where a
// few small batched contractions have no dependence on each other (attention backward: dV, dK, dpos) and each alone
// would leave most of the chip idle and pay its own launch boundary.
struct MultiParams {
    Params p[3];
    int zend[3];  // running end of each problem's blockIdx.z range
    int n;
};
template <class TA, class TB, int NS, int LA, int LB, int BM, int BN, int BK>
__global__ __launch_bounds__(256) void gemm_multi_kernel(MultiParams mp) {
    AVSR_DYN_SMEM(smem);
    int q = 0;
    while (q + 1 < mp.n && (int)blockIdx.z >= mp.zend[q]) q++;
    const Params& p = mp.p[q];
    if ((int)blockIdx.y * BM >= p.M || (int)blockIdx.x * BN >= p.N) return;
    Kernel<TA, TB, NS, LA, LB, BM, BN, BK, 0>::run_at(p, smem, blockIdx.y, blockIdx.z - (q ? mp.zend[q - 1] : 0));
}

template <class TA, class TB, int NS, int LA, int LB>
int launch_multi(const Params* ps, int n, hipStream_t stream) {
    constexpr int BK = 64, BMN = 64;
    MultiParams mp{};
    mp.n = n;
    int gx = 0, gy = 0, gz = 0;
    for (int q = 0; q < n; q++) {
        Params p = ps[q];
        p.k_chunk = ((p.K + BK - 1) / BK) * BK;
        p.nsplit = 1;
        if (p.batch_h < 1) p.batch_h = 1;
        const int nbatch = p.nbatch < 1 ? 1 : p.nbatch;
        gx = std::max(gx, (p.N + BMN - 1) / BMN);
        gy = std::max(gy, (p.M + BMN - 1) / BMN);
        gz += nbatch;
        mp.zend[q] = gz;
        mp.p[q] = p;
    }
    using K = Kernel<TA, TB, NS, LA, LB, BMN, BMN, BK, 0>;
    AVSR_LAUNCH((gemm_multi_kernel<TA, TB, NS, LA, LB, BMN, BMN, BK>), dim3(gx, gy, gz), dim3(256), K::LDS_BYTES, stream, mp);
    return 0;
}

// host-side launch of one (layout, dtype) family: picks the tile shape
template <class TA, class TB, int NS, int LA, int LB, int CV = 0>
int launch(const Params& p0, int force_tile, int split_k, hipStream_t stream) {
    Params p = p0;
    constexpr int BK = 64;
    const long big_tiles = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    const bool big = force_tile == 128 || (force_tile == 0 && big_tiles >= 192);
    const int BMN = big ? 128 : 64;
    if (split_k < 1 || avsr_det()) split_k = 1;  // (deterministic mode: one block per output element, prims.h)
    int kc = (p.K + split_k - 1) / split_k;
    kc = ((kc + BK - 1) / BK) * BK;
    split_k = (p.K + kc - 1) / kc;
    p.k_chunk = kc;
    p.nsplit = split_k;
    if (p.batch_h < 1) p.batch_h = 1;
    const int nbatch = p.nbatch < 1 ? 1 : p.nbatch;
    dim3 grid((p.N + BMN - 1) / BMN, (p.M + BMN - 1) / BMN, split_k * nbatch), block(256);
    if (big) {
        using K = Kernel<TA, TB, NS, LA, LB, 128, 128, BK, CV>;
        AVSR_LAUNCH((gemm_kernel<TA, TB, NS, LA, LB, 128, 128, BK, CV>), grid, block, K::LDS_BYTES, stream, p);
    } else {
        using K = Kernel<TA, TB, NS, LA, LB, 64, 64, BK, CV>;
        AVSR_LAUNCH((gemm_kernel<TA, TB, NS, LA, LB, 64, 64, BK, CV>), grid, block, K::LDS_BYTES, stream, p);
    }
    return 0;
}

// dtype dispatch for one layout
template <int LA, int LB>
int dispatch(const Params& p, int a_dtype, int b_dtype, int precise, int force_tile, int split_k,
             hipStream_t stream) {
    if (precise) {
        if (a_dtype != 0 || b_dtype != 0) return -1;
        return launch<float, float, 2, LA, LB>(p, force_tile, split_k, stream);
    }
    if (a_dtype == 1 && b_dtype == 1) return launch<bf16_t, bf16_t, 1, LA, LB>(p, force_tile, split_k, stream);
    if (a_dtype == 0 && b_dtype == 1) return launch<float, bf16_t, 1, LA, LB>(p, force_tile, split_k, stream);
    if (a_dtype == 1 && b_dtype == 0) return launch<bf16_t, float, 1, LA, LB>(p, force_tile, split_k, stream);
    return launch<float, float, 1, LA, LB>(p, force_tile, split_k, stream);
}

}  // namespace avsr_gemm_impl
