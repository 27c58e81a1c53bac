// convmod.hip -- depthwise 1-D convolution of the Conformer ConvolutionModule on (B,T,C) activations.
//
// Replaces  torch.nn.Conv1d(C, C, K, padding=(K-1)//2, groups=C)  of conformer_encoder.py:25,33 and its
// gradients.  The reference transposes to (B,C,T) first (conformer_encoder.py:31,35); here the channel
// dimension stays innermost so that lanes read consecutive channels (coalesced) and the K-tap window over
// time is staged once in LDS and re-used by every output of the tile.  HBM-bound: 2 B/element in, 2 out.
#include "prims.h"
#include "avsr_hip.h"

namespace {

constexpr int DW_CH = 64;     // channels per block (one per lane)
constexpr int DW_TT = 32;     // outputs per block along time (forward / data gradient)
constexpr int DW_MAXK = 31;

// an intermediate the stand-alone op sequence would have stored in the activation type: bf16 tensors re-round (bit-compatible
// with that sequence, which the bf16 backward pass recomputes); f32 / f16 (forward-only) tensors keep the f32 value
template <class T> AVSR_DEV float round_as_stored(float v) { return v; }
template <> AVSR_DEV float round_as_stored<bf16_t>(float v) { return bf2f(f2bf(v)); }

template <class T>
AVSR_DEV void stage_time_tile(float* xs, const T* x, int b, int t_first, int nrows, int Tlen, int C, int c0) {
    for (int id = threadIdx.x; id < nrows * (DW_CH / 8); id += 256) {
        const int r = id / (DW_CH / 8), cc = (id % (DW_CH / 8)) * 8;
        const int t = t_first + r;
        float v[8];
        if (t >= 0 && t < Tlen && c0 + cc < C) load8(x + ((long)b * Tlen + t) * C + c0 + cc, v);
        else
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) xs[r * DW_CH + cc + e] = v[e];
    }
}

// The same tile of glu(a) = a[:, :C] * sigmoid(a[:, C:]) for a pre-GLU tensor a [B*T, 2C] (conformer_encoder.py:32): the
// depthwise convolution's input is never materialised.  Values are rounded to the storage type exactly as the stand-alone
// GLU kernel would have stored them.
template <class T>
AVSR_DEV void stage_time_tile_glu(float* xs, const T* a, int b, int t_first, int nrows, int Tlen, int C, int c0) {
    for (int id = threadIdx.x; id < nrows * (DW_CH / 8); id += 256) {
        const int r = id / (DW_CH / 8), cc = (id % (DW_CH / 8)) * 8;
        const int t = t_first + r;
        float v[8], g[8];
        if (t >= 0 && t < Tlen && c0 + cc < C) {
            const T* row = a + ((long)b * Tlen + t) * 2 * C + c0 + cc;
            load8(row, v);
            load8(row + C, g);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                v[e] *= avsr_sigmoid(g[e]);
                v[e] = round_as_stored<T>(v[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; e++) xs[r * DW_CH + cc + e] = v[e];
    }
}

// y[b,t,c] = bias[c] + sum_k w[c,k] x[b,t+k-pad,c]      (flip=1: taps reversed -> data gradient)
// glu_in: x is the pre-GLU tensor [B*T, 2C] and the convolution runs on glu(x).
// glu_a != NULL (with flip = 1): the result r = d glu(a) / the data gradient of the convolution -- is pushed through the
// GLU backward on the way out: y is da [B*T, 2C] = (r * sigmoid(g), r * a_lin * sigmoid(g) * (1 - sigmoid(g))).
template <class T>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, T* __restrict__ y, int Tlen,
                                                     int C, int K, int flip, int glu_in, const T* __restrict__ glu_a,
                                                     bf16_t* __restrict__ y2 = nullptr) {
    __shared__ float xs[(DW_TT + DW_MAXK - 1) * DW_CH];
    __shared__ float ws[DW_MAXK * DW_CH];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c0 = blockIdx.x * DW_CH, t0 = blockIdx.y * DW_TT, b = blockIdx.z;
    const int pad = (K - 1) / 2;
    // all DW_MAXK taps are always multiplied (weights beyond K are zero, their x rows staged as real data or zeros):
    // a fixed trip count lets the LDS reads of the taps issue back to back
    if (glu_in) stage_time_tile_glu<T>(xs, x, b, t0 - pad, DW_TT + DW_MAXK - 1, Tlen, C, c0);
    else stage_time_tile<T>(xs, x, b, t0 - pad, DW_TT + DW_MAXK - 1, Tlen, C, c0);
    for (int k = ty; k < DW_MAXK; k += 4)
        ws[k * DW_CH + tx] = (k < K && c0 + tx < C) ? w[(long)(c0 + tx) * K + (flip ? K - 1 - k : k)] : 0.f;
    __syncthreads();
    const float bv = (bias && c0 + tx < C) ? bias[c0 + tx] : 0.f;
#pragma unroll
    for (int o = 0; o < DW_TT / 4; o++) {
        const int t = ty * (DW_TT / 4) + o;
        float acc = bv;
#pragma unroll
        for (int k = 0; k < DW_MAXK; k++) acc += ws[k * DW_CH + tx] * xs[(t + k) * DW_CH + tx];
        if (t0 + t < Tlen && c0 + tx < C) {
            const long row = (long)b * Tlen + t0 + t;
            if (glu_a) {
                acc = round_as_stored<T>(acc);  // the gradient the stand-alone path stored before its GLU backward
                const float lin = Elem<T>::ld(glu_a + row * 2 * C + c0 + tx), sg = avsr_sigmoid(Elem<T>::ld(glu_a + row * 2 * C + C + c0 + tx));
                Elem<T>::st(y + row * 2 * C + c0 + tx, acc * sg);
                Elem<T>::st(y + row * 2 * C + C + c0 + tx, acc * lin * sg * (1.f - sg));
            } else {
                Elem<T>::st(y + row * C + c0 + tx, acc);
                if (y2) y2[row * C + c0 + tx] = f2bf(acc);  // bf16 twin of an f32 output ("hpf" mode)
            }
        }
    }
}

// dw[c,k] += sum_{b,t} dy[b,t,c] x[b,t+k-pad,c] ;  db[c] += sum dy
// A block owns EIGHT channels x all taps: thread = (tap lane k = 0..31, channel 0..7); lane 31 carries the bias sum.  It walks
// every gridDim.y-th (batch element, 256-step time tile) pair: the tile's dy rows and x rows (+ K - 1 halo) are fetched into
// registers one item ahead, parked in LDS (row pitch 9 floats: the 32 tap lanes of a channel read 32 different banks) and
// each thread runs its own 256-term sum -- no cross-thread reduction, and (C / 8) x gridDim.y x 256 atomics at the end.
// (Round 2's version put 64 channels x 4 tap groups in a block: 16 x more atomics onto the same 95 KB of dw; they, not the
// arithmetic, set its 29 us.)
constexpr int DW_W2 = 256;  // time steps per item
constexpr int DW_P2 = 9;    // LDS row pitch (floats)
template <class T>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           float* __restrict__ dw, float* __restrict__ db, int B, int Tlen,
                                                           int C, int K, int glu_in) {
    __shared__ float xs[(DW_W2 + DW_MAXK - 1) * DW_P2];
    __shared__ float ds[DW_W2 * DW_P2];
    const int k = threadIdx.x & 31, cl = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 8;
    const int pad = (K - 1) / 2, xrows = DW_W2 + K - 1;
    const int tiles_t = (Tlen + DW_W2 - 1) / DW_W2, items = B * tiles_t;
    const int kk = min(k, K - 1);
    float vx[2][8], vd[8];
    auto fetch_row = [&](const T* src, int b, int t, bool glu, float (&v)[8]) {
        if (t >= 0 && t < Tlen) {
            if (glu) {
                const T* row = src + ((long)b * Tlen + t) * 2 * C + c0;
                float g[8];
                load8(row, v);
                load8(row + C, g);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    v[e] *= avsr_sigmoid(g[e]);
                    v[e] = round_as_stored<T>(v[e]);  // as the stand-alone GLU kernel would have stored it
                }
            } else {
                load8(src + ((long)b * Tlen + t) * C + c0, v);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = 0.f;
        }
    };
    auto fetch = [&](int it) {
        const int b = it / tiles_t, t0 = (it - b * tiles_t) * DW_W2;
        fetch_row(x, b, t0 - pad + (int)threadIdx.x, glu_in != 0, vx[0]);
        if ((int)threadIdx.x + 256 < xrows) fetch_row(x, b, t0 - pad + (int)threadIdx.x + 256, glu_in != 0, vx[1]);
        fetch_row(dy, b, t0 + (int)threadIdx.x, false, vd);
    };
    float acc = 0.f, sb = 0.f;
    int it = blockIdx.y;
    if (it < items) fetch(it);
    for (; it < items; it += gridDim.y) {
        __syncthreads();  // the previous item's tiles are no longer read
#pragma unroll
        for (int e = 0; e < 8; e++) {
            xs[threadIdx.x * DW_P2 + e] = vx[0][e];
            if ((int)threadIdx.x + 256 < xrows) xs[(threadIdx.x + 256) * DW_P2 + e] = vx[1][e];
            ds[threadIdx.x * DW_P2 + e] = vd[e];
        }
        __syncthreads();
        if (it + (int)gridDim.y < items) fetch(it + gridDim.y);  // in flight during the sums below
        float a4[4] = {0.f, 0.f, 0.f, 0.f}, s4 = 0.f;
        // only the rows the utterance has: beyond them the dy rows are zeros (round 6: at T ~ 100 a 256-step tile spent 60 % of
        // its sum on them -- 26.9 -> 15.0 us per launch, profiles/r6_microbench_convmod.txt; same bits: the skipped terms were exact zeros)
        const int tcur = (it - (it / tiles_t) * tiles_t) * DW_W2, tlim = min(DW_W2, (Tlen - tcur + 3) & ~3);
#pragma unroll 4
        for (int t = 0; t < tlim; t += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float g = ds[(t + u) * DW_P2 + cl];
                a4[u] += g * xs[(t + u + kk) * DW_P2 + cl];
                s4 += g;
            }
        }
        acc += (a4[0] + a4[1]) + (a4[2] + a4[3]);
        sb += s4;
    }
    if (c0 + cl < C) {
        if (k < K) atomicAdd(dw + (long)(c0 + cl) * K + k, acc);
        if (k == 31 && db) atomicAdd(db + c0 + cl, sb);
    }
}

}  // namespace

extern "C" int avsr_dwconv_fwd(const void* x, int dtype, const float* w, const float* bias, void* y, int B, int T,
                               int C, int K, int flip, int glu_in, const void* glu_a, hipStream_t stream) {
    AVSR_REQUIRE(K >= 1 && K <= DW_MAXK && (K & 1), "dwconv: K must be odd and <= 31");
    AVSR_REQUIRE(C % 8 == 0, "dwconv: C must be a multiple of 8");
    if (B <= 0 || T <= 0) return 0;
    dim3 grid((C + DW_CH - 1) / DW_CH, (T + DW_TT - 1) / DW_TT, B), block(256);
    if (dtype == 0)
        AVSR_LAUNCH((dwconv_kernel<float>), grid, block, 0, stream, (const float*)x, w, bias, (float*)y, T, C, K, flip, glu_in,
                    (const float*)glu_a);
    else
        AVSR_LAUNCH((dwconv_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)x, w, bias, (bf16_t*)y, T, C, K, flip,
                    glu_in, (const bf16_t*)glu_a);
    AVSR_CHECK_LAUNCH("dwconv_fwd");
    return 0;
}

// f16 forward (input and output IEEE half) + the bf16 twin y2 (may be NULL) of its output in one pass
extern "C" int avsr_dwconv_fwd_h16(const void* x, const float* w, const float* bias, void* y, void* y2, int B, int T, int C, int K,
                                   int glu_in, hipStream_t stream) {
    AVSR_REQUIRE(K >= 1 && K <= DW_MAXK && (K & 1), "dwconv: K must be odd and <= 31");
    AVSR_REQUIRE(C % 8 == 0, "dwconv: C must be a multiple of 8");
    if (B <= 0 || T <= 0) return 0;
    dim3 grid((C + DW_CH - 1) / DW_CH, (T + DW_TT - 1) / DW_TT, B), block(256);
    AVSR_LAUNCH((dwconv_kernel<f16_t>), grid, block, 0, stream, (const f16_t*)x, w, bias, (f16_t*)y, T, C, K, 0, glu_in,
                (const f16_t*)nullptr, (bf16_t*)y2);
    AVSR_CHECK_LAUNCH("dwconv_fwd_h16");
    return 0;
}

// f32 forward + the bf16 twin of its output in one pass
extern "C" int avsr_dwconv_fwd2(const float* x, const float* w, const float* bias, float* y, void* y2, int B, int T, int C, int K,
                                int glu_in, hipStream_t stream) {
    AVSR_REQUIRE(K >= 1 && K <= DW_MAXK && (K & 1), "dwconv: K must be odd and <= 31");
    AVSR_REQUIRE(C % 8 == 0, "dwconv: C must be a multiple of 8");
    if (B <= 0 || T <= 0) return 0;
    dim3 grid((C + DW_CH - 1) / DW_CH, (T + DW_TT - 1) / DW_TT, B), block(256);
    AVSR_LAUNCH((dwconv_kernel<float>), grid, block, 0, stream, x, w, bias, y, T, C, K, 0, glu_in, (const float*)nullptr,
                (bf16_t*)y2);
    AVSR_CHECK_LAUNCH("dwconv_fwd2");
    return 0;
}

extern "C" int avsr_dwconv_wgrad(const void* x, const void* dy, int dtype, float* dw, float* db, int B, int T, int C,
                                 int K, int glu_in, hipStream_t stream) {
    AVSR_REQUIRE(K >= 1 && K <= DW_MAXK && (K & 1), "dwconv: K must be odd and <= 31");
    AVSR_REQUIRE(C % 8 == 0, "dwconv: C must be a multiple of 8");
    if (B <= 0 || T <= 0) return 0;
    const int items = B * ((T + DW_W2 - 1) / DW_W2), cblocks = C / 8;
    int chunks = (512 + cblocks - 1) / cblocks;  // about two blocks per CU
    if (chunks > items) chunks = items;
    if (chunks > 8) chunks = 8;
    if (avsr_det()) chunks = 1;  // deterministic mode: one block per channel group walks every (utterance, tile) item in order
    dim3 grid(cblocks, chunks), block(256);
    if (dtype == 0)
        AVSR_LAUNCH((dwconv_wgrad_kernel<float>), grid, block, 0, stream, (const float*)x, (const float*)dy, dw, db, B, T, C, K,
                    glu_in);
    else
        AVSR_LAUNCH((dwconv_wgrad_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)dy, dw, db, B, T, C,
                    K, glu_in);
    AVSR_CHECK_LAUNCH("dwconv_wgrad");
    return 0;
}
