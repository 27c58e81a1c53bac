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
constexpr int DW_TW = 64;     // time steps per block (weight gradient)
constexpr int DW_MAXK = 31;

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
                if (sizeof(T) == 2) v[e] = bf2f(f2bf(v[e]));
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
                                                     int C, int K, int flip, int glu_in, const T* __restrict__ glu_a) {
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
                if (sizeof(T) == 2) acc = bf2f(f2bf(acc));  // the gradient the stand-alone path stored before its GLU backward
                const float lin = Elem<T>::ld(glu_a + row * 2 * C + c0 + tx), sg = avsr_sigmoid(Elem<T>::ld(glu_a + row * 2 * C + C + c0 + tx));
                Elem<T>::st(y + row * 2 * C + c0 + tx, acc * sg);
                Elem<T>::st(y + row * 2 * C + C + c0 + tx, acc * lin * sg * (1.f - sg));
            } else {
                Elem<T>::st(y + row * C + c0 + tx, acc);
            }
        }
    }
}

// dw[c,k] += sum_{b,t} dy[b,t,c] x[b,t+k-pad,c] ;  db[c] += sum dy
// A block owns 64 channels and walks every gridDim.y-th (batch element, 64-step time tile) pair with its sums in
// registers; one round of atomics per block at the end.  (One block per tile put B*T/64 colliding atomics on every
// 128-byte line of dw -- one line per channel -- and the serialised atomics, not the arithmetic, set the kernel time.)
template <class T>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           float* __restrict__ dw, float* __restrict__ db, int B, int Tlen,
                                                           int C, int K, int glu_in) {
    __shared__ float xs[(DW_TW + DW_MAXK - 1) * DW_CH];
    __shared__ float ds[DW_TW * DW_CH];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c0 = blockIdx.x * DW_CH;
    const int pad = (K - 1) / 2;
    const int tiles_t = (Tlen + DW_TW - 1) / DW_TW, items = B * tiles_t;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = 0.f;
    float sb = 0.f;
    for (int it = blockIdx.y; it < items; it += gridDim.y) {
        const int b = it / tiles_t, t0 = (it - b * tiles_t) * DW_TW;
        __syncthreads();  // the previous item's tiles are no longer read
        if (glu_in) stage_time_tile_glu<T>(xs, x, b, t0 - pad, DW_TW + K - 1, Tlen, C, c0);
        else stage_time_tile<T>(xs, x, b, t0 - pad, DW_TW + K - 1, Tlen, C, c0);
        stage_time_tile<T>(ds, dy, b, t0, DW_TW, Tlen, C, c0);
        __syncthreads();
        // branch-free inner loop: taps beyond K read a clamped (valid) row and are simply never written back, so the
        // eight LDS reads of a time step issue together instead of one latency-exposed round trip per tap
#pragma unroll 2
        for (int t = 0; t < DW_TW; t++) {
            const float g = ds[t * DW_CH + tx];
            sb += g;
            float xv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) xv[i] = xs[(t + min(ty + 4 * i, K - 1)) * DW_CH + tx];
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] += g * xv[i];
        }
    }
    if (c0 + tx < C) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int k = ty + 4 * i;
            if (k < K) atomicAdd(dw + (long)(c0 + tx) * K + k, acc[i]);
        }
        if (ty == 0 && db) atomicAdd(db + c0 + tx, sb);
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

extern "C" int avsr_dwconv_wgrad(const void* x, const void* dy, int dtype, float* dw, float* db, int B, int T, int C,
                                 int K, int glu_in, hipStream_t stream) {
    AVSR_REQUIRE(K >= 1 && K <= DW_MAXK && (K & 1), "dwconv: K must be odd and <= 31");
    AVSR_REQUIRE(C % 8 == 0, "dwconv: C must be a multiple of 8");
    if (B <= 0 || T <= 0) return 0;
    const int items = B * ((T + DW_TW - 1) / DW_TW), cblocks = (C + DW_CH - 1) / DW_CH;
    int chunks = (512 + cblocks - 1) / cblocks;  // about two blocks per CU, a few items each
    if (chunks > items) chunks = items;
    if (chunks > 16) chunks = 16;  // more, smaller blocks were measured slower (43 chunks: 31 -> 49 us): the per-block atomics dominate
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
