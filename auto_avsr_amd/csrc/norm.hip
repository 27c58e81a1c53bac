// norm.hip -- LayerNorm forward/backward (HBM-bound; one 64-lane wave per row,
// wave-shuffle reductions, 16/32-byte vector accesses).
//
// Replaces the implicit ATen LayerNorm behind
//   espnet/nets/pytorch_backend/transformer/layer_norm.py:12-33  (eps = 1e-12)
// used 5x per Conformer block (conformer_encoder.py:79-88), 3x per decoder
// block (transformer_decoder.py:55-57) and as after_norm in both stacks.
#include "prims.h"
#include "avsr_hip.h"

namespace {

constexpr int LN_THREADS = 256;
constexpr int LN_WAVES = LN_THREADS / 64;
constexpr int LN_MAXV = 4;  // 8-element vectors per lane -> cols <= 64*8*4 = 2048

template <class TY>
__global__ __launch_bounds__(LN_THREADS) void layernorm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    TY* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows,
    int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int row = blockIdx.x * LN_WAVES + wave;
    if (row >= rows) return;  // whole wave leaves together
    const float* xr = x + (size_t)row * cols;
    const int nvec = cols >> 3;  // cols % 8 == 0 (checked on the host side)
    float v[LN_MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            load8(xr + c * 8, v[i]);
#pragma unroll
            for (int e = 0; e < 8; e++) s += v[i][e];
        }
    }
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++) {
        const int c = lane + 64 * i;
        if (c < nvec) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
    }
    const float var = wave_sum(q) / (float)cols;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
    TY* yr = y + (size_t)row * cols;
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            float g[8], b[8], o[8];
            load8(gamma + c * 8, g);
            load8(beta + c * 8, b);
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            store8(yr + c * 8, o);
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma ; optional  dx += dres.
// dgamma/dbeta: per-lane register partials over the block's rows -> LDS -> one atomic per column.
template <class TDY>
__global__ __launch_bounds__(LN_THREADS) void layernorm_bwd_kernel(
    const TDY* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const float* __restrict__ dres, float* __restrict__ dx, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int rows, int cols, int rows_per_block) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nvec = cols >> 3;
    float pg[LN_MAXV][8], pb[LN_MAXV][8];
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) pg[i][e] = pb[i][e] = 0.f;

    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    for (int row = r0 + wave; row < r1; row += LN_WAVES) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        const float* xr = x + (size_t)row * cols;
        const TDY* dyr = dy + (size_t)row * cols;
        float xh[LN_MAXV][8], g[LN_MAXV][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                float xv[8], dv[8], gm[8];
                load8(xr + c * 8, xv);
                load8(dyr + c * 8, dv);
                load8(gamma + c * 8, gm);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    xh[i][e] = (xv[e] - mean) * rstd;
                    g[i][e] = dv[e] * gm[e];
                    s1 += g[i][e];
                    s2 += g[i][e] * xh[i][e];
                    pg[i][e] += dv[e] * xh[i][e];
                    pb[i][e] += dv[e];
                }
            }
        }
        s1 = wave_sum(s1) / (float)cols;
        s2 = wave_sum(s2) / (float)cols;
        float* dxr = dx + (size_t)row * cols;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = rstd * (g[i][e] - s1 - xh[i][e] * s2);
                if (dres) {
                    float rr[8];
                    load8(dres + (size_t)row * cols + c * 8, rr);
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] += rr[e];
                }
                store8(dxr + c * 8, o);
            }
        }
    }
    // cross-wave reduction of the parameter-gradient partials
    __shared__ float red[LN_WAVES][64 * 8 + 8];
    for (int i = 0; i < LN_MAXV; i++) {
        const int c = lane + 64 * i;
        for (int pass = 0; pass < 2; pass++) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; e++) red[wave][lane * 8 + e] = pass ? pb[i][e] : pg[i][e];
            __syncthreads();
            if (wave == 0 && c < nvec) {
                float* dst = pass ? dbeta : dgamma;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < LN_WAVES; w++) t += red[w][lane * 8 + e];
                    atomicAdd(dst + c * 8 + e, t);
                }
            }
        }
    }
}

}  // namespace

extern "C" int avsr_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y,
                                  int y_dtype, float* mean, float* rstd, int rows, int cols,
                                  float eps, hipStream_t stream) {
    AVSR_REQUIRE(cols % 8 == 0 && cols <= 64 * 8 * LN_MAXV, "layernorm: cols must be %8 and <= 2048");
    if (rows == 0) return 0;
    dim3 grid((rows + LN_WAVES - 1) / LN_WAVES), block(LN_THREADS);
    if (y_dtype == 0)
        AVSR_LAUNCH((layernorm_fwd_kernel<float>), grid, block, 0, stream, x, gamma, beta, (float*)y,
                    mean, rstd, rows, cols, eps);
    else
        AVSR_LAUNCH((layernorm_fwd_kernel<bf16_t>), grid, block, 0, stream, x, gamma, beta,
                    (bf16_t*)y, mean, rstd, rows, cols, eps);
    AVSR_CHECK_LAUNCH("layernorm_fwd");
    return 0;
}

// dgamma / dbeta are ACCUMULATED into (caller zeroes them or carries grads over).
extern "C" int avsr_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma,
                                  const float* mean, const float* rstd, const float* dres, float* dx,
                                  float* dgamma, float* dbeta, int rows, int cols,
                                  hipStream_t stream) {
    AVSR_REQUIRE(cols % 8 == 0 && cols <= 64 * 8 * LN_MAXV, "layernorm: cols must be %8 and <= 2048");
    if (rows == 0) return 0;
    const int rpb = 16;
    dim3 grid((rows + rpb - 1) / rpb), block(LN_THREADS);
    if (dy_dtype == 0)
        AVSR_LAUNCH((layernorm_bwd_kernel<float>), grid, block, 0, stream, (const float*)dy, x, gamma,
                    mean, rstd, dres, dx, dgamma, dbeta, rows, cols, rpb);
    else
        AVSR_LAUNCH((layernorm_bwd_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)dy, x,
                    gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, cols, rpb);
    AVSR_CHECK_LAUNCH("layernorm_bwd");
    return 0;
}
