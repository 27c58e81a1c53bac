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
    int cols, float eps, bf16_t* __restrict__ y2 = nullptr) {
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
            if (y2) store8(y2 + (size_t)row * cols + c * 8, o);  // bf16 twin (hpf mode: what the backward pass will read)
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma ; optional  dx += dres.  One wave per row, RPW rows
// per wave.  When the consumer of dx is the output Linear of the previous pre-LN sub-layer, that Linear's whole "backward
// prologue" is produced in the same pass: gout = bf16(alpha * dropout(dx)) (what its two backward GEMMs contract) and
// gsum[c] += sum_r gout[r,c] (its bias gradient: per-lane partial sums over the wave's rows, combined across the block's
// waves in LDS, one atomic per column and block), so the f32 dx is not re-read by a cast / column-sum launch.
template <class TDY>
AVSR_DEV void layernorm_bwd_dx_block(
    const TDY* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in, const float* __restrict__ dres,
    float* __restrict__ dx, bf16_t* __restrict__ gout, float* __restrict__ gsum, float alpha0, float drop_p, uint64_t seed,
    int rows, int cols, int blk, int rpw, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = cols >> 3;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    float pc[LN_MAXV][8];
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) pc[i][e] = 0.f;
    for (int rr = 0; rr < rpw; rr++) {
        const int row = (blk * rpw + rr) * LN_WAVES + wave;
        if (row >= rows) break;  // wave-uniform
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
                    float rv[8];
                    load8(dres + (size_t)row * cols + c * 8, rv);
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] += rv[e];
                }
                store8(dxr + c * 8, o);
                if (gout) {
                    float q[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float v = o[e] * alpha0 *
                            dropout_scale(seed, (uint64_t)row * (uint64_t)cols + c * 8 + e, drop_p, inv_keep);
                        q[e] = bf2f(f2bf(v));
                        pc[i][e] += q[e];
                    }
                    store8(gout + (size_t)row * cols + c * 8, q);
                }
            }
        }
    }
    if (!gsum) return;  // block-uniform
    float* mine = red + (size_t)wave * cols;
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++) {
        const int c = lane + 64 * i;
        if (c < nvec) store8(mine + c * 8, pc[i]);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < cols; j += LN_THREADS) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < LN_WAVES; w++) t += red[(size_t)w * cols + j];
        atomicAdd(gsum + j, t);
    }
}

// dgamma[c] += sum_r dy[r,c]*xhat[r,c] ; dbeta[c] += sum_r dy[r,c].  thread = (8-column chunk, row lane).
template <class TDY>
AVSR_DEV void layernorm_bwd_param_block(
    const TDY* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int cols,
    int rows_per_block, int CL, int bx, int by, float* red) {
    const int cv = cols >> 3;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL, RL = LN_THREADS / CL;
    const int cc = bx * CL + cl;
    const int r0 = by * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    float pg[8], pb[8];
#pragma unroll
    for (int e = 0; e < 8; e++) pg[e] = pb[e] = 0.f;
    if (cc < cv) {
        constexpr int U = 4;  // independent rows in flight per thread: the loop is a chain of HBM/L2 round trips otherwise (8: no gain, round 6)
        for (int rb = r0 + rl; rb < r1; rb += U * RL) {
            float xv[U][8], dv[U][8], m[U], rs[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int r = rb + u * RL;
                if (r < r1) {
                    load8(x + (size_t)r * cols + cc * 8, xv[u]);
                    load8(dy + (size_t)r * cols + cc * 8, dv[u]);
                    m[u] = mean_in[r];
                    rs[u] = rstd_in[r];
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (rb + u * RL >= r1) continue;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    pg[e] += dv[u][e] * (xv[u][e] - m[u]) * rs[u];
                    pb[e] += dv[u][e];
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
        red[threadIdx.x * 16 + e] = pg[e];
        red[threadIdx.x * 16 + 8 + e] = pb[e];
    }
    __syncthreads();
    if (rl == 0 && cc < cv) {
        for (int q = 1; q < RL; q++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                pg[e] += red[(q * CL + cl) * 16 + e];
                pb[e] += red[(q * CL + cl) * 16 + 8 + e];
            }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            atomicAdd(dgamma + cc * 8 + e, pg[e]);
            atomicAdd(dbeta + cc * 8 + e, pb[e]);
        }
    }
}

// Both halves of the backward pass in one grid: blocks [0, npx*npy) reduce the parameter gradients (the long pole: few
// blocks, each walking 64 rows), the rest compute dx one wave per row.  The two are independent, and one of them alone
// does not fill the chip at B*T <= 1600 rows -- together they overlap and one launch boundary disappears.
template <class TDY>
__global__ __launch_bounds__(LN_THREADS) void layernorm_bwd_kernel(
    const TDY* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in, const float* __restrict__ dres,
    float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, bf16_t* __restrict__ gout,
    float* __restrict__ gsum, float alpha0, float drop_p, uint64_t seed0, const uint64_t* __restrict__ seed_dev, int rows,
    int cols, int rows_per_block, int CL, int npx, int npy, int rpw) {
    AVSR_DYN_SMEM(smem);  // max(LN_THREADS * 16, LN_WAVES * cols) floats
    float* red = reinterpret_cast<float*>(smem);
    const int b = blockIdx.x;
    if (b < npx * npy)
        layernorm_bwd_param_block<TDY>(dy, x, mean_in, rstd_in, dgamma, dbeta, rows, cols, rows_per_block, CL, b % npx,
                                       b / npx, red);
    else
        layernorm_bwd_dx_block<TDY>(dy, x, gamma, mean_in, rstd_in, dres, dx, gout, gsum, alpha0, drop_p,
                                    seed0 + (seed_dev ? *seed_dev : 0ull), rows, cols, b - npx * npy, rpw, red);
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

// f32 output + its bf16 twin in one pass (the "hpf" numerical mode: f32 forward, bf16 copies saved for the backward pass)
extern "C" int avsr_layernorm_fwd2(const float* x, const float* gamma, const float* beta, float* y, void* y2, float* mean,
                                   float* rstd, int rows, int cols, float eps, hipStream_t stream) {
    AVSR_REQUIRE(cols % 8 == 0 && cols <= 64 * 8 * LN_MAXV, "layernorm: cols must be %8 and <= 2048");
    if (rows == 0) return 0;
    dim3 grid((rows + LN_WAVES - 1) / LN_WAVES), block(LN_THREADS);
    AVSR_LAUNCH((layernorm_fwd_kernel<float>), grid, block, 0, stream, x, gamma, beta, y, mean, rstd, rows, cols, eps, (bf16_t*)y2);
    AVSR_CHECK_LAUNCH("layernorm_fwd2");
    return 0;
}

// f16 output + its bf16 twin (may be NULL) in one pass (the "mixed" numerical mode: f16 forward operands, bf16 backward)
extern "C" int avsr_layernorm_fwd_h16(const float* x, const float* gamma, const float* beta, void* y, void* y2, float* mean,
                                      float* rstd, int rows, int cols, float eps, hipStream_t stream) {
    AVSR_REQUIRE(cols % 8 == 0 && cols <= 64 * 8 * LN_MAXV, "layernorm: cols must be %8 and <= 2048");
    if (rows == 0) return 0;
    dim3 grid((rows + LN_WAVES - 1) / LN_WAVES), block(LN_THREADS);
    AVSR_LAUNCH((layernorm_fwd_kernel<f16_t>), grid, block, 0, stream, x, gamma, beta, (f16_t*)y, mean, rstd, rows, cols, eps,
                (bf16_t*)y2);
    AVSR_CHECK_LAUNCH("layernorm_fwd_h16");
    return 0;
}

// dgamma / dbeta (and gsum) are ACCUMULATED into (caller zeroes them or carries grads over).
// gout (bf16 [rows][cols], may be NULL) = bf16(alpha * dropout(dx)) with the dropout stream of avsr_cast_transpose_colsum
// (element index row * cols + col); gsum (f32 [cols], may be NULL, needs gout) += column sums of gout.
extern "C" int avsr_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma,
                                  const float* mean, const float* rstd, const float* dres, float* dx,
                                  float* dgamma, float* dbeta, void* gout, float* gsum, float alpha, float drop_p,
                                  uint64_t seed, const uint64_t* seed_dev, int rows, int cols, hipStream_t stream) {
    AVSR_REQUIRE(cols % 8 == 0 && cols <= 64 * 8 * LN_MAXV, "layernorm: cols must be %8 and <= 2048");
    AVSR_REQUIRE(gsum == nullptr || gout != nullptr, "layernorm_bwd: gsum needs gout");
    if (rows == 0) return 0;
    // with a column-sum output two rows per wave once the grid fills the chip anyway: half the atomics per launch
    // deterministic mode: one parameter-gradient block per column group (all rows, fixed order), and the column sums of gout from an
    // ordered pass over the stored values instead of one atomic per block and column
    float* gsum_det = nullptr;
    if (avsr_det() && gsum) {
        gsum_det = gsum;
        gsum = nullptr;
    }
    const int rpw = avsr_tune_knobs[25] > 0 ? avsr_tune_knobs[25] : ((gsum && rows > 1024) ? 2 : 1);  // knob 25: rows per wave of the dx blocks (A/B)
    const int ndx = (rows + LN_WAVES * rpw - 1) / (LN_WAVES * rpw);
    const int cv = cols >> 3;
    const int CL = cv >= 32 ? 32 : (cv >= 16 ? 16 : 8);
    const int rpb = avsr_det() ? rows : 8 * (LN_THREADS / CL);  // 64 rows per block; 32 was measured slower (more colliding atomics per column)
    const int npx = (cv + CL - 1) / CL, npy = (rows + rpb - 1) / rpb;
    dim3 grid(npx * npy + ndx), block(LN_THREADS);
    size_t lds = (size_t)LN_THREADS * 16 * sizeof(float);
    if (gsum && (size_t)LN_WAVES * cols * sizeof(float) > lds) lds = (size_t)LN_WAVES * cols * sizeof(float);
    if (dy_dtype == 0)
        AVSR_LAUNCH((layernorm_bwd_kernel<float>), grid, block, lds, stream, (const float*)dy, x, gamma, mean, rstd, dres, dx,
                    dgamma, dbeta, (bf16_t*)gout, gsum, alpha, drop_p, seed, seed_dev, rows, cols, rpb, CL, npx, npy, rpw);
    else
        AVSR_LAUNCH((layernorm_bwd_kernel<bf16_t>), grid, block, lds, stream, (const bf16_t*)dy, x, gamma, mean, rstd, dres,
                    dx, dgamma, dbeta, (bf16_t*)gout, gsum, alpha, drop_p, seed, seed_dev, rows, cols, rpb, CL, npx, npy, rpw);
    if (gsum_det) avsr_colsum_det(gout, 1, cols, rows, cols, gsum_det, stream);
    AVSR_CHECK_LAUNCH("layernorm_bwd");
    return 0;
}
