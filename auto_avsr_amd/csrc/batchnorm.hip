// batchnorm.hip -- training-mode BatchNorm over the rows of a channels-last [rows, C] tensor, fused with
// the activation (SiLU) and an optional residual add, forward and backward.
//
// Replaces BatchNorm1d of the ConvolutionModule (conformer_encoder.py:26,33 + SiLU :28) and
// BatchNorm2d/3d of the visual front-end (resnet.py:34,61,77,212).  Statistics are taken over every row
// including padded frames (the reference does not mask them).
//
// Structure = what the cross-rank synchronisation of `sync_batchnorm=True` (train.py:31) needs:
//   bn_stats      per-column partial sums of (x - shift), (x - shift)^2 with shift = x[0, c]   (local)
//   [all-gather of the 3C+1 floats across ranks -- done by the caller through RCCL]
//   bn_finalize   Chan merge of the per-rank partials -> mean, invstd, running-stat update
//   bn_act_fwd    y = act(gamma * (x-mean)*invstd + beta (+ add))
//   bn_bwd_reduce per-column sums of dz and dz*xhat                                            (local)
//   [all-reduce of the 2C floats]
//   bn_bwd_apply  dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)), dadd = dz
#include "prims.h"
#include "avsr_hip.h"

namespace {

constexpr int BN_THREADS = 256;

AVSR_DEV float act_fwd(float z, int act) { return act == 1 ? avsr_silu(z) : z; }
AVSR_DEV float act_grad(float z, int act) {
    if (act != 1) return 1.f;
    const float s = avsr_sigmoid(z);
    return s * (1.f + z * (1.f - s));
}

// Column reduction skeleton: thread = (column chunk of 8, row lane); MODE 0: stats, MODE 1: backward sums
template <class T, int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_colreduce_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ add, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ out /* MODE0: [3][C] shift,s1,s2 ; MODE1: [2][C] sum dz, sum dz*xhat */, long rows, int C,
    int CL, int rows_per_block, int act) {
    __shared__ float red[BN_THREADS * 16];
    const int cv = C >> 3;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL, RL = BN_THREADS / CL;
    const int cc = blockIdx.x * CL + cl;
    float a[8], b[8], sh[8], mu[8], is[8], ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; e++) a[e] = b[e] = sh[e] = mu[e] = is[e] = ga[e] = be[e] = 0.f;
    if (cc < cv) {
        if (MODE == 0) load8(x + cc * 8, sh);
        else {
            load8(mean + cc * 8, mu);
            load8(invstd + cc * 8, is);
            load8(gamma + cc * 8, ga);
            load8(beta + cc * 8, be);
        }
        constexpr int U = 4;  // independent rows in flight per thread (memory-level parallelism; 8 was measured no faster)
        for (long r0 = (long)blockIdx.y * rows_per_block; r0 < rows; r0 += (long)gridDim.y * rows_per_block) {
        const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
        for (long rb = r0 + rl; rb < r1; rb += (long)U * RL) {
            float v[U][8], g[U][8], ad[U][8];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const long r = rb + (long)u * RL;
                if (r < r1) {
                    load8(x + r * C + cc * 8, v[u]);
                    if (MODE == 1) {
                        load8(dy + r * C + cc * 8, g[u]);
                        if (add) load8(add + r * C + cc * 8, ad[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const long r = rb + (long)u * RL;
                if (r >= r1) continue;
                if (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float d = v[u][e] - sh[e];
                        a[e] += d;
                        b[e] += d * d;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float xh = (v[u][e] - mu[e]) * is[e];
                        const float z = xh * ga[e] + be[e] + (add ? ad[u][e] : 0.f);
                        const float dz = g[u][e] * act_grad(z, act);
                        a[e] += dz;
                        b[e] += dz * xh;
                    }
                }
            }
        }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
        red[threadIdx.x * 16 + e] = a[e];
        red[threadIdx.x * 16 + 8 + e] = b[e];
    }
    __syncthreads();
    if (rl == 0 && cc < cv) {
        for (int q = 1; q < RL; q++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                a[e] += red[(q * CL + cl) * 16 + e];
                b[e] += red[(q * CL + cl) * 16 + 8 + e];
            }
        // per-block partials (no atomics: thousands of blocks hitting the same two cache lines serialise in L2)
        float* part = out + (long)blockIdx.y * 2 * C;
        store8(part + cc * 8, a);
        store8(part + C + cc * 8, b);
    }
}

// dst[base + j][c] = sum_p part[p][j][c], j = 0,1 ; MODE 0 also records the shift row x[0, c].
// block = 16 consecutive columns x 16 partial lanes (short dependent chains), combined in LDS.
template <class T>
__global__ __launch_bounds__(256) void bn_partial_sum_kernel(const float* __restrict__ part, int nparts, int C,
                                                             float* __restrict__ dst, int base, const T* __restrict__ x0,
                                                             float* __restrict__ count_out, float count) {
    __shared__ float red[16][17];
    if (count_out && blockIdx.x == 0 && threadIdx.x == 0) *count_out = count;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + tx;  // over 2*C
    // four independent partial sums per thread: the loop is a chain of L2 / HBM round trips otherwise (8 us for 512 partials)
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < 2 * C) {
        int p = ty;
        for (; p + 48 < nparts; p += 64) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = part[(long)(p + 16 * u) * 2 * C + i];
#pragma unroll
            for (int u = 0; u < 4; u++) s4[u] += v[u];
        }
        for (; p < nparts; p += 16) s4[0] += part[(long)p * 2 * C + i];
    }
    const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q++) t += red[q][tx];
        dst[(long)base * C + i] = t;
        if (x0 && i < C) dst[i] = Elem<T>::ld(x0 + i);
    }
}

// Single-rank statistics in one step: sums the MODE 0 partials [nparts][2][C] (as bn_partial_sum_kernel) and finishes
// the channel (as bn_finalize_kernel with one rank) -- one launch and one dependent round trip less per BatchNorm.
template <class T>
__global__ __launch_bounds__(256) void bn_partial_finalize_kernel(
    const float* __restrict__ part, int nparts, int C, const T* __restrict__ x0, float n, float eps, float momentum,
    float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ running_mean,
    float* __restrict__ running_var, int64_t* __restrict__ num_batches_tracked) {
    __shared__ float red1[16][17], red2[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tx;
    float a4[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        int p = ty;
        for (; p + 48 < nparts; p += 64) {  // eight independent loads in flight per thread
            float va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                va[u] = part[(long)(p + 16 * u) * 2 * C + c];
                vb[u] = part[(long)(p + 16 * u) * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a4[u] += va[u];
                b4[u] += vb[u];
            }
        }
        for (; p < nparts; p += 16) {
            a4[0] += part[(long)p * 2 * C + c];
            b4[0] += part[(long)p * 2 * C + C + c];
        }
    }
    const float a = (a4[0] + a4[1]) + (a4[2] + a4[3]), b = (b4[0] + b4[1]) + (b4[2] + b4[3]);
    red1[ty][tx] = a;
    red2[ty][tx] = b;
    __syncthreads();
    if (ty != 0 || c >= C) return;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        s1 += (double)red1[q][tx];
        s2 += (double)red2[q][tx];
    }
    const double nn = (double)n;
    const double mean = (double)Elem<T>::ld(x0 + c) + s1 / nn;
    const double m2 = fmax(s2 - s1 * s1 / nn, 0.0);  // (f32 partials of a channel with |mean| >> std may cancel below zero)
    const double var = m2 / nn;
    mean_out[c] = (float)mean;
    invstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = nn > 1.0 ? m2 / (nn - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

// stats: rank w at stats + w*ss: [3][C] (shift, s1, s2); count of rank w at counts[w*cs]; one thread per channel
__global__ void bn_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ counts, int W, int C,
                                   long ss, long cs, float eps, float momentum, float* __restrict__ mean_out,
                                   float* __restrict__ invstd_out, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, int64_t* __restrict__ num_batches_tracked,
                                   float* __restrict__ n_total_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;  // nn.BatchNorm's batch counter, kept on the device
    double n_tot = 0.0, mean = 0.0;
    for (int w = 0; w < W; w++) {
        const double n = counts[w * cs];
        if (n <= 0.0) continue;
        const double mw = (double)stats[w * ss + c] + (double)stats[w * ss + C + c] / n;
        mean += n * mw;
        n_tot += n;
    }
    mean /= n_tot;
    if (c == 0 && n_total_out) *n_total_out = (float)n_tot;
    double m2 = 0.0;
    for (int w = 0; w < W; w++) {
        const double n = counts[w * cs];
        if (n <= 0.0) continue;
        const double s1 = stats[w * ss + C + c], s2 = stats[w * ss + 2 * C + c];
        const double mw = (double)stats[w * ss + c] + s1 / n;
        m2 += fmax(s2 - s1 * s1 / n, 0.0) + n * (mw - mean) * (mw - mean);
    }
    const double var = m2 / n_tot;
    mean_out[c] = (float)mean;
    invstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = n_tot > 1.0 ? m2 / (n_tot - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

__global__ void bn_eval_params_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      float eps, int C, float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean_out[c] = running_mean[c];
    invstd_out[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// TA / TY: storage types of the residual input and of the output (sp8_t: the split8 layout the split-plane convolution reads
// without a conversion pass -- prims.h)
template <class T, class TA = T, class TY = T>
__global__ __launch_bounds__(BN_THREADS) void bn_act_fwd_kernel(const T* __restrict__ x, const TA* __restrict__ add,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, TY* __restrict__ y,
                                                                long rows, int C, int act, bf16_t* __restrict__ y2 = nullptr) {
    const int cv = C >> 3;
    const long nvec = rows * cv;
    for (long i = (long)blockIdx.x * BN_THREADS + threadIdx.x; i < nvec; i += (long)gridDim.x * BN_THREADS) {
        const int c = (int)(i % cv) * 8;
        float v[8], mu[8], is[8], ga[8], be[8], ad[8], o[8];
        load8(x + i * 8, v);
        load8(mean + c, mu);
        load8(invstd + c, is);
        load8(gamma + c, ga);
        load8(beta + c, be);
        if (add) load8(add + i * 8, ad);
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = act_fwd((v[e] - mu[e]) * is[e] * ga[e] + be[e] + (add ? ad[e] : 0.f), act);
        store8(y + i * 8, o);
        if (y2) store8(y2 + i * 8, o);  // bf16 twin (hpf mode)
    }
}

// The 3x3 / stride 2 / pad 1 case (the video stem) of bn_act_pool_fwd_kernel below, straight-line: all nine window loads
// are issued up front from clamped addresses (out-of-image taps are masked afterwards), so a thread has 9 x 16 B in flight
// instead of one load per loop trip.  (A 2 x 2-windows-per-thread variant -- 25 taps for four outputs -- was measured 2.5x
// SLOWER: 96 accumulator registers per thread on top of the taps.)  Also records xsel = the RAW input at the arg-max: the backward reduce pass then
// runs on the pooled tensors alone (sum over pooled outputs of dpool * act'(z(xsel)) == sum over pixels of dz).
template <class T, class TY = T>
__global__ __launch_bounds__(BN_THREADS) void bn_act_pool3_fwd_kernel(
    const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, TY* __restrict__ y, uint8_t* __restrict__ idx,
    T* __restrict__ xsel, long N, int H, int W, int C, int OH, int OW, int act, bf16_t* __restrict__ y2 = nullptr,
    bf16_t* __restrict__ xsel2 = nullptr) {
    const int cv = C >> 3;
    const long total = N * OH * OW * cv;
    for (long i = (long)blockIdx.x * BN_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * BN_THREADS) {
        const int c = (int)(i % cv) * 8;
        long r = i / cv;
        const int ow = (int)(r % OW);
        r /= OW;
        const int oh = (int)(r % OH);
        const long n = r / OH;
        float v[9][8];
        unsigned ok = 0;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int ih = oh * 2 + t / 3 - 1, iw = ow * 2 + t % 3 - 1;
            const bool in = ih >= 0 && ih < H && iw >= 0 && iw < W;
            ok |= (in ? 1u : 0u) << t;
            const int ihc = min(max(ih, 0), H - 1), iwc = min(max(iw, 0), W - 1);
            load8(x + ((n * H + ihc) * W + iwc) * C + c, v[t]);
        }
        float mu[8], is[8], ga[8], be[8], m[8], xs[8];
        int am[8];
        load8(mean + c, mu);
        load8(invstd + c, is);
        load8(gamma + c, ga);
        load8(beta + c, be);
        // max over the window of act(bn(x)) needs TWO activations, not nine: bn is affine in x (monotone either way) and every
        // activation here (identity, ReLU, SiLU) is monotone or falls-then-rises, so the maximum over a set of inputs is attained
        // at the set's largest or smallest x.  The SiLU evaluations (v_exp + v_rcp: ~12 VALU each) were what bound this kernel
        // -- 9 per pooled output, 2x its byte time; compares are one instruction.  Ties keep the first tap in window order.
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float xM = -INFINITY, xm = INFINITY;
            int tM = 0, tm = 0;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const bool in = (ok >> t) & 1u;
                const float xv = v[t][e];
                const bool up = in && xv > xM, dn = in && xv < xm;
                xM = up ? xv : xM;
                tM = up ? t : tM;
                xm = dn ? xv : xm;
                tm = dn ? t : tm;
            }
            float aM = act_fwd((xM - mu[e]) * is[e] * ga[e] + be[e], act);
            float am_ = act_fwd((xm - mu[e]) * is[e] * ga[e] + be[e], act);
            if (sizeof(T) == 2) {  // the values the unfused path would have stored
                aM = bf2f(f2bf(aM));
                am_ = bf2f(f2bf(am_));
            }
            const bool lo = am_ > aM || (am_ == aM && tm < tM);
            m[e] = lo ? am_ : aM;
            am[e] = lo ? tm : tM;
            xs[e] = lo ? xm : xM;
        }
        store8(y + i * 8, m);
        if (y2) store8(y2 + i * 8, m);            // bf16 twin of an f32 result (hpf / mixed modes: what the backward pass reads)
        if (xsel) store8(xsel + i * 8, xs);
        if (xsel2) store8(xsel2 + i * 8, xs);     // ... and the arg-max inputs straight in bf16 (only the backward pass reads them)
        uint64_t pk = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) pk |= (uint64_t)am[e] << (8 * e);
        *reinterpret_cast<uint64_t*>(idx + i * 8) = pk;
    }
}

// y[n,oh,ow,c] = max over the KxK / stride S / pad P window of act(bn(x[n,ih,iw,c])), idx = kh*K+kw of the first maximum:
// bn_act_fwd + maxpool_fwd (pool.hip) without the full-resolution activation between them -- the video stem's BN + SiLU
// output is 396 MB per 1600-frame batch, written once and read once only to be pooled 4:1.  Each activated value is
// rounded to the storage type before the comparison, exactly as if it had been stored and re-loaded.
template <class T>
__global__ __launch_bounds__(BN_THREADS) void bn_act_pool_fwd_kernel(
    const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y, uint8_t* __restrict__ idx, long N,
    int H, int W, int C, int OH, int OW, int K, int S, int P, int act) {
    const int cv = C >> 3;
    const long total = N * OH * OW * cv;
    for (long i = (long)blockIdx.x * BN_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * BN_THREADS) {
        const int c = (int)(i % cv) * 8;
        long r = i / cv;
        const int ow = (int)(r % OW);
        r /= OW;
        const int oh = (int)(r % OH);
        const long n = r / OH;
        float mu[8], is[8], ga[8], be[8], m[8];
        uint8_t am[8];
        load8(mean + c, mu);
        load8(invstd + c, is);
        load8(gamma + c, ga);
        load8(beta + c, be);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            m[e] = -INFINITY;
            am[e] = 0;
        }
        for (int kh = 0; kh < K; kh++) {
            const int ih = oh * S + kh - P;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < K; kw++) {
                const int iw = ow * S + kw - P;
                if (iw < 0 || iw >= W) continue;
                float v[8];
                load8(x + ((n * H + ih) * W + iw) * C + c, v);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float a = act_fwd((v[e] - mu[e]) * is[e] * ga[e] + be[e], act);
                    if (sizeof(T) == 2) a = bf2f(f2bf(a));  // the value the unfused path would have stored
                    if (a > m[e]) {
                        m[e] = a;
                        am[e] = (uint8_t)(kh * K + kw);
                    }
                }
            }
        }
        store8(y + i * 8, m);
        uint64_t pk = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) pk |= (uint64_t)am[e] << (8 * e);
        *reinterpret_cast<uint64_t*>(idx + i * 8) = pk;
    }
}

// sums: [2][C] all-reduced (sum dz, sum dz*xhat); n = global row count
template <class T>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ add, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ sums, float inv_n0, const float* __restrict__ n_dev, T* __restrict__ dx,
    T* __restrict__ dadd, long rows, int C, int act) {
    const float inv_n = n_dev ? 1.0f / *n_dev : inv_n0;
    const int cv = C >> 3;
    const long nvec = rows * cv;
    for (long i = (long)blockIdx.x * BN_THREADS + threadIdx.x; i < nvec; i += (long)gridDim.x * BN_THREADS) {
        const int c = (int)(i % cv) * 8;
        float v[8], g[8], mu[8], is[8], ga[8], be[8], ad[8], s1[8], s2[8], o[8], oz[8];
        load8(x + i * 8, v);
        load8(dy + i * 8, g);
        load8(mean + c, mu);
        load8(invstd + c, is);
        load8(gamma + c, ga);
        load8(beta + c, be);
        load8(sums + c, s1);
        load8(sums + C + c, s2);
        if (add) load8(add + i * 8, ad);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float xh = (v[e] - mu[e]) * is[e];
            const float z = xh * ga[e] + be[e] + (add ? ad[e] : 0.f);
            const float dz = g[e] * act_grad(z, act);
            oz[e] = dz;
            o[e] = ga[e] * is[e] * (dz - s1[e] * inv_n - xh * s2[e] * inv_n);
        }
        store8(dx + i * 8, o);
        if (dadd) store8(dadd + i * 8, oz);
    }
}

// ---- single-launch BatchNorm(+activation) for SMALL [rows, C] activations on one rank (the ConvolutionModule's
// BatchNorm1d over rows = B*T <= 2048 frames, conformer_encoder.py:26,33): a block owns 8 channels and keeps its whole
// column slab (rows x 8 values) in registers, so statistics, running-stat update and the normalised output are one pass
// over HBM and ONE launch instead of three (column reduce -> finalize -> apply, ~5 us of launch boundary each on a
// 2.4 MB tensor).  Same arithmetic as that chain: shifted sums in f32, the channel finished in double.
constexpr int BNS_THREADS = 512, BNS_R = 4;  // rows <= 2048

AVSR_DEV void bns_block_sum16(float (&a)[8], float (&b)[8], float* red /* [8 waves][16] */) {
#pragma unroll
    for (int e = 0; e < 8; e++) {
        a[e] = wave_sum(a[e]);
        b[e] = wave_sum(b[e]);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            red[w * 16 + e] = a[e];
            red[w * 16 + 8 + e] = b[e];
        }
    }
    __syncthreads();
}

template <class T, class TO = T>
__global__ __launch_bounds__(BNS_THREADS) void bn_small_fwd_kernel(
    const T* __restrict__ x, int rows, int C, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
    int64_t* __restrict__ num_batches_tracked, int act, TO* __restrict__ y, float* __restrict__ mean_out,
    float* __restrict__ invstd_out, bf16_t* __restrict__ y2 = nullptr) {
    __shared__ float red[8 * 16];
    __shared__ float mu_s[8], is_s[8];
    const int c0 = blockIdx.x * 8;
    float v[BNS_R][8], sh[8], a[8], b[8];
    load8(x + c0, sh);
#pragma unroll
    for (int e = 0; e < 8; e++) a[e] = b[e] = 0.f;
#pragma unroll
    for (int u = 0; u < BNS_R; u++) {
        const int r = threadIdx.x + u * BNS_THREADS;
        if (r < rows) load8(x + (long)r * C + c0, v[u]);
    }
#pragma unroll
    for (int u = 0; u < BNS_R; u++) {
        const int r = threadIdx.x + u * BNS_THREADS;
        if (r >= rows) continue;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float d = v[u][e] - sh[e];
            a[e] += d;
            b[e] += d * d;
        }
    }
    bns_block_sum16(a, b, red);
    if (threadIdx.x < 8) {
        const int e = threadIdx.x, c = c0 + e;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < BNS_THREADS / 64; w++) {
            s1 += (double)red[w * 16 + e];
            s2 += (double)red[w * 16 + 8 + e];
        }
        const double nn = (double)rows;
        const double mean = (double)Elem<T>::ld(x + c) + s1 / nn;
        const double m2 = fmax(s2 - s1 * s1 / nn, 0.0);  // (f32 partials of a channel with |mean| >> std may cancel below zero)
        const double var = m2 / nn;
        const float mf = (float)mean, isf = (float)(1.0 / sqrt(var + (double)eps));
        mean_out[c] = mf;
        invstd_out[c] = isf;
        mu_s[e] = mf;
        is_s[e] = isf;
        if (running_mean) {
            const double unbiased = nn > 1.0 ? m2 / (nn - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
        if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    }
    __syncthreads();
    float ga[8], be[8];
    load8(gamma + c0, ga);
    load8(beta + c0, be);
#pragma unroll
    for (int u = 0; u < BNS_R; u++) {
        const int r = threadIdx.x + u * BNS_THREADS;
        if (r >= rows) continue;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = act_fwd((v[u][e] - mu_s[e]) * is_s[e] * ga[e] + be[e], act);
        store8(y + (long)r * C + c0, o);
        if (y2) store8(y2 + (long)r * C + c0, o);  // bf16 twin (hpf mode)
    }
}

// backward of the same: dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)), dgamma = sum dz*xhat, dbeta = sum dz,
// dz = dy * act'(z) -- reduce and apply in one launch (the slab of x and dy stays in registers between them)
template <class T>
__global__ __launch_bounds__(BNS_THREADS) void bn_small_bwd_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, int rows, int C, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta, int act,
    T* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[8 * 16];
    __shared__ float s1_s[8], s2_s[8];
    const int c0 = blockIdx.x * 8;
    float xh[BNS_R][8], dz[BNS_R][8], mu[8], is[8], ga[8], be[8], a[8], b[8];
    load8(mean + c0, mu);
    load8(invstd + c0, is);
    load8(gamma + c0, ga);
    load8(beta + c0, be);
#pragma unroll
    for (int e = 0; e < 8; e++) a[e] = b[e] = 0.f;
#pragma unroll
    for (int u = 0; u < BNS_R; u++) {
        const int r = threadIdx.x + u * BNS_THREADS;
        if (r < rows) {
            load8(x + (long)r * C + c0, xh[u]);
            load8(dy + (long)r * C + c0, dz[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < BNS_R; u++) {
        const int r = threadIdx.x + u * BNS_THREADS;
        if (r >= rows) continue;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float h = (xh[u][e] - mu[e]) * is[e];
            const float z = h * ga[e] + be[e];
            const float d = dz[u][e] * act_grad(z, act);
            xh[u][e] = h;
            dz[u][e] = d;
            a[e] += d;
            b[e] += d * h;
        }
    }
    bns_block_sum16(a, b, red);
    if (threadIdx.x < 8) {
        const int e = threadIdx.x;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < BNS_THREADS / 64; w++) {
            s1 += red[w * 16 + e];
            s2 += red[w * 16 + 8 + e];
        }
        s1_s[e] = s1;
        s2_s[e] = s2;
        dbeta[c0 + e] = s1;
        dgamma[c0 + e] = s2;
    }
    __syncthreads();
    const float inv_n = 1.0f / (float)rows;
#pragma unroll
    for (int u = 0; u < BNS_R; u++) {
        const int r = threadIdx.x + u * BNS_THREADS;
        if (r >= rows) continue;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = ga[e] * is[e] * (dz[u][e] - s1_s[e] * inv_n - xh[u][e] * s2_s[e] * inv_n);
        store8(dx + (long)r * C + c0, o);
    }
}

// ---- backward of maxpool(act(bn(x))) without the full-resolution gradient tensor: the gradient of the activation at
// (n, ih, iw) is gathered from the pooled gradient -- sum over the (at most 4 for K=3, S=2) windows that contain the pixel
// and whose recorded argmax is that pixel -- wherever the two BatchNorm backward passes need it (pool.hip maxpool_bwd
// would write it to HBM once, 396 MB for the video stem, and both passes would read it back).
template <class T>
AVSR_DEV void pool_grad8(const uint8_t* __restrict__ idx, const T* __restrict__ dpool, long n, int ih, int iw, int c, int C,
                         int OH, int OW, int K, int S, int P, float* g) {
#pragma unroll
    for (int e = 0; e < 8; e++) g[e] = 0.f;
    const int oh_lo = max(0, (ih + P - K + 1 + S - 1) / S), oh_hi = min(OH - 1, (ih + P) / S);
    const int ow_lo = max(0, (iw + P - K + 1 + S - 1) / S), ow_hi = min(OW - 1, (iw + P) / S);
    for (int oh = oh_lo; oh <= oh_hi; oh++)
        for (int ow = ow_lo; ow <= ow_hi; ow++) {
            const int me = (ih - (oh * S - P)) * K + (iw - (ow * S - P));
            const long o = ((n * OH + oh) * OW + ow) * C + c;
            const uint64_t pk = *reinterpret_cast<const uint64_t*>(idx + o);
            float d[8];
            load8(dpool + o, d);
#pragma unroll
            for (int e = 0; e < 8; e++)
                if ((int)((pk >> (8 * e)) & 0xff) == me) g[e] += d[e];
        }
}

// partial sums (sum dz, sum dz * xhat) per block, as bn_colreduce_kernel<T, 1>; rows are the N*H*W pixels
template <class T>
__global__ __launch_bounds__(BN_THREADS) void bn_pool_bwd_reduce_kernel(
    const T* __restrict__ x, const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ out, long rows, int H, int W, int C, int OH, int OW, int K, int S, int P, int CL,
    int rows_per_block, int act) {
    __shared__ float red[BN_THREADS * 16];
    const int cv = C >> 3;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL, RL = BN_THREADS / CL;
    const int cc = blockIdx.x * CL + cl;
    float a[8], b[8], mu[8], is[8], ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; e++) a[e] = b[e] = mu[e] = is[e] = ga[e] = be[e] = 0.f;
    if (cc < cv) {
        load8(mean + cc * 8, mu);
        load8(invstd + cc * 8, is);
        load8(gamma + cc * 8, ga);
        load8(beta + cc * 8, be);
        for (long r0 = (long)blockIdx.y * rows_per_block; r0 < rows; r0 += (long)gridDim.y * rows_per_block) {
            const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
            for (long r = r0 + rl; r < r1; r += RL) {
                const int iw = (int)(r % W);
                const long q = r / W;
                const int ih = (int)(q % H);
                float v[8], g[8];
                load8(x + r * C + cc * 8, v);
                pool_grad8<T>(idx, dpool, q / H, ih, iw, cc * 8, C, OH, OW, K, S, P, g);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float xh = (v[e] - mu[e]) * is[e];
                    const float dz = g[e] * act_grad(xh * ga[e] + be[e], act);
                    a[e] += dz;
                    b[e] += dz * xh;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
        red[threadIdx.x * 16 + e] = a[e];
        red[threadIdx.x * 16 + 8 + e] = b[e];
    }
    __syncthreads();
    if (rl == 0 && cc < cv) {
        for (int q = 1; q < RL; q++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                a[e] += red[(q * CL + cl) * 16 + e];
                b[e] += red[(q * CL + cl) * 16 + 8 + e];
            }
        float* part = out + (long)blockIdx.y * 2 * C;
        store8(part + cc * 8, a);
        store8(part + C + cc * 8, b);
    }
}

// dx = gamma * invstd * (dz - sum_dz / n - xhat * sum_dz_xhat / n), dz from the pooled gradient, as bn_bwd_apply_kernel
template <class T>
__global__ __launch_bounds__(BN_THREADS) void bn_pool_bwd_apply_kernel(
    const T* __restrict__ x, const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ sums, float inv_n0, const float* __restrict__ n_dev, T* __restrict__ dx, long rows, int H,
    int W, int C, int OH, int OW, int K, int S, int P, int act) {
    const float inv_n = n_dev ? 1.0f / *n_dev : inv_n0;
    const int cv = C >> 3;
    const long nvec = rows * cv;
    for (long i = (long)blockIdx.x * BN_THREADS + threadIdx.x; i < nvec; i += (long)gridDim.x * BN_THREADS) {
        const int c = (int)(i % cv) * 8;
        const long r = i / cv;
        const int iw = (int)(r % W);
        const long q = r / W;
        const int ih = (int)(q % H);
        float v[8], g[8], mu[8], is[8], ga[8], be[8], s1[8], s2[8], o[8];
        load8(x + i * 8, v);
        pool_grad8<T>(idx, dpool, q / H, ih, iw, c, C, OH, OW, K, S, P, g);
        load8(mean + c, mu);
        load8(invstd + c, is);
        load8(gamma + c, ga);
        load8(beta + c, be);
        load8(sums + c, s1);
        load8(sums + C + c, s2);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float xh = (v[e] - mu[e]) * is[e];
            const float dz = g[e] * act_grad(xh * ga[e] + be[e], act);
            o[e] = ga[e] * is[e] * (dz - s1[e] * inv_n - xh * s2[e] * inv_n);
        }
        store8(dx + i * 8, o);
    }
}

// 3x3 / stride 2 / pad 1 case of bn_pool_bwd_apply_kernel, one thread per 2 x 2 PIXEL QUAD (rows 2a, 2a+1; columns 2b,
// 2b+1) and 8-channel chunk.  The quad's pixels lie in the four windows (a, b), (a, b+1), (a+1, b), (a+1, b+1) only --
// pixel (2a, 2b) is tap (1,1) of window (a, b); (2a, 2b+1) is tap (1,2) of (a, b) and (1,0) of (a, b+1); (2a+1, 2b) is tap
// (2,1) of (a, b) and (0,1) of (a+1, b); (2a+1, 2b+1) is tap (2,2), (2,0), (0,2), (0,0) of the four -- so a thread loads
// four (idx, dpool) pairs for four pixels (a pixel-per-thread gather loads up to four pairs per PIXEL) and all twelve
// loads of a thread are in flight together; out-of-range windows / pixels are clamped and masked.
template <class T>
__global__ __launch_bounds__(BN_THREADS) void bn_pool3_bwd_apply_kernel(
    const T* __restrict__ x, const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ sums, float inv_n0, const float* __restrict__ n_dev, T* __restrict__ dx, long N, int H,
    int W, int C, int OH, int OW, int act) {
    const float inv_n = n_dev ? 1.0f / *n_dev : inv_n0;
    const int cv = C >> 3, QH = (H + 1) >> 1, QW = (W + 1) >> 1;
    const long nq = N * QH * QW * cv;
    for (long i = (long)blockIdx.x * BN_THREADS + threadIdx.x; i < nq; i += (long)gridDim.x * BN_THREADS) {
        const int c = (int)(i % cv) * 8;
        long r = i / cv;
        const int qb = (int)(r % QW);
        r /= QW;
        const int qa = (int)(r % QH);
        const long n = r / QH;
        // the four windows
        uint64_t pk[4];
        float d[4][8];
        bool wok[4];
#pragma unroll
        for (int wa = 0; wa < 2; wa++)
#pragma unroll
            for (int wb = 0; wb < 2; wb++) {
                const int oh = qa + wa, ow = qb + wb;
                wok[wa * 2 + wb] = oh < OH && ow < OW;
                const long o = ((n * OH + min(oh, OH - 1)) * OW + min(ow, OW - 1)) * C + c;
                pk[wa * 2 + wb] = *reinterpret_cast<const uint64_t*>(idx + o);
                load8(dpool + o, d[wa * 2 + wb]);
            }
        // the four pixels
        float v[4][8];
        bool pok[4];
        long poff[4];
#pragma unroll
        for (int pa = 0; pa < 2; pa++)
#pragma unroll
            for (int pb = 0; pb < 2; pb++) {
                const int ih = 2 * qa + pa, iw = 2 * qb + pb;
                pok[pa * 2 + pb] = ih < H && iw < W;
                poff[pa * 2 + pb] = ((n * H + min(ih, H - 1)) * W + min(iw, W - 1)) * C + c;
                load8(x + poff[pa * 2 + pb], v[pa * 2 + pb]);
            }
        float mu[8], is[8], ga[8], be[8], s1[8], s2[8];
        load8(mean + c, mu);
        load8(invstd + c, is);
        load8(gamma + c, ga);
        load8(beta + c, be);
        load8(sums + c, s1);
        load8(sums + C + c, s2);
        // (window, tap) pairs of every pixel: tap index kh * 3 + kw
        constexpr int NW[4] = {1, 2, 2, 4};
        constexpr int WIN[4][4] = {{0, 0, 0, 0}, {0, 1, 0, 0}, {0, 2, 0, 0}, {0, 1, 2, 3}};
        constexpr int TAP[4][4] = {{4, 0, 0, 0}, {5, 3, 0, 0}, {7, 1, 0, 0}, {8, 6, 2, 0}};
#pragma unroll
        for (int p = 0; p < 4; p++) {
            float g[8], o[8];
#pragma unroll
            for (int e = 0; e < 8; e++) g[e] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (q >= NW[p]) continue;
                const int wi = WIN[p][q], tap = TAP[p][q];
#pragma unroll
                for (int e = 0; e < 8; e++)
                    g[e] += (wok[wi] && (int)((pk[wi] >> (8 * e)) & 0xff) == tap) ? d[wi][e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float xh = (v[p][e] - mu[e]) * is[e];
                const float dz = g[e] * act_grad(xh * ga[e] + be[e], act);
                o[e] = ga[e] * is[e] * (dz - s1[e] * inv_n - xh * s2[e] * inv_n);
            }
            if (pok[p]) store8(dx + poff[p], o);
        }
    }
}

static inline int pick_cl(int cv) { return cv >= 32 ? 32 : (cv >= 16 ? 16 : 8); }
static inline int ew_grid(long nvec) {
    long b = (nvec + BN_THREADS - 1) / BN_THREADS;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

static inline int bn_parts(long rows, int rpb, int gx) {
    long need = (rows + rpb - 1) / rpb;
    long cap = 512 / (gx < 1 ? 1 : gx);  // (1024 was measured no faster: 14.2 -> 14.6 us, and the second stage slower)
    if (cap < 1) cap = 1;
    return (int)(need < cap ? need : cap);
}

extern "C" int64_t avsr_bn_workspace_floats(int C) { return (int64_t)1024 * 2 * C; }

// stats: [3][C] f32 (overwritten); workspace: avsr_bn_workspace_floats(C) floats; *count_out = rows when != NULL
extern "C" int avsr_bn_stats(const void* x, int dtype, float* stats, float* workspace, int64_t rows, int C,
                             float* count_out, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "batchnorm: C must be a multiple of 8");
    if (rows <= 0) return 0;
    const int cv = C >> 3, CL = pick_cl(cv);
    const int rpb = 128 * (BN_THREADS / CL) / 8;
    const int gx = (cv + CL - 1) / CL;
    const int parts = bn_parts(rows, rpb, gx);
    dim3 grid(gx, parts), block(BN_THREADS);
    dim3 g2((2 * C + 15) / 16);
    if (dtype == 0) {
        AVSR_LAUNCH((bn_colreduce_kernel<float, 0>), grid, block, 0, stream, (const float*)x, (const float*)nullptr,
                    (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                    (const float*)nullptr, workspace, (long)rows, C, CL, rpb, 0);
        AVSR_LAUNCH((bn_partial_sum_kernel<float>), g2, dim3(256), 0, stream, (const float*)workspace, parts, C, stats, 1, (const float*)x,
                    count_out, (float)rows);
    } else if (dtype == 2) {
        AVSR_LAUNCH((bn_colreduce_kernel<f16_t, 0>), grid, block, 0, stream, (const f16_t*)x, (const f16_t*)nullptr,
                    (const f16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                    (const float*)nullptr, workspace, (long)rows, C, CL, rpb, 0);
        AVSR_LAUNCH((bn_partial_sum_kernel<f16_t>), g2, dim3(256), 0, stream, (const float*)workspace, parts, C, stats, 1, (const f16_t*)x,
                    count_out, (float)rows);
    } else {
        AVSR_LAUNCH((bn_colreduce_kernel<bf16_t, 0>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)nullptr,
                    (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                    (const float*)nullptr, workspace, (long)rows, C, CL, rpb, 0);
        AVSR_LAUNCH((bn_partial_sum_kernel<bf16_t>), g2, dim3(256), 0, stream, (const float*)workspace, parts, C, stats, 1, (const bf16_t*)x,
                    count_out, (float)rows);
    }
    AVSR_CHECK_LAUNCH("bn_stats");
    return 0;
}

// avsr_bn_stats + avsr_bn_finalize for one rank (world == 1) in two launches instead of three
extern "C" int avsr_bn_stats_finalize(const void* x, int dtype, float* workspace, int64_t rows, int C, float eps,
                                      float momentum, float* mean, float* invstd, float* running_mean,
                                      float* running_var, int64_t* num_batches_tracked, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "batchnorm: C must be a multiple of 8");
    AVSR_REQUIRE(rows > 0, "batchnorm: no rows");
    const int cv = C >> 3, CL = pick_cl(cv);
    const int rpb = 128 * (BN_THREADS / CL) / 8;
    const int gx = (cv + CL - 1) / CL;
    const int parts = bn_parts(rows, rpb, gx);
    dim3 grid(gx, parts), block(BN_THREADS);
    dim3 g2((C + 15) / 16);
    if (dtype == 0) {
        AVSR_LAUNCH((bn_colreduce_kernel<float, 0>), grid, block, 0, stream, (const float*)x, (const float*)nullptr,
                    (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                    (const float*)nullptr, workspace, (long)rows, C, CL, rpb, 0);
        AVSR_LAUNCH((bn_partial_finalize_kernel<float>), g2, dim3(256), 0, stream, (const float*)workspace, parts, C,
                    (const float*)x, (float)rows, eps, momentum, mean, invstd, running_mean, running_var,
                    num_batches_tracked);
    } else if (dtype == 2) {
        AVSR_LAUNCH((bn_colreduce_kernel<f16_t, 0>), grid, block, 0, stream, (const f16_t*)x, (const f16_t*)nullptr,
                    (const f16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                    (const float*)nullptr, workspace, (long)rows, C, CL, rpb, 0);
        AVSR_LAUNCH((bn_partial_finalize_kernel<f16_t>), g2, dim3(256), 0, stream, (const float*)workspace, parts, C,
                    (const f16_t*)x, (float)rows, eps, momentum, mean, invstd, running_mean, running_var,
                    num_batches_tracked);
    } else {
        AVSR_LAUNCH((bn_colreduce_kernel<bf16_t, 0>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)nullptr,
                    (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                    (const float*)nullptr, workspace, (long)rows, C, CL, rpb, 0);
        AVSR_LAUNCH((bn_partial_finalize_kernel<bf16_t>), g2, dim3(256), 0, stream, (const float*)workspace, parts, C,
                    (const bf16_t*)x, (float)rows, eps, momentum, mean, invstd, running_mean, running_var,
                    num_batches_tracked);
    }
    AVSR_CHECK_LAUNCH("bn_stats_finalize");
    return 0;
}

// Statistics left behind by a producer's epilogue (avsr_conv2d_f32s_stats): part [ntiles][2][C] UNSHIFTED column sums / sums of
// squares of 128-row tiles.  First 256 blocks fold the tiles into ws [256][2][C] (block j: tiles j, j + 256, ...; plain stores),
// then the shared finalize kernels run on those 256 partials with a zero shift row (zeros: >= C zero floats).
// _finalize_parts: single rank, (mean, invstd) + running statistics; _stats_parts: the [3][C] (+ count) payload of the
// cross-rank merge (avsr_bn_finalize).
constexpr int BN_PART_SLOTS = 256;
__global__ __launch_bounds__(256) void bn_parts_fold_kernel(const float* __restrict__ part, int ntiles, int C2, float* __restrict__ ws) {
    __shared__ float red[256];
    const int lanes_per = 256 / C2 > 0 ? 256 / C2 : 1;  // row lanes when 2 C < 256
    const int col = threadIdx.x % C2, rl = threadIdx.x / C2;
    for (int c0 = 0; c0 < C2; c0 += 256) {  // (2 C <= 256: one trip)
        const int c = c0 + col;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < C2 && rl < lanes_per) {
            int t = blockIdx.x + BN_PART_SLOTS * rl;
            const int step = BN_PART_SLOTS * lanes_per;
            for (; t + 3 * step < ntiles; t += 4 * step) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = part[(size_t)(t + u * step) * C2 + c];
#pragma unroll
                for (int u = 0; u < 4; u++) a[u] += v[u];
            }
            for (; t < ntiles; t += step) a[0] += part[(size_t)t * C2 + c];
        }
        red[threadIdx.x] = (a[0] + a[1]) + (a[2] + a[3]);
        __syncthreads();
        if (rl == 0 && c < C2) {
            float v = 0.f;
            for (int q = 0; q < lanes_per; q++) v += red[q * C2 + col];
            ws[(size_t)blockIdx.x * C2 + c] = v;
        }
        __syncthreads();
    }
}

extern "C" int avsr_bn_finalize_parts(const float* part, int ntiles, int C, const float* zeros, float* ws, int64_t rows, float eps,
                                      float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                                      int64_t* num_batches_tracked, hipStream_t stream) {
    AVSR_REQUIRE(ntiles >= 1 && C >= 1 && rows >= 1 && zeros != nullptr && ws != nullptr, "bn_finalize_parts: bad arguments");
    AVSR_LAUNCH(bn_parts_fold_kernel, dim3(BN_PART_SLOTS), dim3(256), 0, stream, part, ntiles, 2 * C, ws);
    dim3 g2((C + 15) / 16);
    AVSR_LAUNCH((bn_partial_finalize_kernel<float>), g2, dim3(256), 0, stream, (const float*)ws, BN_PART_SLOTS, C, zeros, (float)rows, eps,
                momentum, mean, invstd, running_mean, running_var, num_batches_tracked);
    AVSR_CHECK_LAUNCH("bn_finalize_parts");
    return 0;
}
extern "C" int avsr_bn_stats_parts(const float* part, int ntiles, int C, const float* zeros, float* ws, float* stats, float* count_out,
                                   int64_t rows, hipStream_t stream) {
    AVSR_REQUIRE(ntiles >= 1 && C >= 1 && rows >= 1 && zeros != nullptr && ws != nullptr, "bn_stats_parts: bad arguments");
    AVSR_LAUNCH(bn_parts_fold_kernel, dim3(BN_PART_SLOTS), dim3(256), 0, stream, part, ntiles, 2 * C, ws);
    dim3 g2((2 * C + 15) / 16);
    AVSR_LAUNCH((bn_partial_sum_kernel<float>), g2, dim3(256), 0, stream, (const float*)ws, BN_PART_SLOTS, C, stats, 1, zeros, count_out,
                (float)rows);
    AVSR_CHECK_LAUNCH("bn_stats_parts");
    return 0;
}

extern "C" int avsr_bn_finalize(const float* stats, const float* counts, int world, int C, int64_t stats_stride,
                                int64_t counts_stride, float eps, float momentum, float* mean, float* invstd,
                                float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                float* n_total, hipStream_t stream) {
    dim3 grid((C + 127) / 128), block(128);
    const long ss = stats_stride > 0 ? (long)stats_stride : 3L * C, cs = counts_stride > 0 ? (long)counts_stride : 1L;
    AVSR_LAUNCH(bn_finalize_kernel, grid, block, 0, stream, stats, counts, world, C, ss, cs, eps, momentum, mean, invstd,
                running_mean, running_var, num_batches_tracked, n_total);
    AVSR_CHECK_LAUNCH("bn_finalize");
    return 0;
}

extern "C" int avsr_bn_eval_params(const float* running_mean, const float* running_var, float eps, int C, float* mean,
                                   float* invstd, hipStream_t stream) {
    dim3 grid((C + 127) / 128), block(128);
    AVSR_LAUNCH(bn_eval_params_kernel, grid, block, 0, stream, running_mean, running_var, eps, C, mean, invstd);
    AVSR_CHECK_LAUNCH("bn_eval_params");
    return 0;
}

extern "C" int avsr_bn_act_fwd(const void* x, const void* add, int dtype, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, void* y, int64_t rows, int C, int act,
                               hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "batchnorm: C must be a multiple of 8");
    if (rows <= 0) return 0;
    dim3 grid(ew_grid(rows * (C >> 3))), block(BN_THREADS);
    if (dtype == 0)
        AVSR_LAUNCH((bn_act_fwd_kernel<float>), grid, block, 0, stream, (const float*)x, (const float*)add, mean, invstd,
                    gamma, beta, (float*)y, (long)rows, C, act);
    else
        AVSR_LAUNCH((bn_act_fwd_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)add, mean,
                    invstd, gamma, beta, (bf16_t*)y, (long)rows, C, act);
    AVSR_CHECK_LAUNCH("bn_act_fwd");
    return 0;
}

// f32 in / f32 out + the bf16 twin of the output in one pass (the "hpf" numerical mode).  layout (round 5): bit 0 -- y is written
// in the split8 layout (prims.h sp8_t: what avsr_conv2d_f32s* reads as a pre-split A operand, tile codes 23 - 26); bit 1 -- `add`
// is stored in the split8 layout.
extern "C" int avsr_bn_act_fwd2(const float* x, const void* add, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, void* y, void* y2, int64_t rows, int C, int act, int layout, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "batchnorm: C must be a multiple of 8");
    if (rows <= 0) return 0;
    dim3 grid(ew_grid(rows * (C >> 3))), block(BN_THREADS);
    const bool ys = layout & 1, as = (layout & 2) && add;
    if (ys && as)
        AVSR_LAUNCH((bn_act_fwd_kernel<float, sp8_t, sp8_t>), grid, block, 0, stream, x, (const sp8_t*)add, mean, invstd, gamma, beta,
                    (sp8_t*)y, (long)rows, C, act, (bf16_t*)y2);
    else if (ys)
        AVSR_LAUNCH((bn_act_fwd_kernel<float, float, sp8_t>), grid, block, 0, stream, x, (const float*)add, mean, invstd, gamma, beta,
                    (sp8_t*)y, (long)rows, C, act, (bf16_t*)y2);
    else if (as)
        AVSR_LAUNCH((bn_act_fwd_kernel<float, sp8_t, float>), grid, block, 0, stream, x, (const sp8_t*)add, mean, invstd, gamma, beta,
                    (float*)y, (long)rows, C, act, (bf16_t*)y2);
    else
        AVSR_LAUNCH((bn_act_fwd_kernel<float>), grid, block, 0, stream, x, (const float*)add, mean, invstd, gamma, beta, (float*)y,
                    (long)rows, C, act, (bf16_t*)y2);
    AVSR_CHECK_LAUNCH("bn_act_fwd2");
    return 0;
}

// f16 in / f16 out + the bf16 twin y2 (may be NULL) of the output in one pass (the "mixed" numerical mode)
extern "C" int avsr_bn_act_fwd_h16(const void* x, const void* add, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, void* y, void* y2, int64_t rows, int C, int act, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "batchnorm: C must be a multiple of 8");
    if (rows <= 0) return 0;
    dim3 grid(ew_grid(rows * (C >> 3))), block(BN_THREADS);
    AVSR_LAUNCH((bn_act_fwd_kernel<f16_t>), grid, block, 0, stream, (const f16_t*)x, (const f16_t*)add, mean, invstd, gamma, beta,
                (f16_t*)y, (long)rows, C, act, (bf16_t*)y2);
    AVSR_CHECK_LAUNCH("bn_act_fwd_h16");
    return 0;
}

// y [N][OH][OW][C], idx uint8 [N][OH][OW][C] = maxpool(act(bn(x [N][H][W][C]))) with OH = (H + 2P - K)/S + 1
extern "C" int avsr_bn_act_pool_fwd(const void* x, int dtype, const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, void* y, uint8_t* idx, void* xsel, int64_t N, int H, int W, int C,
                                    int K, int S, int P, int act, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "batchnorm: C must be a multiple of 8");
    AVSR_REQUIRE(idx != nullptr, "bn_act_pool_fwd: idx required");
    const int OH = (H + 2 * P - K) / S + 1, OW = (W + 2 * P - K) / S + 1;
    if (N <= 0) return 0;
    dim3 grid(ew_grid((long)N * OH * OW * (C >> 3))), block(BN_THREADS);
    if (K == 3 && S == 2 && P == 1) {
        if (dtype == 0)
            AVSR_LAUNCH((bn_act_pool3_fwd_kernel<float>), grid, block, 0, stream, (const float*)x, mean, invstd, gamma, beta,
                        (float*)y, idx, (float*)xsel, (long)N, H, W, C, OH, OW, act);
        else
            AVSR_LAUNCH((bn_act_pool3_fwd_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)x, mean, invstd, gamma, beta,
                        (bf16_t*)y, idx, (bf16_t*)xsel, (long)N, H, W, C, OH, OW, act);
        AVSR_CHECK_LAUNCH("bn_act_pool_fwd");
        return 0;
    }
    AVSR_REQUIRE(xsel == nullptr, "bn_act_pool_fwd: xsel is produced by the 3x3 / stride 2 / pad 1 kernel only");
    if (dtype == 0)
        AVSR_LAUNCH((bn_act_pool_fwd_kernel<float>), grid, block, 0, stream, (const float*)x, mean, invstd, gamma, beta,
                    (float*)y, idx, (long)N, H, W, C, OH, OW, K, S, P, act);
    else
        AVSR_LAUNCH((bn_act_pool_fwd_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)x, mean, invstd, gamma, beta,
                    (bf16_t*)y, idx, (long)N, H, W, C, OH, OW, K, S, P, act);
    AVSR_CHECK_LAUNCH("bn_act_pool_fwd");
    return 0;
}

// f32 input / output of the 3x3 / stride 2 / pad 1 case + the bf16 twin y2 of the pooled output and xsel2, the arg-max inputs in bf16
// (either may be NULL), in the same pass -- the hpf / mixed modes' stem: no cast launches over the pooled tensors afterwards
// y_split8: the pooled output is written in the split8 layout (prims.h sp8_t) for a split-plane consumer
extern "C" int avsr_bn_act_pool3_fwd2(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                      void* y, void* y2, uint8_t* idx, void* xsel2, int64_t N, int H, int W, int C, int act,
                                      int y_split8, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "batchnorm: C must be a multiple of 8");
    AVSR_REQUIRE(idx != nullptr, "bn_act_pool3_fwd2: idx required");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    if (N <= 0) return 0;
    dim3 grid(ew_grid((long)N * OH * OW * (C >> 3))), block(BN_THREADS);
    if (y_split8)
        AVSR_LAUNCH((bn_act_pool3_fwd_kernel<float, sp8_t>), grid, block, 0, stream, x, mean, invstd, gamma, beta, (sp8_t*)y, idx,
                    (float*)nullptr, (long)N, H, W, C, OH, OW, act, (bf16_t*)y2, (bf16_t*)xsel2);
    else
        AVSR_LAUNCH((bn_act_pool3_fwd_kernel<float>), grid, block, 0, stream, x, mean, invstd, gamma, beta, (float*)y, idx,
                    (float*)nullptr, (long)N, H, W, C, OH, OW, act, (bf16_t*)y2, (bf16_t*)xsel2);
    AVSR_CHECK_LAUNCH("bn_act_pool3_fwd2");
    return 0;
}

// backward of avsr_bn_act_pool_fwd, first pass: sums [2][C] = (sum dz, sum dz * xhat) over the N*H*W pixels, the gradient
// gathered from dpool [N][OH][OW][C] through idx; workspace: avsr_bn_workspace_floats(C) floats
extern "C" int avsr_bn_pool_bwd_reduce(const void* x, const void* dpool, const uint8_t* idx, int dtype, const float* mean,
                                       const float* invstd, const float* gamma, const float* beta, float* sums,
                                       float* workspace, int64_t N, int H, int W, int C, int K, int S, int P, int act,
                                       hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "batchnorm: C must be a multiple of 8");
    const int OH = (H + 2 * P - K) / S + 1, OW = (W + 2 * P - K) / S + 1;
    const long rows = (long)N * H * W;
    if (rows <= 0) return 0;
    const int cv = C >> 3, CL = pick_cl(cv);
    const int rpb = 128 * (BN_THREADS / CL) / 8;
    const int gx = (cv + CL - 1) / CL;
    const int parts = bn_parts(rows, rpb, gx);
    dim3 grid(gx, parts), block(BN_THREADS);
    dim3 g2((2 * C + 15) / 16);
    if (dtype == 0)
        AVSR_LAUNCH((bn_pool_bwd_reduce_kernel<float>), grid, block, 0, stream, (const float*)x, (const float*)dpool, idx, mean,
                    invstd, gamma, beta, workspace, rows, H, W, C, OH, OW, K, S, P, CL, rpb, act);
    else
        AVSR_LAUNCH((bn_pool_bwd_reduce_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)dpool, idx,
                    mean, invstd, gamma, beta, workspace, rows, H, W, C, OH, OW, K, S, P, CL, rpb, act);
    AVSR_LAUNCH((bn_partial_sum_kernel<float>), g2, dim3(256), 0, stream, (const float*)workspace, parts, C, sums, 0,
                (const float*)nullptr, (float*)nullptr, 0.f);
    AVSR_CHECK_LAUNCH("bn_pool_bwd_reduce");
    return 0;
}

// second pass: dx [N][H][W][C] from the (all-reduced) sums; inv_n / n_dev as avsr_bn_bwd_apply
extern "C" int avsr_bn_pool_bwd_apply(const void* x, const void* dpool, const uint8_t* idx, int dtype, const float* mean,
                                      const float* invstd, const float* gamma, const float* beta, const float* sums,
                                      float inv_n, const float* n_dev, void* dx, int64_t N, int H, int W, int C, int K,
                                      int S, int P, int act, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "batchnorm: C must be a multiple of 8");
    const int OH = (H + 2 * P - K) / S + 1, OW = (W + 2 * P - K) / S + 1;
    const long rows = (long)N * H * W;
    if (rows <= 0) return 0;
    dim3 grid(ew_grid(rows * (C >> 3))), block(BN_THREADS);
    if (K == 3 && S == 2 && P == 1) {
        dim3 gq(ew_grid((long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C >> 3)));  // one thread per 2 x 2 pixel quad and chunk
        if (dtype == 0)
            AVSR_LAUNCH((bn_pool3_bwd_apply_kernel<float>), gq, block, 0, stream, (const float*)x, (const float*)dpool, idx,
                        mean, invstd, gamma, beta, sums, inv_n, n_dev, (float*)dx, (long)N, H, W, C, OH, OW, act);
        else
            AVSR_LAUNCH((bn_pool3_bwd_apply_kernel<bf16_t>), gq, block, 0, stream, (const bf16_t*)x, (const bf16_t*)dpool,
                        idx, mean, invstd, gamma, beta, sums, inv_n, n_dev, (bf16_t*)dx, (long)N, H, W, C, OH, OW, act);
        AVSR_CHECK_LAUNCH("bn_pool_bwd_apply");
        return 0;
    }
    if (dtype == 0)
        AVSR_LAUNCH((bn_pool_bwd_apply_kernel<float>), grid, block, 0, stream, (const float*)x, (const float*)dpool, idx, mean,
                    invstd, gamma, beta, sums, inv_n, n_dev, (float*)dx, rows, H, W, C, OH, OW, K, S, P, act);
    else
        AVSR_LAUNCH((bn_pool_bwd_apply_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)dpool, idx,
                    mean, invstd, gamma, beta, sums, inv_n, n_dev, (bf16_t*)dx, rows, H, W, C, OH, OW, K, S, P, act);
    AVSR_CHECK_LAUNCH("bn_pool_bwd_apply");
    return 0;
}

// sums: [2][C] f32 (overwritten); workspace: avsr_bn_workspace_floats(C) floats
extern "C" int avsr_bn_bwd_reduce(const void* x, const void* dy, const void* add, int dtype, const float* mean,
                                  const float* invstd, const float* gamma, const float* beta, float* sums,
                                  float* workspace, int64_t rows, int C, int act, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "batchnorm: C must be a multiple of 8");
    if (rows <= 0) return 0;
    const int cv = C >> 3, CL = pick_cl(cv);
    const int rpb = 128 * (BN_THREADS / CL) / 8;
    const int gx = (cv + CL - 1) / CL;
    const int parts = bn_parts(rows, rpb, gx);
    dim3 grid(gx, parts), block(BN_THREADS);
    dim3 g2((2 * C + 15) / 16);
    if (dtype == 0)
        AVSR_LAUNCH((bn_colreduce_kernel<float, 1>), grid, block, 0, stream, (const float*)x, (const float*)dy,
                    (const float*)add, mean, invstd, gamma, beta, workspace, (long)rows, C, CL, rpb, act);
    else
        AVSR_LAUNCH((bn_colreduce_kernel<bf16_t, 1>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)dy,
                    (const bf16_t*)add, mean, invstd, gamma, beta, workspace, (long)rows, C, CL, rpb, act);
    AVSR_LAUNCH((bn_partial_sum_kernel<float>), g2, dim3(256), 0, stream, (const float*)workspace, parts, C, sums, 0, (const float*)nullptr,
                (float*)nullptr, 0.f);
    AVSR_CHECK_LAUNCH("bn_bwd_reduce");
    return 0;
}

extern "C" int avsr_bn_bwd_apply(const void* x, const void* dy, const void* add, int dtype, const float* mean,
                                 const float* invstd, const float* gamma, const float* beta, const float* sums,
                                 float inv_n, const float* n_dev, void* dx, void* dadd, int64_t rows, int C,
                                 int act, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "batchnorm: C must be a multiple of 8");
    if (rows <= 0) return 0;
    dim3 grid(ew_grid(rows * (C >> 3))), block(BN_THREADS);
    if (dtype == 0)
        AVSR_LAUNCH((bn_bwd_apply_kernel<float>), grid, block, 0, stream, (const float*)x, (const float*)dy,
                    (const float*)add, mean, invstd, gamma, beta, sums, inv_n, n_dev, (float*)dx, (float*)dadd, (long)rows, C, act);
    else
        AVSR_LAUNCH((bn_bwd_apply_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)dy,
                    (const bf16_t*)add, mean, invstd, gamma, beta, sums, inv_n, n_dev, (bf16_t*)dx, (bf16_t*)dadd, (long)rows, C, act);
    AVSR_CHECK_LAUNCH("bn_bwd_apply");
    return 0;
}

extern "C" int avsr_bn_small_max_rows(void) { return BNS_THREADS * BNS_R; }

extern "C" int avsr_bn_small_fwd(const void* x, int dtype, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                 int act, void* y, float* mean, float* invstd, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "bn_small: C must be a multiple of 8");
    AVSR_REQUIRE(rows >= 1 && rows <= BNS_THREADS * BNS_R, "bn_small: rows out of range (avsr_bn_small_max_rows)");
    dim3 grid(C / 8), block(BNS_THREADS);
    if (dtype == 0)
        AVSR_LAUNCH((bn_small_fwd_kernel<float>), grid, block, 0, stream, (const float*)x, (int)rows, C, gamma, beta, eps, momentum,
                    running_mean, running_var, num_batches_tracked, act, (float*)y, mean, invstd);
    else
        AVSR_LAUNCH((bn_small_fwd_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)x, (int)rows, C, gamma, beta, eps,
                    momentum, running_mean, running_var, num_batches_tracked, act, (bf16_t*)y, mean, invstd);
    AVSR_CHECK_LAUNCH("bn_small_fwd");
    return 0;
}

// f32 input, f32 (y_dtype 0) or f16 (y_dtype 2) output + the bf16 twin y2 of the output in one pass.  f16 output: the mixed mode
// keeps the convolution module's element-wise chain (pointwise-1 output -> GLU -> depthwise conv -> BatchNorm) in f32 and rounds
// to f16 only here, where the next consumer is an MFMA operand.
extern "C" int avsr_bn_small_fwd2(const float* x, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                                  float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                  int act, void* y, int y_dtype, void* y2, float* mean, float* invstd, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "bn_small: C must be a multiple of 8");
    AVSR_REQUIRE(rows >= 1 && rows <= BNS_THREADS * BNS_R, "bn_small: rows out of range (avsr_bn_small_max_rows)");
    AVSR_REQUIRE(y_dtype == 0 || y_dtype == 2, "bn_small_fwd2: output f32 (0) or f16 (2)");
    if (y_dtype == 2)
        AVSR_LAUNCH((bn_small_fwd_kernel<float, f16_t>), dim3(C / 8), dim3(BNS_THREADS), 0, stream, x, (int)rows, C, gamma, beta, eps,
                    momentum, running_mean, running_var, num_batches_tracked, act, (f16_t*)y, mean, invstd, (bf16_t*)y2);
    else
        AVSR_LAUNCH((bn_small_fwd_kernel<float>), dim3(C / 8), dim3(BNS_THREADS), 0, stream, x, (int)rows, C, gamma, beta, eps, momentum,
                    running_mean, running_var, num_batches_tracked, act, (float*)y, mean, invstd, (bf16_t*)y2);
    AVSR_CHECK_LAUNCH("bn_small_fwd2");
    return 0;
}

// f16 input / output + the bf16 twin y2 (may be NULL) of the output
extern "C" int avsr_bn_small_fwd_h16(const void* x, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                                     float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                     int act, void* y, void* y2, float* mean, float* invstd, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "bn_small: C must be a multiple of 8");
    AVSR_REQUIRE(rows >= 1 && rows <= BNS_THREADS * BNS_R, "bn_small: rows out of range (avsr_bn_small_max_rows)");
    AVSR_LAUNCH((bn_small_fwd_kernel<f16_t>), dim3(C / 8), dim3(BNS_THREADS), 0, stream, (const f16_t*)x, (int)rows, C, gamma, beta,
                eps, momentum, running_mean, running_var, num_batches_tracked, act, (f16_t*)y, mean, invstd, (bf16_t*)y2);
    AVSR_CHECK_LAUNCH("bn_small_fwd_h16");
    return 0;
}

extern "C" int avsr_bn_small_bwd(const void* x, const void* dy, int dtype, int64_t rows, int C, const float* mean,
                                 const float* invstd, const float* gamma, const float* beta, int act, void* dx, float* dgamma,
                                 float* dbeta, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "bn_small: C must be a multiple of 8");
    AVSR_REQUIRE(rows >= 1 && rows <= BNS_THREADS * BNS_R, "bn_small: rows out of range (avsr_bn_small_max_rows)");
    dim3 grid(C / 8), block(BNS_THREADS);
    if (dtype == 0)
        AVSR_LAUNCH((bn_small_bwd_kernel<float>), grid, block, 0, stream, (const float*)x, (const float*)dy, (int)rows, C, mean,
                    invstd, gamma, beta, act, (float*)dx, dgamma, dbeta);
    else
        AVSR_LAUNCH((bn_small_bwd_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)dy, (int)rows, C,
                    mean, invstd, gamma, beta, act, (bf16_t*)dx, dgamma, dbeta);
    AVSR_CHECK_LAUNCH("bn_small_bwd");
    return 0;
}
