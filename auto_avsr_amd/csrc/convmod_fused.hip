// convmod_fused.hip (round 6) -- the element-wise middle of the Conformer ConvolutionModule as ONE launch each way on a single rank:
//     forward :  a [B*T, 2C] (pointwise-1 output)  ->  GLU  ->  depthwise Conv1d(K)  ->  BatchNorm1d (batch statistics over every
//                frame, running-stat update)  ->  SiLU  ->  s [B*T, C]
//     backward:  ds  ->  BatchNorm + SiLU backward  ->  depthwise weight / bias gradient + data gradient  ->  GLU backward  ->  da
// Replaces conformer_encoder.py:32-34 (`glu`, `depthwise_conv`, `norm`, `activation`) under autograd.  Until round 5 these were the
// launches dwconv (+ GLU) | bn_small_fwd forward and bn_small_bwd | dwconv_wgrad | dwconv (flip, + GLU backward) backward: five
// dependent launches of 10 - 23 us per layer on a 1600 x 768 activation -- latency, not bytes.  Every one of
// these operations is LOCAL TO A CHANNEL: a block that owns 8 channels and every row (B*T <= 2048 frames, as bn_small_*) can do
// the whole chain -- the GLU output (and, backward, the BatchNorm's input gradient) of its channels sits in LDS as a [rows][8] f32
// slab, the time stencil of the depthwise convolution reads it from there, the BatchNorm sums are block-local.
// MEASURED (round 6, tools/microbench_convmod.py -> profiles/r6_microbench_convmod.txt): SLOWER than the launches it merges -- forward
// 29.7 us vs 22.5 us, backward 112 us vs 38.4 us per layer at 1600 x 768, K = 31; replayed step 23.76 vs 22.82 ms.  Owning every
// frame of a channel group caps the grid at C / 8 = 96 blocks: the three 31-tap stencils (forward, data gradient, weight gradient)
// read their LDS slab through 96 CUs' LDS ports, where the separate kernels spread the same reads over 576 - 800 blocks on 256 CUs,
// and the launch boundaries saved (3 x ~2 us) do not pay for that.  Kept as an opt-in (AVSR_CONVMOD_FUSED=1) with its parity test
// (tests/test_convmod_kernels.py::test_convmod_fused_middle); the default path keeps the separate launches.
// Arithmetic: exactly the sequence of the launches it replaces (same rounding points -- a bf16 tensor the old sequence stored is
// re-rounded here --, same summation order in the BatchNorm; the depthwise weight gradient sums its rows in another order).
#include "prims.h"
#include "avsr_hip.h"

namespace {

constexpr int CF_THREADS = 512, CF_R = 4, CF_MAXK = 31;  // rows <= CF_THREADS * CF_R = 2048

template <class T> AVSR_DEV float ras(float v) { return v; }  // "round as stored"
template <> AVSR_DEV float ras<bf16_t>(float v) { return bf2f(f2bf(v)); }

AVSR_DEV float silu_f(float z) { return avsr_silu(z); }
AVSR_DEV float silu_grad(float z) {
    const float s = avsr_sigmoid(z);
    return s * (1.f + z * (1.f - s));
}

AVSR_DEV void block_sum16(float (&a)[8], float (&b)[8], float* red /* [8 waves][16] */) {
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

// TA: storage type of a / c (float or bf16_t); TS: of s (float, bf16_t or f16_t)
template <class TA, class TS>
__global__ __launch_bounds__(CF_THREADS) void convmod_dwbn_fwd_kernel(
    const TA* __restrict__ a, const float* __restrict__ wdw, const float* __restrict__ bdw, int rows, int Tlen, int C, int K,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
    float* __restrict__ running_var, int64_t* __restrict__ num_batches_tracked, TA* __restrict__ c_out, bf16_t* __restrict__ c2,
    TS* __restrict__ s, bf16_t* __restrict__ s2, float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    AVSR_DYN_SMEM(smem);
    float* G = reinterpret_cast<float*>(smem);  // [rows][8]: glu(a) of this block's channels
    float* ws = G + (size_t)rows * 8;            // [K][8]
    float* red = ws + CF_MAXK * 8;               // [8 waves][16]
    float* bc = red + 8 * 16;                    // [8] shift row / [16] mean, invstd
    const int c0 = blockIdx.x * 8, pad = (K - 1) / 2;
    for (int i = threadIdx.x; i < K * 8; i += CF_THREADS) ws[i] = wdw[(long)(c0 + (i & 7)) * K + (i >> 3)];
#pragma unroll
    for (int u = 0; u < CF_R; u++) {
        const int r = threadIdx.x + u * CF_THREADS;
        if (r < rows) {
            float v[8], g[8];
            load8(a + (long)r * 2 * C + c0, v);
            load8(a + (long)r * 2 * C + C + c0, g);
#pragma unroll
            for (int e = 0; e < 8; e++) G[r * 8 + e] = ras<TA>(v[e] * avsr_sigmoid(g[e]));
        }
    }
    __syncthreads();
    float bv[8], v[CF_R][8];
    if (bdw) load8(bdw + c0, bv);
    else
#pragma unroll
        for (int e = 0; e < 8; e++) bv[e] = 0.f;
#pragma unroll
    for (int u = 0; u < CF_R; u++) {
        const int r = threadIdx.x + u * CF_THREADS;
        if (r >= rows) continue;
        const int t = r % Tlen;
#pragma unroll
        for (int e = 0; e < 8; e++) v[u][e] = bv[e];
        for (int k = 0; k < K; k++) {
            const int tt = t + k - pad;
            if (tt < 0 || tt >= Tlen) continue;  // (zero padding at the utterance's ends)
            const float* gr = G + (r + k - pad) * 8;
#pragma unroll
            for (int e = 0; e < 8; e++) v[u][e] += ws[k * 8 + e] * gr[e];
        }
#pragma unroll
        for (int e = 0; e < 8; e++) v[u][e] = ras<TA>(v[u][e]);
        if (c_out) store8(c_out + (long)r * C + c0, v[u]);
        if (c2) store8(c2 + (long)r * C + c0, v[u]);
        if (r == 0)
#pragma unroll
            for (int e = 0; e < 8; e++) bc[e] = v[u][e];  // the shift of the statistics: row 0 (as bn_small_fwd_kernel)
    }
    __syncthreads();
    // ---- BatchNorm statistics, running-stat update, normalise + SiLU: bn_small_fwd_kernel on the slab in registers
    float sh[8], sa[8], sb[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        sh[e] = bc[e];
        sa[e] = sb[e] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < CF_R; u++) {
        const int r = threadIdx.x + u * CF_THREADS;
        if (r >= rows) continue;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float d = v[u][e] - sh[e];
            sa[e] += d;
            sb[e] += d * d;
        }
    }
    block_sum16(sa, sb, red);
    if (threadIdx.x < 8) {
        const int e = threadIdx.x, c = c0 + e;
        double s1 = 0.0, sq = 0.0;
#pragma unroll
        for (int w = 0; w < CF_THREADS / 64; w++) {
            s1 += (double)red[w * 16 + e];
            sq += (double)red[w * 16 + 8 + e];
        }
        const double nn = (double)rows;
        const double mean = (double)sh[e] + s1 / nn;
        const double m2 = fmax(sq - s1 * s1 / nn, 0.0);
        const double var = m2 / nn;
        const float mf = (float)mean, isf = (float)(1.0 / sqrt(var + (double)eps));
        mean_out[c] = mf;
        invstd_out[c] = isf;
        bc[8 + e] = mf;
        bc[16 + e] = isf;
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
    for (int u = 0; u < CF_R; u++) {
        const int r = threadIdx.x + u * CF_THREADS;
        if (r >= rows) continue;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = silu_f((v[u][e] - bc[8 + e]) * bc[16 + e] * ga[e] + be[e]);
        store8(s + (long)r * C + c0, o);
        if (s2) store8(s2 + (long)r * C + c0, o);
    }
}

// T: storage type of a, c, ds and da (bf16_t in every mode but "precise")
template <class T>
__global__ __launch_bounds__(CF_THREADS) void convmod_dwbn_bwd_kernel(
    const T* __restrict__ a, const T* __restrict__ c, const T* __restrict__ ds, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ wdw,
    int rows, int Tlen, int C, int K, T* __restrict__ da, float* __restrict__ dwdw, float* __restrict__ dbdw, float* __restrict__ dgamma,
    float* __restrict__ dbeta) {
    AVSR_DYN_SMEM(smem);
    float* Dc = reinterpret_cast<float*>(smem);  // [rows][8]: gradient of the depthwise output (BatchNorm input)
    float* G = Dc + (size_t)rows * 8;            // [rows][8]: glu(a)
    float* ws = G + (size_t)rows * 8;            // [K][8]
    float* red = ws + CF_MAXK * 8;               // [8 waves][16]; re-used as [2][8][32] by the weight-gradient reduction (512 floats)
    float* bc = red + 512;                       // [16] sums
    const int c0 = blockIdx.x * 8, pad = (K - 1) / 2;
    for (int i = threadIdx.x; i < K * 8; i += CF_THREADS) ws[i] = wdw[(long)(c0 + (i & 7)) * K + (i >> 3)];
    float mu[8], is[8], ga[8], be[8], sa[8], sb[8];
    load8(mean + c0, mu);
    load8(invstd + c0, is);
    load8(gamma + c0, ga);
    load8(beta + c0, be);
    float xh[CF_R][8], dz[CF_R][8], lin[CF_R][8], sg[CF_R][8];
#pragma unroll
    for (int e = 0; e < 8; e++) sa[e] = sb[e] = 0.f;
#pragma unroll
    for (int u = 0; u < CF_R; u++) {
        const int r = threadIdx.x + u * CF_THREADS;
        if (r < rows) {
            load8(c + (long)r * C + c0, xh[u]);
            load8(ds + (long)r * C + c0, dz[u]);
            load8(a + (long)r * 2 * C + c0, lin[u]);
            load8(a + (long)r * 2 * C + C + c0, sg[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < CF_R; u++) {
        const int r = threadIdx.x + u * CF_THREADS;
        if (r >= rows) continue;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float h = (xh[u][e] - mu[e]) * is[e];
            const float z = h * ga[e] + be[e];
            const float d = dz[u][e] * silu_grad(z);
            xh[u][e] = h;
            dz[u][e] = d;
            sa[e] += d;
            sb[e] += d * h;
            sg[u][e] = avsr_sigmoid(sg[u][e]);
            G[r * 8 + e] = ras<T>(lin[u][e] * sg[u][e]);
        }
    }
    block_sum16(sa, sb, red);
    if (threadIdx.x < 8) {
        const int e = threadIdx.x;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < CF_THREADS / 64; w++) {
            s1 += red[w * 16 + e];
            s2 += red[w * 16 + 8 + e];
        }
        bc[e] = s1;
        bc[8 + e] = s2;
        dbeta[c0 + e] = s1;
        dgamma[c0 + e] = s2;
    }
    __syncthreads();
    const float inv_n = 1.0f / (float)rows;
#pragma unroll
    for (int u = 0; u < CF_R; u++) {
        const int r = threadIdx.x + u * CF_THREADS;
        if (r >= rows) continue;
#pragma unroll
        for (int e = 0; e < 8; e++)
            Dc[r * 8 + e] = ras<T>(ga[e] * is[e] * (dz[u][e] - bc[e] * inv_n - xh[u][e] * bc[8 + e] * inv_n));
    }
    __syncthreads();
    // ---- depthwise weight / bias gradient: thread = (tap lane k, channel e, row half); lane 31 of a channel carries the bias sum
    {
        const int k = threadIdx.x & 31, e = (threadIdx.x >> 5) & 7, half = threadIdx.x >> 8;
        const int kk = min(k, K - 1);
        const int r0 = half ? rows / 2 : 0, r1 = half ? rows : rows / 2;
        float acc = 0.f, sbias = 0.f;
        int t = r0 % Tlen;
        for (int r = r0; r < r1; r++) {
            const float g = Dc[r * 8 + e];
            const int tt = t + kk - pad;
            if (tt >= 0 && tt < Tlen) acc += g * G[(r + kk - pad) * 8 + e];
            sbias += g;
            if (++t == Tlen) t = 0;
        }
        __syncthreads();  // (red is free again)
        red[(half * 8 + e) * 32 + k] = k == 31 ? sbias : acc;
        // (lane 31 is never a tap: K <= 31)
        __syncthreads();
        if (half == 0) {
            const float tot = red[e * 32 + k] + red[(8 + e) * 32 + k];
            if (k < K) dwdw[(long)(c0 + e) * K + k] += tot;  // (single writer: no atomics; taps are lanes 0 .. K-1 <= 30)
            if (k == 31 && dbdw) dbdw[c0 + e] += tot;
        }
    }
    // ---- data gradient of the depthwise convolution (taps reversed) with the GLU backward as its epilogue
#pragma unroll
    for (int u = 0; u < CF_R; u++) {
        const int r = threadIdx.x + u * CF_THREADS;
        if (r >= rows) continue;
        const int t = r % Tlen;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = 0.f;
        for (int k = 0; k < K; k++) {
            const int tt = t + k - pad;
            if (tt < 0 || tt >= Tlen) continue;
            const float* gr = Dc + (r + k - pad) * 8;
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] += ws[(K - 1 - k) * 8 + e] * gr[e];
        }
        float o1[8], o2[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float g = ras<T>(acc[e]);
            o1[e] = g * sg[u][e];
            o2[e] = g * lin[u][e] * sg[u][e] * (1.f - sg[u][e]);
        }
        store8(da + (long)r * 2 * C + c0, o1);
        store8(da + (long)r * 2 * C + C + c0, o2);
    }
}

}  // namespace

extern "C" int avsr_convmod_fused_max_rows(void) { return CF_THREADS * CF_R; }

// a [rows = B*T][2C] (dtype 0 = f32, 1 = bf16), depthwise weights wdw [C][K] / bias bdw [C] (may be NULL), BatchNorm gamma / beta / eps /
// momentum / running statistics / batch counter (may be NULL) -> s [rows][C] (s_dtype 0 / 1 / 2) and its bf16 twin s2 (may be NULL);
// c_out (dtype of a, may be NULL) / c2 (bf16, may be NULL): the depthwise output the backward pass needs; mean / invstd [C].
extern "C" int avsr_convmod_dwbn_fwd(const void* a, int dtype, const float* wdw, const float* bdw, int B, int T, int C, int K,
                                     const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                     float* running_var, int64_t* num_batches_tracked, void* c_out, void* c2, void* s, int s_dtype,
                                     void* s2, float* mean, float* invstd, hipStream_t stream) {
    AVSR_REQUIRE(K >= 1 && K <= CF_MAXK && (K & 1), "convmod_dwbn: K must be odd and <= 31");
    AVSR_REQUIRE(C % 8 == 0, "convmod_dwbn: C must be a multiple of 8");
    const long rows = (long)B * T;
    AVSR_REQUIRE(rows >= 1 && rows <= CF_THREADS * CF_R, "convmod_dwbn: 1 <= B*T <= 2048");
    AVSR_REQUIRE((dtype == 0 || dtype == 1) && s_dtype >= 0 && s_dtype <= 2, "convmod_dwbn: bad dtype");
    const size_t lds = ((size_t)rows * 8 + CF_MAXK * 8 + 8 * 16 + 32) * sizeof(float);
    dim3 grid(C / 8), block(CF_THREADS);
#define AVSR_GO(TA, TS)                                                                                                              \
    AVSR_LAUNCH((convmod_dwbn_fwd_kernel<TA, TS>), grid, block, lds, stream, (const TA*)a, wdw, bdw, (int)rows, T, C, K, gamma, beta, eps, \
                momentum, running_mean, running_var, num_batches_tracked, (TA*)c_out, (bf16_t*)c2, (TS*)s, (bf16_t*)s2, mean, invstd)
    if (dtype == 0) {
        if (s_dtype == 0) AVSR_GO(float, float);
        else if (s_dtype == 1) AVSR_GO(float, bf16_t);
        else AVSR_GO(float, f16_t);
    } else {
        AVSR_REQUIRE(s_dtype == 1, "convmod_dwbn: a bf16 chain writes a bf16 result");
        AVSR_GO(bf16_t, bf16_t);
    }
#undef AVSR_GO
    AVSR_CHECK_LAUNCH("convmod_dwbn_fwd");
    return 0;
}

// a / c / ds / da in dtype (0 = f32, 1 = bf16); dwdw [C][K] / dbdw [C] (may be NULL) are ADDED to (the caller zeroes them);
// dgamma / dbeta [C] are overwritten.
extern "C" int avsr_convmod_dwbn_bwd(const void* a, const void* c, const void* ds, int dtype, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, const float* wdw, int B, int T, int C, int K, void* da,
                                     float* dwdw, float* dbdw, float* dgamma, float* dbeta, hipStream_t stream) {
    AVSR_REQUIRE(K >= 1 && K <= CF_MAXK && (K & 1), "convmod_dwbn: K must be odd and <= 31");
    AVSR_REQUIRE(C % 8 == 0, "convmod_dwbn: C must be a multiple of 8");
    const long rows = (long)B * T;
    AVSR_REQUIRE(rows >= 1 && rows <= CF_THREADS * CF_R, "convmod_dwbn: 1 <= B*T <= 2048");
    AVSR_REQUIRE(dtype == 0 || dtype == 1, "convmod_dwbn: bad dtype");
    const size_t lds = ((size_t)rows * 16 + CF_MAXK * 8 + 512 + 16) * sizeof(float);
    dim3 grid(C / 8), block(CF_THREADS);
    if (dtype == 0)
        AVSR_LAUNCH((convmod_dwbn_bwd_kernel<float>), grid, block, lds, stream, (const float*)a, (const float*)c, (const float*)ds, mean,
                    invstd, gamma, beta, wdw, (int)rows, T, C, K, (float*)da, dwdw, dbdw, dgamma, dbeta);
    else
        AVSR_LAUNCH((convmod_dwbn_bwd_kernel<bf16_t>), grid, block, lds, stream, (const bf16_t*)a, (const bf16_t*)c, (const bf16_t*)ds,
                    mean, invstd, gamma, beta, wdw, (int)rows, T, C, K, (bf16_t*)da, dwdw, dbdw, dgamma, dbeta);
    AVSR_CHECK_LAUNCH("convmod_dwbn_bwd");
    return 0;
}
