// pool.hip -- spatial pooling of the front-ends on channels-last tensors (HBM-bound, 8 channels per lane).
//   maxpool   MaxPool3d((1,3,3), stride (1,2,2), padding (0,1,1))   frontend/resnet.py:214-218  (-inf padding)
//   avgpool   AdaptiveAvgPool2d(1) (resnet.py:117,164) / AvgPool1d(20,20) (resnet1d.py:143-146): mean over
//             groups of `win` consecutive pixels.
#include "prims.h"
#include "avsr_hip.h"

namespace {

// y[n,oh,ow,c] = max over the KxK window (stride S, pad P) of x[n,ih,iw,c]; idx = kh*K+kw of the FIRST maximum in
// scan order (the element torch routes the gradient to), one byte per output element
template <class T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ idx, long N, int H, int W, int C, int OH,
                                                          int OW, int K, int S, int P) {
    const int cv = C >> 3;
    const long total = N * OH * OW * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * 8;
        long r = i / cv;
        const int ow = (int)(r % OW);
        r /= OW;
        const int oh = (int)(r % OH);
        const long n = r / OH;
        float m[8];
        uint8_t am[8];
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
                for (int e = 0; e < 8; e++)
                    if (v[e] > m[e]) {
                        m[e] = v[e];
                        am[e] = (uint8_t)(kh * K + kw);
                    }
            }
        }
        store8(y + i * 8, m);
        if (idx) {
            uint64_t pk = 0;
#pragma unroll
            for (int e = 0; e < 8; e++) pk |= (uint64_t)am[e] << (8 * e);
            *reinterpret_cast<uint64_t*>(idx + i * 8) = pk;
        }
    }
}

// dx[n,ih,iw,c] = sum of dy over the (at most ceil(K/S)^2) windows whose recorded arg-max is (ih,iw)
template <class T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const uint8_t* __restrict__ idx, const T* __restrict__ dy,
                                                          T* __restrict__ dx, long N, int H, int W, int C, int OH, int OW,
                                                          int K, int S, int P) {
    const int cv = C >> 3;
    const long total = N * H * W * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * 8;
        long r = i / cv;
        const int iw = (int)(r % W);
        r /= W;
        const int ih = (int)(r % H);
        const long n = r / H;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = 0.f;
        const int oh_lo = max(0, (ih + P - K + 1 + S - 1) / S), oh_hi = min(OH - 1, (ih + P) / S);
        const int ow_lo = max(0, (iw + P - K + 1 + S - 1) / S), ow_hi = min(OW - 1, (iw + P) / S);
        for (int oh = oh_lo; oh <= oh_hi; oh++)
            for (int ow = ow_lo; ow <= ow_hi; ow++) {
                const int me = (ih - (oh * S - P)) * K + (iw - (ow * S - P));
                const long o = ((n * OH + oh) * OW + ow) * C + c;
                const uint64_t pk = *reinterpret_cast<const uint64_t*>(idx + o);
                float g[8];
                load8(dy + o, g);
#pragma unroll
                for (int e = 0; e < 8; e++)
                    if ((int)((pk >> (8 * e)) & 0xff) == me) acc[e] += g[e];
            }
        store8(dx + i * 8, acc);
    }
}

// y[g, c] = mean_{j < win} x[g*win + j, c]   (f32 output)
template <class T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, float* __restrict__ y, long groups,
                                                          int win, int C) {
    const int cv = C >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < groups * cv; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * 8;
        const long g = i / cv;
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; e++) s[e] = 0.f;
        for (int j = 0; j < win; j++) {
            float v[8];
            load8(x + (g * win + j) * C + c, v);
#pragma unroll
            for (int e = 0; e < 8; e++) s[e] += v[e];
        }
        const float inv = 1.f / (float)win;
#pragma unroll
        for (int e = 0; e < 8; e++) s[e] *= inv;
        store8(y + i * 8, s);
    }
}
template <class T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, T* __restrict__ dx, long groups,
                                                          int win, int C) {
    const int cv = C >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < groups * win * cv; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * 8;
        const long g = (i / cv) / win;
        float v[8];
        load8(dy + g * C + c, v);
        const float inv = 1.f / (float)win;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] *= inv;
        store8(dx + i * 8, v);
    }
}

static inline unsigned grid_for(long n) {
    long b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

extern "C" int avsr_maxpool2d_fwd(const void* x, void* y, uint8_t* idx, int dtype, int64_t N, int H, int W, int C, int K, int S, int P,
                                  hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "maxpool: C must be a multiple of 8");
    const int OH = (H + 2 * P - K) / S + 1, OW = (W + 2 * P - K) / S + 1;
    if (N <= 0) return 0;
    const long total = (long)N * OH * OW * (C >> 3);
    if (dtype == 0)
        AVSR_LAUNCH((maxpool_fwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, stream, (const float*)x, (float*)y, idx, (long)N, H, W, C, OH, OW, K, S, P);
    else
        AVSR_LAUNCH((maxpool_fwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, idx, (long)N, H, W, C, OH, OW, K, S, P);
    AVSR_CHECK_LAUNCH("maxpool2d_fwd");
    return 0;
}

extern "C" int avsr_maxpool2d_bwd(const uint8_t* idx, const void* dy, void* dx, int dtype, int64_t N, int H, int W, int C, int K,
                                  int S, int P, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "maxpool: C must be a multiple of 8");
    const int OH = (H + 2 * P - K) / S + 1, OW = (W + 2 * P - K) / S + 1;
    if (N <= 0) return 0;
    const long total = (long)N * H * W * (C >> 3);
    if (dtype == 0)
        AVSR_LAUNCH((maxpool_bwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, stream, idx, (const float*)dy, (float*)dx, (long)N, H, W, C, OH, OW, K, S, P);
    else
        AVSR_LAUNCH((maxpool_bwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, stream, idx, (const bf16_t*)dy, (bf16_t*)dx, (long)N, H, W, C, OH, OW, K, S, P);
    AVSR_CHECK_LAUNCH("maxpool2d_bwd");
    return 0;
}

extern "C" int avsr_avgpool_fwd(const void* x, int dtype, float* y, int64_t groups, int win, int C, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "avgpool: C must be a multiple of 8");
    if (groups <= 0) return 0;
    if (dtype == 0) AVSR_LAUNCH((avgpool_fwd_kernel<float>), dim3(grid_for(groups * (C >> 3))), dim3(256), 0, stream, (const float*)x, y, (long)groups, win, C);
    else if (dtype == 2) AVSR_LAUNCH((avgpool_fwd_kernel<f16_t>), dim3(grid_for(groups * (C >> 3))), dim3(256), 0, stream, (const f16_t*)x, y, (long)groups, win, C);
    else AVSR_LAUNCH((avgpool_fwd_kernel<bf16_t>), dim3(grid_for(groups * (C >> 3))), dim3(256), 0, stream, (const bf16_t*)x, y, (long)groups, win, C);
    AVSR_CHECK_LAUNCH("avgpool_fwd");
    return 0;
}

extern "C" int avsr_avgpool_bwd(const float* dy, void* dx, int dtype, int64_t groups, int win, int C, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "avgpool: C must be a multiple of 8");
    if (groups <= 0) return 0;
    if (dtype == 0) AVSR_LAUNCH((avgpool_bwd_kernel<float>), dim3(grid_for(groups * win * (C >> 3))), dim3(256), 0, stream, dy, (float*)dx, (long)groups, win, C);
    else AVSR_LAUNCH((avgpool_bwd_kernel<bf16_t>), dim3(grid_for(groups * win * (C >> 3))), dim3(256), 0, stream, dy, (bf16_t*)dx, (long)groups, win, C);
    AVSR_CHECK_LAUNCH("avgpool_bwd");
    return 0;
}
