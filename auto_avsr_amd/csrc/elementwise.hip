// elementwise.hip -- HBM-bound glue kernels of the Conformer / decoder blocks: everything that is not
// a contraction and could not be folded into a GEMM epilogue.  All are grid-stride, 8 elements
// (16 B bf16 / 32 B f32) per lane per access.
//
//   scale_dropout   x*alpha with inverted-dropout mask        (embedding.py:179-184 xscale+dropout;
//                                                               conformer_encoder.py:114,141,150,157 dropout
//                                                               on the backward side; ctc.py:54 dropout)
//   head_bias       q + pos_bias_u / q + pos_bias_v            (attention.py:176-178)
//   glu             a * sigmoid(b) over the channel halves     (conformer_encoder.py:32)
//   colsum          bias gradients (sum over rows)
#include "prims.h"
#include "avsr_hip.h"

namespace {

constexpr int EW_THREADS = 256;

static inline int ew_grid(long nvec) {
    long b = (nvec + EW_THREADS - 1) / EW_THREADS;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

template <class TI, class TO>
__global__ __launch_bounds__(EW_THREADS) void scale_dropout_kernel(const TI* __restrict__ x, TO* __restrict__ out,
                                                                   long n, float alpha0, const float* alpha_dev, float p, uint64_t seed0,
                                                                   const uint64_t* seed_dev, const float* __restrict__ add,
                                                                   long period) {
    const float alpha = alpha0 * (alpha_dev ? *alpha_dev : 1.f);
    const uint64_t seed = seed0 + (seed_dev ? *seed_dev : 0ull);
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const long nvec = n >> 3;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < nvec; i += (long)gridDim.x * EW_THREADS) {
        float v[8], a[8];
        load8(x + i * 8, v);
        if (add) load8(add + (i * 8) % period, a);  // period % 8 == 0
#pragma unroll
        for (int e = 0; e < 8; e++)
            v[e] = (v[e] * alpha + (add ? a[e] : 0.f)) * dropout_scale(seed, (uint64_t)(i * 8 + e), p, inv_keep);
        store8(out + i * 8, v);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long i = nvec * 8; i < n; i++)
            Elem<TO>::st(out + i, (Elem<TI>::ld(x + i) * alpha + (add ? add[i % period] : 0.f)) *
                                      dropout_scale(seed, (uint64_t)i, p, inv_keep));
}

// out1 = x + b1[col], out2 = x + b2[col]   (cols % 8 == 0)
template <class T>
__global__ __launch_bounds__(EW_THREADS) void head_bias_kernel(const T* __restrict__ x, long ldx,
                                                               const float* __restrict__ b1,
                                                               const float* __restrict__ b2, T* __restrict__ o1,
                                                               T* __restrict__ o2, long rows, int cols,
                                                               bf16_t* __restrict__ t1 = nullptr, bf16_t* __restrict__ t2 = nullptr) {
    const int cv = cols >> 3;
    const long nvec = rows * cv;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < nvec; i += (long)gridDim.x * EW_THREADS) {
        const long r = i / cv;
        const int c = (int)(i % cv) * 8;
        float v[8], u[8], w[8], a[8], b[8];
        load8(x + r * ldx + c, v);
        load8(b1 + c, u);
        load8(b2 + c, w);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            a[e] = v[e] + u[e];
            b[e] = v[e] + w[e];
        }
        store8(o1 + r * cols + c, a);
        store8(o2 + r * cols + c, b);
        if (t1) {  // bf16 twins of f32 outputs (the "hpf" numerical mode: f32 forward, bf16 copies saved for the backward pass)
            store8(t1 + r * cols + c, a);
            store8(t2 + r * cols + c, b);
        }
    }
}

// dq = d1 + d2 (written with row stride ldo); db1 += colsum(d1), db2 += colsum(d2).
// thread = (8-column chunk, row lane): CL chunk lanes x RL row lanes per block, rows strided by RL; the RL
// partials meet in LDS and one lane per column issues the atomics.
template <class T>
__global__ __launch_bounds__(EW_THREADS) void head_bias_bwd_kernel(const T* __restrict__ d1,
                                                                   const T* __restrict__ d2, T* __restrict__ dq,
                                                                   long ldo, float* __restrict__ db1,
                                                                   float* __restrict__ db2, long rows, int cols,
                                                                   int rows_per_block, int CL) {
    __shared__ float red[EW_THREADS * 16];
    const int cv = cols >> 3;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL, RL = EW_THREADS / CL;
    const int cc = blockIdx.x * CL + cl;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; e++) s1[e] = s2[e] = 0.f;
    if (cc < cv) {
        for (long r = r0 + rl; r < r1; r += RL) {
            float a[8], b[8], o[8];
            load8(d1 + r * cols + cc * 8, a);
            if (d2) load8(d2 + r * cols + cc * 8, b);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (!d2) b[e] = 0.f;
                o[e] = a[e] + b[e];
                s1[e] += a[e];
                s2[e] += b[e];
            }
            if (dq) store8(dq + r * ldo + cc * 8, o);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
        red[threadIdx.x * 16 + e] = s1[e];
        red[threadIdx.x * 16 + 8 + e] = s2[e];
    }
    __syncthreads();
    if (rl == 0 && cc < cv) {
        for (int q = 1; q < RL; q++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                s1[e] += red[(q * CL + cl) * 16 + e];
                s2[e] += red[(q * CL + cl) * 16 + 8 + e];
            }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if (db1) atomicAdd(db1 + cc * 8 + e, s1[e]);
            if (db2) atomicAdd(db2 + cc * 8 + e, s2[e]);
        }
    }
}

// g[r, c] = a[r, c] * sigmoid(a[r, C + c])
template <class T>
__global__ __launch_bounds__(EW_THREADS) void glu_fwd_kernel(const T* __restrict__ a, T* __restrict__ g, long rows,
                                                             int C) {
    const int cv = C >> 3;
    const long nvec = rows * cv;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < nvec; i += (long)gridDim.x * EW_THREADS) {
        const long r = i / cv;
        const int c = (int)(i % cv) * 8;
        float x[8], y[8], o[8];
        load8(a + r * 2 * C + c, x);
        load8(a + r * 2 * C + C + c, y);
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = x[e] * avsr_sigmoid(y[e]);
        store8(g + r * C + c, o);
    }
}
template <class T>
__global__ __launch_bounds__(EW_THREADS) void glu_bwd_kernel(const T* __restrict__ a, const T* __restrict__ dg,
                                                             T* __restrict__ da, long rows, int C) {
    const int cv = C >> 3;
    const long nvec = rows * cv;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < nvec; i += (long)gridDim.x * EW_THREADS) {
        const long r = i / cv;
        const int c = (int)(i % cv) * 8;
        float x[8], y[8], d[8], o1[8], o2[8];
        load8(a + r * 2 * C + c, x);
        load8(a + r * 2 * C + C + c, y);
        load8(dg + r * C + c, d);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float s = avsr_sigmoid(y[e]);
            o1[e] = d[e] * s;
            o2[e] = d[e] * x[e] * s * (1.f - s);
        }
        store8(da + r * 2 * C + c, o1);
        store8(da + r * 2 * C + C + c, o2);
    }
}

}  // namespace


template <class TI, class TO>
static void launch_scale_dropout(const void* x, void* out, long n, float alpha, const float* alpha_dev, float p,
                                 uint64_t seed, const uint64_t* seed_dev, const float* add, long period, hipStream_t stream) {
    dim3 grid(ew_grid(n >> 3)), block(EW_THREADS);
    AVSR_LAUNCH((scale_dropout_kernel<TI, TO>), grid, block, 0, stream, (const TI*)x, (TO*)out, n, alpha, alpha_dev, p, seed, seed_dev, add, period);
}

extern "C" int avsr_scale_dropout(const void* x, int x_dtype, void* out, int out_dtype, int64_t n, float alpha,
                                  const float* alpha_dev, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                                  const float* add, int64_t add_period, hipStream_t stream) {
    if (n <= 0) return 0;
    AVSR_REQUIRE(add == nullptr || (add_period > 0 && add_period % 8 == 0), "scale_dropout: add_period must be a positive multiple of 8");
    AVSR_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0, "scale_dropout: 16-byte alignment");
    if (x_dtype == 0 && out_dtype == 0) launch_scale_dropout<float, float>(x, out, n, alpha, alpha_dev, drop_p, seed, seed_dev, add, (long)add_period, stream);
    else if (x_dtype == 0 && out_dtype == 1) launch_scale_dropout<float, bf16_t>(x, out, n, alpha, alpha_dev, drop_p, seed, seed_dev, add, (long)add_period, stream);
    else if (x_dtype == 0 && out_dtype == 2) launch_scale_dropout<float, f16_t>(x, out, n, alpha, alpha_dev, drop_p, seed, seed_dev, add, (long)add_period, stream);
    // x_dtype 3: the split8 storage layout of an f32-sized activation (prims.h sp8_t; n % 8 == 0)
    else if (x_dtype == 3 && n % 8 != 0) { avsr_set_error("scale_dropout: a split8 source needs n % 8 == 0"); return 1; }
    else if (x_dtype == 3 && out_dtype == 0) launch_scale_dropout<sp8_t, float>(x, out, n, alpha, alpha_dev, drop_p, seed, seed_dev, add, (long)add_period, stream);
    else if (x_dtype == 3 && out_dtype == 1) launch_scale_dropout<sp8_t, bf16_t>(x, out, n, alpha, alpha_dev, drop_p, seed, seed_dev, add, (long)add_period, stream);
    else if (x_dtype == 3 && out_dtype == 2) launch_scale_dropout<sp8_t, f16_t>(x, out, n, alpha, alpha_dev, drop_p, seed, seed_dev, add, (long)add_period, stream);
    else if (x_dtype == 2 && out_dtype == 0) launch_scale_dropout<f16_t, float>(x, out, n, alpha, alpha_dev, drop_p, seed, seed_dev, add, (long)add_period, stream);
    else if (x_dtype == 2 && out_dtype == 1) launch_scale_dropout<f16_t, bf16_t>(x, out, n, alpha, alpha_dev, drop_p, seed, seed_dev, add, (long)add_period, stream);
    else if (x_dtype == 2 || out_dtype == 2) { avsr_set_error("scale_dropout: unsupported f16 combination"); return 1; }
    else if (out_dtype == 0) launch_scale_dropout<bf16_t, float>(x, out, n, alpha, alpha_dev, drop_p, seed, seed_dev, add, (long)add_period, stream);
    else launch_scale_dropout<bf16_t, bf16_t>(x, out, n, alpha, alpha_dev, drop_p, seed, seed_dev, add, (long)add_period, stream);
    AVSR_CHECK_LAUNCH("scale_dropout");
    return 0;
}

extern "C" int avsr_head_bias_fwd(const void* x, int dtype, int64_t ldx, const float* b1, const float* b2, void* o1,
                                  void* o2, int64_t rows, int cols, hipStream_t stream) {
    AVSR_REQUIRE(cols % 8 == 0 && ldx % 8 == 0, "head_bias: cols and ldx must be multiples of 8");
    if (rows <= 0) return 0;
    dim3 grid(ew_grid(rows * (cols >> 3))), block(EW_THREADS);
    if (dtype == 0)
        AVSR_LAUNCH((head_bias_kernel<float>), grid, block, 0, stream, (const float*)x, (long)ldx, b1, b2, (float*)o1,
                    (float*)o2, (long)rows, cols);
    else
        AVSR_LAUNCH((head_bias_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)x, (long)ldx, b1, b2,
                    (bf16_t*)o1, (bf16_t*)o2, (long)rows, cols);
    AVSR_CHECK_LAUNCH("head_bias_fwd");
    return 0;
}

// f32 outputs + their bf16 twins in one pass
extern "C" int avsr_head_bias_fwd2(const float* x, int64_t ldx, const float* b1, const float* b2, float* o1, float* o2, void* t1,
                                   void* t2, int64_t rows, int cols, hipStream_t stream) {
    AVSR_REQUIRE(cols % 8 == 0 && ldx % 8 == 0, "head_bias_fwd: cols/ldx must be multiples of 8");
    AVSR_REQUIRE((t1 == nullptr) == (t2 == nullptr), "head_bias_fwd2: both twins or none");
    if (rows <= 0) return 0;
    dim3 grid(ew_grid(rows * (cols >> 3))), block(EW_THREADS);
    AVSR_LAUNCH((head_bias_kernel<float>), grid, block, 0, stream, x, (long)ldx, b1, b2, o1, o2, (long)rows, cols, (bf16_t*)t1,
                (bf16_t*)t2);
    AVSR_CHECK_LAUNCH("head_bias_fwd2");
    return 0;
}

// f16 input / outputs + their bf16 twins (mixed mode: the attention forward reads f16, its backward the twins)
extern "C" int avsr_head_bias_fwd_h16(const void* x, int64_t ldx, const float* b1, const float* b2, void* o1, void* o2, void* t1,
                                      void* t2, int64_t rows, int cols, hipStream_t stream) {
    AVSR_REQUIRE(cols % 8 == 0 && ldx % 8 == 0, "head_bias_fwd: cols/ldx must be multiples of 8");
    AVSR_REQUIRE((t1 == nullptr) == (t2 == nullptr), "head_bias_fwd_h16: both twins or none");
    if (rows <= 0) return 0;
    dim3 grid(ew_grid(rows * (cols >> 3))), block(EW_THREADS);
    AVSR_LAUNCH((head_bias_kernel<f16_t>), grid, block, 0, stream, (const f16_t*)x, (long)ldx, b1, b2, (f16_t*)o1, (f16_t*)o2,
                (long)rows, cols, (bf16_t*)t1, (bf16_t*)t2);
    AVSR_CHECK_LAUNCH("head_bias_fwd_h16");
    return 0;
}

// dq[r, :] (row stride ldo) = d1 + d2 (d2 may be NULL; dq may be NULL); db1 += colsum(d1); db2 += colsum(d2).
// With d2 = dq = db2 = NULL this is the plain bias-gradient column sum.
namespace {
// deterministic column sums: thread = (8-column chunk, row lane); 32 chunks x 8 row lanes per block; fixed summation order
template <class T>
__global__ __launch_bounds__(256) void colsum_det_kernel(const T* __restrict__ src, long ld, long rows, int cols, float* __restrict__ out) {
    __shared__ float red[256 * 8];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c0 = (blockIdx.x * 32 + cl) * 8;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; e++) s[e] = 0.f;
    if (c0 < cols) {
        const bool vec = c0 + 8 <= cols && (ld % 8 == 0);
        for (long r = rl; r < rows; r += 8) {
            float v[8];
            if (vec) load8(src + r * ld + c0, v);
            else
                for (int e = 0; e < 8; e++) v[e] = c0 + e < cols ? Elem<T>::ld(src + r * ld + c0 + e) : 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) s[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) red[threadIdx.x * 8 + e] = s[e];
    __syncthreads();
    if (rl == 0 && c0 < cols) {
        for (int q = 1; q < 8; q++)
#pragma unroll
            for (int e = 0; e < 8; e++) s[e] += red[(q * 32 + cl) * 8 + e];
        for (int e = 0; e < 8 && c0 + e < cols; e++) out[c0 + e] += s[e];
    }
}
// deterministic row sums of a bf16 matrix: one wave per row, lanes stride the columns, butterfly in a fixed order
__global__ __launch_bounds__(256) void rowsum_det_kernel(const bf16_t* __restrict__ src, long ld, long rows, long cols, float* __restrict__ out) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    float s = 0.f;
    for (long c = lane; c < cols; c += 64) s += bf2f(src[r * ld + c]);
    s = wave_sum(s);
    if (lane == 0) out[r] += s;
}
}  // namespace

int avsr_colsum_det(const void* src, int dtype, long ld, long rows, int cols, float* out, hipStream_t stream) {
    if (rows <= 0 || cols <= 0) return 0;
    dim3 grid((cols + 255) / 256), block(256);
    if (dtype == 0) AVSR_LAUNCH((colsum_det_kernel<float>), grid, block, 0, stream, (const float*)src, ld, rows, cols, out);
    else if (dtype == 1) AVSR_LAUNCH((colsum_det_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)src, ld, rows, cols, out);
    else AVSR_LAUNCH((colsum_det_kernel<f16_t>), grid, block, 0, stream, (const f16_t*)src, ld, rows, cols, out);
    return 0;
}
int avsr_rowsum_det_bf16(const void* src, long ld, long rows, long cols, float* out, hipStream_t stream) {
    if (rows <= 0 || cols <= 0) return 0;
    AVSR_LAUNCH(rowsum_det_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)src, ld, rows, cols, out);
    return 0;
}

extern "C" int avsr_head_bias_bwd(const void* d1, const void* d2, int dtype, void* dq, int64_t ldo, float* db1,
                                  float* db2, int64_t rows, int cols, hipStream_t stream) {
    AVSR_REQUIRE(cols % 8 == 0 && (dq == nullptr || ldo % 8 == 0), "head_bias_bwd: cols/ldo must be multiples of 8");
    if (rows <= 0) return 0;
    const int cv = cols >> 3;
    const int CL = cv >= 32 ? 32 : (cv >= 16 ? 16 : 8);
    // deterministic mode: ONE block per column group walks all rows (a single writer per bias-gradient element)
    const int rpb = avsr_det() ? (int)rows : 8 * (EW_THREADS / CL);  // 8 rows per thread
    dim3 grid((cv + CL - 1) / CL, (unsigned)((rows + rpb - 1) / rpb)), block(EW_THREADS);
    if (dtype == 0)
        AVSR_LAUNCH((head_bias_bwd_kernel<float>), grid, block, 0, stream, (const float*)d1, (const float*)d2,
                    (float*)dq, (long)ldo, db1, db2, (long)rows, cols, rpb, CL);
    else
        AVSR_LAUNCH((head_bias_bwd_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)d1, (const bf16_t*)d2,
                    (bf16_t*)dq, (long)ldo, db1, db2, (long)rows, cols, rpb, CL);
    AVSR_CHECK_LAUNCH("head_bias_bwd");
    return 0;
}

extern "C" int avsr_glu_fwd(const void* a, void* g, int dtype, int64_t rows, int C, hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "glu: C must be a multiple of 8");
    if (rows <= 0) return 0;
    dim3 grid(ew_grid(rows * (C >> 3))), block(EW_THREADS);
    if (dtype == 0) AVSR_LAUNCH((glu_fwd_kernel<float>), grid, block, 0, stream, (const float*)a, (float*)g, (long)rows, C);
    else AVSR_LAUNCH((glu_fwd_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)a, (bf16_t*)g, (long)rows, C);
    AVSR_CHECK_LAUNCH("glu_fwd");
    return 0;
}

extern "C" int avsr_glu_bwd(const void* a, const void* dg, void* da, int dtype, int64_t rows, int C,
                            hipStream_t stream) {
    AVSR_REQUIRE(C % 8 == 0, "glu: C must be a multiple of 8");
    if (rows <= 0) return 0;
    dim3 grid(ew_grid(rows * (C >> 3))), block(EW_THREADS);
    if (dtype == 0)
        AVSR_LAUNCH((glu_bwd_kernel<float>), grid, block, 0, stream, (const float*)a, (const float*)dg, (float*)da,
                    (long)rows, C);
    else
        AVSR_LAUNCH((glu_bwd_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)a, (const bf16_t*)dg,
                    (bf16_t*)da, (long)rows, C);
    AVSR_CHECK_LAUNCH("glu_bwd");
    return 0;
}
