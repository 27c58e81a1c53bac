// gemm_fast.hip -- the tuned path of the GEMM family: C[M,N] = epi(A[M,K] . B[N,K]^T) with both operands bf16
// and k-contiguous (K a multiple of 64).  Every dense contraction of the bf16 hot path is brought into this
// form: Linear forward (B = bf16 weight copy), data gradient (B = transposed weight copy), weight gradient
// (A = dY^T, B = x^T transposed activation copies with the token dimension zero-padded to 64).
//
// CDNA4 structure:
//  * operands go HBM -> LDS directly by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) into a
//    STAGES-deep ring; no VGPR staging, no ds_write pass; loads of tile t+STAGES-1 are issued before tile t is
//    multiplied, and are waited for with a COUNTED s_waitcnt vmcnt (never 0 in the steady state) in front of a raw
//    s_barrier, so DMA stays in flight across barriers;
//  * LDS image per operand stage: [rows][64] bf16 = 128-byte rows, lane-linear (what the DMA writes), with the
//    16-byte chunk index XOR-swizzled by (row>>1)&7 -- applied to the per-lane SOURCE address on the way in and to
//    the ds_read_b128 address on the way out -- conflict-free for the 32x32x16 MFMA fragment reads;
//  * 256 threads = 2x2 waves, each wave a (BM/2)x(BN/2) grid of v_mfma_f32_32x32x16_bf16 accumulators;
//  * rows beyond M / N are clamped on load (never stored); split-K over blockIdx.z with f32 atomics.
#include "gemm_core.h"
#include "avsr_hip.h"

namespace {

using avsr_gemm_impl::Params;

// CV = 0: plain A[M][K].  CV = 1 / 2: A is the im2col view of a channels-last image tensor (forward / data gradient,
// see gemm_core.h Params); channels are a multiple of 64, so a 64-wide k-tile lies inside one filter tap and the
// tap decode is wave-uniform; out-of-image taps read a caller-provided page of zeros.
template <int BM, int BN, int STAGES, int CV = 0>
struct FastKernel {
    static constexpr int BK = 64;
    static constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int A_LOADS = BM / 32, B_LOADS = BN / 32;  // wave-instructions per wave per stage
    static constexpr int LPT = A_LOADS + B_LOADS;               // LDS-DMA ops per thread per tile
    static constexpr size_t LDS_BYTES = (size_t)STAGES * STAGE_BYTES;
    static_assert(LDS_BYTES >= (size_t)BM * (BN + 4) * 4, "operand ring must be able to hold the epilogue tile");

    // per-lane decode of the A rows this lane stages (fixed for the whole k loop)
    struct RowInfo {
        long base[A_LOADS];          // element offset of pixel (n, 0, 0, 0) of the gathered tensor
        int y0[A_LOADS], x0[A_LOADS];  // CV1: oh*s-ph, ow*s-pw ; CV2: ih+ph, iw+pw
    };
    static AVSR_DEV RowInfo decode_rows(const Params& p, int m0, int wave, int lane) {
        RowInfo ri;
        const int rsub = lane >> 3;
#pragma unroll
        for (int i = 0; i < A_LOADS; i++) {
            const int m = min(m0 + (wave * A_LOADS + i) * 8 + rsub, p.M - 1);
            const int pix = p.cOH * p.cOW;
            const int n = m / pix, r = m - n * pix;
            const int y = r / p.cOW, x = r - y * p.cOW;
            ri.base[i] = (long)n * p.cH * p.cW * p.cC;
            if (CV == 1) {
                ri.y0[i] = y * p.cS - p.cPH;
                ri.x0[i] = x * p.cS - p.cPW;
            } else {
                ri.y0[i] = y + p.cPH;
                ri.x0[i] = x + p.cPW;
            }
        }
        return ri;
    }

    static AVSR_DEV void issue(const bf16_t* A, const bf16_t* B, int lda, int ldb, int m0, int n0, int M, int N, int k0,
                               char* stage, int wave, int lane, const Params& p, const RowInfo& ri) {
        const int rsub = lane >> 3, pc = lane & 7;
        int kh = 0, kw = 0, cbase = 0;
        if (CV != 0) {  // wave-uniform tap decode of this k-tile
            const int tap = k0 / p.cC;
            cbase = k0 - tap * p.cC;
            kh = tap / p.cKW;
            kw = tap - kh * p.cKW;
        }
#pragma unroll
        for (int i = 0; i < A_LOADS; i++) {
            const int r = (wave * A_LOADS + i) * 8 + rsub;  // row inside the tile
            const int c = pc ^ ((r >> 1) & 7);              // source chunk that lands in physical chunk pc
            const bf16_t* src;
            if (CV == 0) {
                const int gr = min(m0 + r, M - 1);
                src = A + (size_t)gr * lda + k0 + c * 8;
            } else if (CV == 1) {
                const int ih = ri.y0[i] + kh, iw = ri.x0[i] + kw;
                const bool ok = ih >= 0 && ih < p.cH && iw >= 0 && iw < p.cW;
                src = ok ? A + ri.base[i] + ((long)ih * p.cW + iw) * p.cC + cbase + c * 8
                         : reinterpret_cast<const bf16_t*>(p.gate);  // zero page
            } else {
                const int th = ri.y0[i] - kh, tw = ri.x0[i] - kw;
                const int oh = th / p.cS, ow = tw / p.cS;
                const bool ok = th >= 0 && tw >= 0 && oh * p.cS == th && ow * p.cS == tw && oh < p.cH && ow < p.cW;
                src = ok ? A + ri.base[i] + ((long)oh * p.cW + ow) * p.cC + cbase + c * 8
                         : reinterpret_cast<const bf16_t*>(p.gate);
            }
            glds16(src, stage + (wave * A_LOADS + i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; i++) {
            const int r = (wave * B_LOADS + i) * 8 + rsub;
            const int c = pc ^ ((r >> 1) & 7);
            const int gr = min(n0 + r, N - 1);
            glds16(B + (size_t)gr * ldb + k0 + c * 8, stage + A_BYTES + (wave * B_LOADS + i) * 1024);
        }
    }

    static AVSR_DEV bf16x8 frag(const char* base, int r, int chunk) {
        return *reinterpret_cast<const bf16x8*>(base + r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4));
    }

    static AVSR_DEV void run(const Params& p, char* smem) {
        const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
        const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave >> 1, wn = wave & 1;
        const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
        const int zs = blockIdx.z;
        const int kbeg = zs * p.k_chunk;
        const int kend = min(p.K, kbeg + p.k_chunk);
        const int nt = (kend - kbeg) / BK;

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

        RowInfo ri;
        if (CV != 0) ri = decode_rows(p, m0, wave, lane);
        // prologue: tiles 0 .. STAGES-2 in flight
#pragma unroll
        for (int s = 0; s < STAGES - 1; s++)
            if (s < nt) issue(A, B, p.lda, p.ldb, m0, n0, p.M, p.N, kbeg + s * BK, smem + s * STAGE_BYTES, wave, lane, p, ri);

        for (int t = 0; t < nt; t++) {
            // retire tile t: loads of at most STAGES-2 later tiles may stay in flight
            const int later = min(STAGES - 2, nt - 1 - t);
            switch (later) {  // wave-uniform; counts are immediates
                case 0: wait_vmcnt<0>(); break;
                case 1: wait_vmcnt<LPT>(); break;
                case 2: wait_vmcnt<2 * LPT>(); break;
                case 3: wait_vmcnt<3 * LPT>(); break;
                case 4: wait_vmcnt<4 * LPT>(); break;
                default: wait_vmcnt<5 * LPT>(); break;
            }
            block_barrier_raw();  // tile t is in LDS for every wave; everyone is done reading tile t-1's buffer
            if (t + STAGES - 1 < nt)
                issue(A, B, p.lda, p.ldb, m0, n0, p.M, p.N, kbeg + (t + STAGES - 1) * BK,
                      smem + ((t + STAGES - 1) % STAGES) * STAGE_BYTES, wave, lane, p, ri);
            const char* As = smem + (t % STAGES) * STAGE_BYTES;
            const char* Bs = As + A_BYTES;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ks++) {
                const int chunk = ks * 2 + (lane >> 5);
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; i++) fa[i] = frag(As, wm * WM + i * 32 + (lane & 31), chunk);
#pragma unroll
                for (int j = 0; j < TN; j++) fb[j] = frag(Bs, wn * WN + j * 32 + (lane & 31), chunk);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acc[i][j] = mfma32(fa[i], fb[j], acc[i][j]);
            }
        }
        if (CV != 0) {
            Params q = p;
            q.gate = nullptr;  // in conv mode the field carries the zero page, not an activation gate
            avsr_gemm_impl::epilogue_lds<BM, BN, TM, TN>(acc, q, m0, n0, wm * WM, wn * WN, zs, 0, smem);
        } else {
            avsr_gemm_impl::epilogue_lds<BM, BN, TM, TN>(acc, p, m0, n0, wm * WM, wn * WN, zs, 0, smem);
        }
    }
};

template <int BM, int BN, int STAGES, int CV = 0>
__global__ __launch_bounds__(256) void gemm_fast_kernel(Params p) {
    AVSR_DYN_SMEM(smem);
    FastKernel<BM, BN, STAGES, CV>::run(p, smem);
}

template <int BM, int BN, int STAGES, int CV = 0>
void launch_fast(Params& p, int split_k, hipStream_t stream) {
    int kc = (p.K + split_k - 1) / split_k;
    kc = ((kc + 63) / 64) * 64;
    split_k = (p.K + kc - 1) / kc;
    p.k_chunk = kc;
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, split_k), block(256);
    AVSR_LAUNCH((gemm_fast_kernel<BM, BN, STAGES, CV>), grid, block, (FastKernel<BM, BN, STAGES, CV>::LDS_BYTES), stream, p);
}

}  // namespace

// tile: 0 auto, 1 = 64x64, 2 = 128x64, 3 = 128x128
extern "C" int avsr_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias,
                                 int act, const void* gate, int gate_dtype, int ldg, float gate_scale, float drop_p,
                                 uint64_t seed, const uint64_t* seed_dev, float alpha, const float* alpha_dev,
                                 const void* resid, int resid_dtype, int ldr, void* C, int c_dtype, int ldc, int accumulate,
                                 int split_k, int tile, hipStream_t stream) {
    AVSR_REQUIRE(K > 0 && K % 64 == 0, "gemm_bf16_nt: K must be a positive multiple of 64");
    AVSR_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "gemm_bf16_nt: lda/ldb must be multiples of 8 elements");
    AVSR_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm_bf16_nt: operands must be 16-byte aligned");
    AVSR_REQUIRE(!(accumulate && c_dtype != 0), "gemm_bf16_nt: accumulate needs an f32 output");
    AVSR_REQUIRE(!(split_k > 1 && !accumulate), "gemm_bf16_nt: split-K needs accumulate=1");
    if (M <= 0 || N <= 0) return 0;
    Params p{};
    p.A = A; p.B = B; p.lda = lda; p.ldb = ldb;
    p.M = M; p.N = N; p.K = K;
    p.bias = bias; p.act = act;
    p.gate = gate; p.gate_dtype = gate_dtype; p.ldg = ldg; p.gate_scale = gate_scale;
    p.drop_p = drop_p; p.seed = seed; p.seed_dev = seed_dev;
    p.alpha = alpha; p.alpha_dev = alpha_dev;
    p.resid = reinterpret_cast<const float*>(resid); p.resid_dtype = resid_dtype; p.ldr = ldr;
    p.C = C; p.c_dtype = c_dtype; p.ldc = ldc; p.accumulate = accumulate;
    p.nsplit = 1; p.batch_h = 1; p.nbatch = 1;
    if (split_k < 1) split_k = 1;
    if (tile == 0) {
        const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        const long t12864 = (long)((M + 127) / 128) * ((N + 63) / 64);
        tile = t128 >= 224 ? 3 : (t12864 >= 200 ? 2 : 1);
    }
    if (tile == 3) launch_fast<128, 128, 3>(p, split_k, stream);
    else if (tile == 2) launch_fast<128, 64, 3>(p, split_k, stream);
    // 3 stages = 48 KiB -> 3 blocks/CU.  Measured: deeper rings (4-6 stages) are slower -- the LDS-DMA path wants more
    // co-resident waves issuing, not more bytes in flight per wave.
    else launch_fast<64, 64, 3>(p, split_k, stream);
    AVSR_CHECK_LAUNCH("gemm_bf16_nt");
    return 0;
}

// bf16 implicit-GEMM convolution on the tuned kernel: forward (dgrad = 0: x[N,H,W,Cin] * wp[Cout][KH][KW][Cin] ->
// y[N,OH,OW,Cout]) or data gradient (dgrad = 1: dy[N,OH,OW,Cout] * wpd[Cin][KH][KW][Cout] (+resid) -> dx[N,H,W,Cin]).
// The gathered tensor's channel count must be a multiple of 64; zero_page: >= 16 zero bytes in device memory.
extern "C" int avsr_conv2d_bf16(int dgrad, const void* src, const void* wp, const void* resid, void* out, const void* zero_page,
                                int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_h, int pad_w,
                                hipStream_t stream) {
    const int OH = (H + 2 * pad_h - KH) / stride + 1, OW = (W + 2 * pad_w - KW) / stride + 1;
    const int Cg = dgrad ? Cout : Cin;  // channels of the gathered tensor
    AVSR_REQUIRE(Cg % 64 == 0, "conv2d_bf16: gathered channel count must be a multiple of 64");
    AVSR_REQUIRE(zero_page != nullptr, "conv2d_bf16: zero page required");
    if (N <= 0) return 0;
    Params p{};
    p.A = src; p.B = wp;
    p.K = KH * KW * Cg; p.lda = Cg; p.ldb = p.K;
    p.alpha = 1.f; p.gate_scale = 1.f;
    p.nsplit = 1; p.batch_h = 1; p.nbatch = 1;
    p.gate = zero_page;
    p.c_dtype = 1; p.C = out;
    p.cKH = KH; p.cKW = KW; p.cS = stride; p.cPH = pad_h; p.cPW = pad_w; p.cC = Cg; p.cT = 1; p.cKT = 1;
    if (!dgrad) {
        p.M = N * OH * OW; p.N = Cout; p.ldc = Cout;
        p.cH = H; p.cW = W; p.cOH = OH; p.cOW = OW;
    } else {
        p.M = N * H * W; p.N = Cin; p.ldc = Cin;
        p.cH = OH; p.cW = OW; p.cOH = H; p.cOW = W;
        p.resid = reinterpret_cast<const float*>(resid); p.resid_dtype = 1; p.ldr = Cin;
    }
    const bool wide = p.N >= 128;
    if (!dgrad) {
        if (wide) launch_fast<128, 128, 3, 1>(p, 1, stream); else launch_fast<128, 64, 3, 1>(p, 1, stream);
    } else {
        if (wide) launch_fast<128, 128, 3, 2>(p, 1, stream); else launch_fast<128, 64, 3, 2>(p, 1, stream);
    }
    AVSR_CHECK_LAUNCH("conv2d_bf16");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// dst[c][r] = src[r][c] as bf16, dst row pitch ldd >= R (columns [R, ldd) zero-filled): the k-contiguous copies
// (W^T for the data gradient, dY^T / x^T for the weight gradient) that bring every contraction into NT form.
namespace {
// 64x64 tile: each thread loads 2 x 8 consecutive source columns (16/32-byte loads), applies v = alpha*dropout(src),
// optionally writes the bf16 copy (dst), parks the values transposed in a bf16 LDS tile (pitch 72) from which 2 x 8
// consecutive destination elements are written with 16-byte stores (dstT), and optionally adds the tile's column
// sums into colsum (bias gradients) -- the whole "backward prologue" of a Linear layer in one pass over dY.
template <class T>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const T* __restrict__ src, long lds_, bf16_t* __restrict__ dst,
                                                             bf16_t* __restrict__ dstT, long ldd, float* __restrict__ colsum,
                                                             int R, int Ccols, float alpha0, const float* alpha_dev,
                                                             float drop_p, uint64_t seed0, const uint64_t* seed_dev) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 72];  // tile[c][r]
    __shared__ float csum[4][64];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const bool vec_ok = (lds_ % 8 == 0) && (Ccols % 8 == 0);
    const float alpha = alpha0 * (alpha_dev ? *alpha_dev : 1.f);
    const uint64_t seed = seed0 + (seed_dev ? *seed_dev : 0ull);
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    float part[8];
#pragma unroll
    for (int e = 0; e < 8; e++) part[e] = 0.f;
    // thread (rsub = id>>3 in 0..31, chunk = id&7) handles rows rsub and rsub+32 of the tile, columns chunk*8..+8
    const int cc = (threadIdx.x & 7) * 8;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int r = (threadIdx.x >> 3) + 32 * half;
        float v[8];
        const int gr = r0 + r, gc = c0 + cc;
        if (gr < R && vec_ok && gc + 8 <= Ccols) {
            load8(src + (long)gr * lds_ + gc, v);
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = (gr < R && gc + e < Ccols) ? Elem<T>::ld(src + (long)gr * lds_ + gc + e) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            v[e] *= alpha * dropout_scale(seed, (uint64_t)gr * (uint64_t)Ccols + gc + e, drop_p, inv_keep);
            const bf16_t q = f2bf(v[e]);
            tile[(cc + e) * 72 + r] = q;
            part[e] += bf2f(q);
        }
        if (dst && gr < R) {
            if (vec_ok && gc + 8 <= Ccols) store8(dst + (long)gr * Ccols + gc, v);
            else
                for (int e = 0; e < 8; e++)
                    if (gc + e < Ccols) dst[(long)gr * Ccols + gc + e] = f2bf(v[e]);
        }
    }
    if (colsum) {
        // reduce the 32 row-lanes sharing a column chunk: lanes l, l+8, ... within a wave, then across the 4 waves
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float t = part[e];
            t += __shfl_xor(t, 8);
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            part[e] = t;
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane < 8)
#pragma unroll
            for (int e = 0; e < 8; e++) csum[wave][lane * 8 + e] = part[e];
    }
    __syncthreads();
    if (colsum && threadIdx.x < 64) {
        const int gc = c0 + threadIdx.x;
        if (gc < Ccols) atomicAdd(colsum + gc, csum[0][threadIdx.x] + csum[1][threadIdx.x] + csum[2][threadIdx.x] + csum[3][threadIdx.x]);
    }
    if (dstT) {
        for (int id = threadIdx.x; id < 64 * 8; id += 256) {
            const int c = id >> 3, rr = (id & 7) * 8;
            const int gc = c0 + c, gr = r0 + rr;  // dst row gc, dst cols gr..gr+7 (ldd % 8 == 0, gr % 8 == 0)
            if (gc < Ccols && gr < ldd)
                *reinterpret_cast<bf16x8*>(dstT + (long)gc * ldd + gr) = *reinterpret_cast<const bf16x8*>(tile + c * 72 + rr);
        }
    }
}
}  // namespace

extern "C" int avsr_transpose_cast(const void* src, int src_dtype, int64_t ld_src, void* dst, int64_t ld_dst, int R, int C,
                                   hipStream_t stream) {
    AVSR_REQUIRE(ld_dst >= R && ld_dst % 8 == 0, "transpose_cast: destination pitch must cover the source rows and be a multiple of 8");
    if (R <= 0 || C <= 0) return 0;
    dim3 grid((C + 63) / 64, (unsigned)((ld_dst + 63) / 64)), block(256);
    if (src_dtype == 0)
        AVSR_LAUNCH((transpose_cast_kernel<float>), grid, block, 0, stream, (const float*)src, (long)ld_src, (bf16_t*)nullptr, (bf16_t*)dst,
                    (long)ld_dst, (float*)nullptr, R, C, 1.f, (const float*)nullptr, 0.f, 0ull, (const uint64_t*)nullptr);
    else
        AVSR_LAUNCH((transpose_cast_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)src, (long)ld_src, (bf16_t*)nullptr, (bf16_t*)dst,
                    (long)ld_dst, (float*)nullptr, R, C, 1.f, (const float*)nullptr, 0.f, 0ull, (const uint64_t*)nullptr);
    AVSR_CHECK_LAUNCH("transpose_cast");
    return 0;
}

// One pass over a [R][C] matrix (f32 or bf16, row pitch ld_src): v = alpha * dropout(src);
//   dst  (bf16 [R][C], may be NULL)          = v
//   dstT (bf16 [C][ld_dstT], may be NULL)    = v^T, columns [R, ld_dstT) zero
//   colsum (f32 [C], may be NULL)           += column sums of the bf16-rounded v   (bias gradient)
extern "C" int avsr_cast_transpose_colsum(const void* src, int src_dtype, int64_t ld_src, void* dst, void* dstT,
                                          int64_t ld_dstT, float* colsum, int R, int C, float alpha, const float* alpha_dev,
                                          float drop_p, uint64_t seed, const uint64_t* seed_dev, hipStream_t stream) {
    AVSR_REQUIRE(dstT == nullptr || (ld_dstT >= R && ld_dstT % 8 == 0), "cast_transpose_colsum: bad transposed pitch");
    if (R <= 0 || C <= 0) return 0;
    const long rows_cover = dstT ? ld_dstT : R;
    dim3 grid((C + 63) / 64, (unsigned)((rows_cover + 63) / 64)), block(256);
    if (src_dtype == 0)
        AVSR_LAUNCH((transpose_cast_kernel<float>), grid, block, 0, stream, (const float*)src, (long)ld_src, (bf16_t*)dst, (bf16_t*)dstT,
                    (long)ld_dstT, colsum, R, C, alpha, alpha_dev, drop_p, seed, seed_dev);
    else
        AVSR_LAUNCH((transpose_cast_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)src, (long)ld_src, (bf16_t*)dst, (bf16_t*)dstT,
                    (long)ld_dstT, colsum, R, C, alpha, alpha_dev, drop_p, seed, seed_dev);
    AVSR_CHECK_LAUNCH("cast_transpose_colsum");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-tensor weight preparation: ONE launch casts every registered f32 weight [R][C] to its bf16 copy ([R][C]) and/or
// its transposed bf16 copy ([C][ldT], zero tail) -- what the optimizer step would otherwise trigger as ~300 tiny
// launches per training step.  The table lives in device memory (addresses of parameters are stable).
struct AvsrCastEntry {
    const float* src;
    bf16_t* dst;   // may be null
    bf16_t* dstT;  // may be null
    int R, C, ldT, blk0, tiles_c, pad0, pad1, pad2;
};

namespace {
__global__ __launch_bounds__(256) void multi_cast_transpose_kernel(const AvsrCastEntry* __restrict__ table, int n) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 72];
    // binary search: last entry with blk0 <= blockIdx.x
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const AvsrCastEntry e = table[lo];
    const int local = blockIdx.x - e.blk0;
    const int r0 = (local / e.tiles_c) * 64, c0 = (local % e.tiles_c) * 64;
    const bool vec_ok = (e.C % 8 == 0);
    const int cc = (threadIdx.x & 7) * 8;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int r = (threadIdx.x >> 3) + 32 * half;
        const int gr = r0 + r, gc = c0 + cc;
        float v[8];
        if (gr < e.R && vec_ok && gc + 8 <= e.C) {
            load8(e.src + (long)gr * e.C + gc, v);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = (gr < e.R && gc + k < e.C) ? e.src[(long)gr * e.C + gc + k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) tile[(cc + k) * 72 + r] = f2bf(v[k]);
        if (e.dst && gr < e.R) {
            if (vec_ok && gc + 8 <= e.C) store8(e.dst + (long)gr * e.C + gc, v);
            else
                for (int k = 0; k < 8; k++)
                    if (gc + k < e.C) e.dst[(long)gr * e.C + gc + k] = f2bf(v[k]);
        }
    }
    __syncthreads();
    if (e.dstT) {
        for (int id = threadIdx.x; id < 64 * 8; id += 256) {
            const int c = id >> 3, rr = (id & 7) * 8;
            const int gc = c0 + c, gr = r0 + rr;
            if (gc < e.C && gr < e.ldT)
                *reinterpret_cast<bf16x8*>(e.dstT + (long)gc * e.ldT + gr) = *reinterpret_cast<const bf16x8*>(tile + c * 72 + rr);
        }
    }
}
}  // namespace

// table: n entries of 64 bytes {src, dst, dstT, R, C, ldT, blk0, tiles_c, 0, 0, 0}; blk0 = running sum of
// ceil(max(R, ldT)/64) * ceil(C/64); total_blocks = the final sum
extern "C" int avsr_multi_cast_transpose(const void* table, int n, int total_blocks, hipStream_t stream) {
    if (n <= 0 || total_blocks <= 0) return 0;
    AVSR_LAUNCH(multi_cast_transpose_kernel, dim3(total_blocks), dim3(256), 0, stream,
                reinterpret_cast<const AvsrCastEntry*>(table), n);
    AVSR_CHECK_LAUNCH("multi_cast_transpose");
    return 0;
}
