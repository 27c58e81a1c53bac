// gemm_tn_fast.hip -- bf16 weight-gradient contractions without transposed copies:
//     C[M][N] (+)= sum_k A[k][m] * B[k][n]        A: [K][lda], B: [K][ldb], both bf16, m / n contiguous
// (dW = dY^T x for every Linear / pointwise conv, and -- with B gathered through the im2col index map -- the
// convolution weight gradients).  The contraction index is the SLOW dimension of both operands, so tiles are staged
// k-major in LDS exactly as they lie in HBM (LDS-DMA, 128-byte rows, 3-stage ring, counted vmcnt as in
// gemm_fast.hip) and the MFMA fragments are fetched with the CDNA4 LDS transpose read ds_read_b64_tr_b16.
// Rows k >= K and columns beyond M / N are redirected to a caller-provided page of zeros; split-K over blockIdx.z
// with f32 atomics.
#include "gemm_tn_kernel.h"
#include "gemm_pair.h"
#include "avsr_hip.h"

namespace {

using avsr_gemm_impl::Params;

using avsr_tn::TnKernel;

template <int STAGES, int CV>
__global__ __launch_bounds__(256) void gemm_tn_fast_kernel(Params p) {
    AVSR_DYN_SMEM(smem);
    TnKernel<STAGES, CV>::run(p, smem);
}

template <int CV>
void launch_tn(Params& p, int split_k, hipStream_t stream) {
    if (split_k < 1) split_k = 1;
    int kc = (p.K + split_k - 1) / split_k;
    kc = ((kc + 63) / 64) * 64;
    split_k = (p.K + kc - 1) / kc;
    p.k_chunk = kc;
    dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, split_k), block(256);
    AVSR_LAUNCH((gemm_tn_fast_kernel<3, CV>), grid, block, (TnKernel<3, CV>::LDS_BYTES), stream, p);
}

}  // namespace

// C[M][N] (f32; accumulate: atomicAdd into, else overwrite -- split_k > 1 needs accumulate) = sum_k A[k][m] B[k][n]
// colsum_a (f32 [M], may be NULL; accumulated into): += sum_k A[k][m]
extern "C" int avsr_gemm_bf16_tn(const void* A, int lda, const void* B, int ldb, int M, int N, int K, float* C, int ldc,
                                 int accumulate, int split_k, const void* zero_page, float* colsum_a, hipStream_t stream) {
    AVSR_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && M % 8 == 0 && N % 8 == 0, "gemm_bf16_tn: M, N, lda, ldb must be multiples of 8");
    AVSR_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm_bf16_tn: operands must be 16-byte aligned");
    AVSR_REQUIRE(zero_page != nullptr, "gemm_bf16_tn: zero page required");
    AVSR_REQUIRE(!(split_k > 1 && !accumulate), "gemm_bf16_tn: split-K needs accumulate=1");
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    Params p{};
    p.A = A; p.B = B; p.lda = lda; p.ldb = ldb; p.M = M; p.N = N; p.K = K;
    p.alpha = 1.f; p.gate_scale = 1.f; p.gate = zero_page;
    p.C = C; p.c_dtype = 0; p.ldc = ldc; p.accumulate = accumulate;
    p.colsum_a = colsum_a;
    p.nsplit = 1; p.batch_h = 1; p.nbatch = 1;
    if (avsr_det()) {
        // deterministic mode: one block per weight-gradient tile walks all of K in order (no k split, no pairing); the bias gradient
        // from an ordered column-sum pass over A
        p.colsum_a = nullptr;
        launch_tn<0>(p, 1, stream);
        if (colsum_a) avsr_colsum_det(A, 1, lda, K, M, colsum_a, stream);
        AVSR_CHECK_LAUNCH("gemm_bf16_tn");
        return 0;
    }
    if (avsr_pair::stash_tn(p, split_k, stream)) return 0;  // launched by avsr_gemm_pair_end (gemm_pair.hip)
    launch_tn<0>(p, split_k, stream);
    AVSR_CHECK_LAUNCH("gemm_bf16_tn");
    return 0;
}

// dwp[Cout][KH][KW][Cin] (f32, caller zeroes) += dy[N,OH,OW,Cout]^T im2col(x[N,H,W,Cin]); bf16, Cin % 64 == 0
extern "C" int avsr_conv2d_wgrad_bf16(const void* dy, const void* x, float* dwp, const void* zero_page, int N, int H, int W,
                                      int Cin, int Cout, int KH, int KW, int stride, int pad_h, int pad_w,
                                      hipStream_t stream) {
    AVSR_REQUIRE(Cin % 64 == 0 && Cout % 8 == 0, "conv2d_wgrad_bf16: Cin must be a multiple of 64, Cout of 8");
    AVSR_REQUIRE(zero_page != nullptr, "conv2d_wgrad_bf16: zero page required");
    const int OH = (H + 2 * pad_h - KH) / stride + 1, OW = (W + 2 * pad_w - KW) / stride + 1;
    if (N <= 0) return 0;
    Params p{};
    p.A = dy; p.B = x; p.lda = Cout; p.ldb = Cin;
    p.M = Cout; p.N = KH * KW * Cin; p.K = N * OH * OW;
    p.alpha = 1.f; p.gate_scale = 1.f; p.gate = zero_page;
    p.C = dwp; p.c_dtype = 0; p.ldc = p.N; p.accumulate = 1;
    p.nsplit = 1; p.batch_h = 1; p.nbatch = 1;
    p.cH = H; p.cW = W; p.cC = Cin; p.cOH = OH; p.cOW = OW; p.cKH = KH; p.cKW = KW; p.cS = stride; p.cPH = pad_h; p.cPW = pad_w;
    p.cT = 1; p.cKT = 1;
    const long tiles = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
    // target block count (knob 24: A/B).  Measured on the audio trunk's Conv1d shapes (tools/microbench_wgrad1d.py ->
    // profiles/r6_microbench_wgrad1d.txt): 3 - 12 tiles 41 -> 35 us at 512 blocks (fewer colliding atomics per element), 192 tiles
    // 173 -> 163 us at 2048
    const long target = avsr_tune_knobs[24] > 0 ? avsr_tune_knobs[24] : (tiles <= 12 ? 512 : (tiles >= 96 ? 2048 : 1024));
    long split = target / (tiles < 1 ? 1 : tiles);
    if (split > p.K / 512) split = p.K / 512;
    if (split < 1 || avsr_det()) split = 1;
    launch_tn<3>(p, (int)split, stream);
    AVSR_CHECK_LAUNCH("conv2d_wgrad_bf16");
    return 0;
}
