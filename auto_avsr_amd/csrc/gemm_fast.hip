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
#include "gemm_fast_kernel.h"
#include "gemm_pair.h"
#include "avsr_hip.h"

namespace {

using avsr_gemm_impl::Params;

// run-time tuning knobs (avsr_tune, common.hip): 0 = conv tile override, 1 = XCD-aware tile order, 2 = ablation
// (benchmarks only)
#define g_tune avsr_tune_knobs

using avsr_fast::FastKernel;

template <int BM, int BN, int STAGES, int CV, int WGM, int WGN, int ABL, int KS, int F16 = 0, int WP = 1>
__global__ __launch_bounds__(64 * WGM * WGN * KS) void gemm_fast_kernel(Params p) {
    AVSR_DYN_SMEM(smem);
    FastKernel<BM, BN, STAGES, CV, WGM, WGN, ABL, KS, F16, WP>::run(p, smem);
}

template <int BM, int BN, int STAGES, int CV = 0, int WGM = 2, int WGN = 2, int ABL = 0, int KS = 1, int F16 = 0, int WP = 1>
void launch_fast(Params& p, int split_k, hipStream_t stream) {
    using K = FastKernel<BM, BN, STAGES, CV, WGM, WGN, ABL, KS, F16, WP>;
    // XCD-aware tile order (knob 1: 0 = automatic, 1 = on, 2 = off).  Automatic: on for the 64x64 GEMM tile -- measured
    // with operands that are NOT cache-resident (tools/microbench_xcd.py: A written by the previous kernel, weights
    // streamed from HBM, as in the training step): every XCD otherwise pulls the whole problem through its own L2
    // (rocprofv3 FETCH_SIZE ~8x the operand bytes); 1500x768x768 10.5 -> 9.4 us, 1500x768x3072 28.5 -> 22.6 us.
    // Neutral-to-slower for the 128-row tiles with wide N and for operands that sit in the Infinity Cache.
    p.k_rot = CV == 0 ? g_tune[7] : 0;
    p.xcd_order = g_tune[1] == 0 ? (CV == 0 && BM == 64 && BN == 64) : (g_tune[1] == 1);
    int gy;
    if (CV == 0) {
        int kc = (p.K + split_k - 1) / split_k;
        kc = ((kc + 63) / 64) * 64;
        split_k = (p.K + kc - 1) / kc;
        p.k_chunk = kc;
        gy = (p.M + BM - 1) / BM;
    } else {
        split_k = 1;
        int t0 = 0;
        for (int c = 0; c < p.ncls; c++) {
            p.cls_tile0[c] = t0;
            t0 += (int)(((long)p.cN * p.cls_h[c] * p.cls_w[c] + BM - 1) / BM);
        }
        p.cls_tile0[p.ncls] = t0;
        gy = t0;
    }
    dim3 grid((p.N + BN - 1) / BN, gy, split_k), block(K::NTHR);
    AVSR_LAUNCH((gemm_fast_kernel<BM, BN, STAGES, CV, WGM, WGN, ABL, KS, F16, WP>), grid, block, K::LDS_BYTES, stream, p);
}

// tile codes shared by the GEMM and convolution entry points
//   1 = 64x64 (3 stages, 4 waves)   2 = 128x64   3 = 128x128   4 = 128x128 with 2 stages (2 blocks per CU)
//   5 = 256x128, 8 waves of 64x64   6 = 256x128, 4 waves of 128x64   7 = 128x64 with 2 stages   8 = 256x64, 8 waves
//   9 / 10 = 64x64 with 5 / 8 stages   11 / 12 = 128x64 with 4 / 6 stages   13 = 128x128 with 4 stages (plain GEMM only)
template <int CV>
bool launch_tile(int tile, Params& p, int split_k, hipStream_t stream) {
    switch (tile) {
        case 1: if (CV == 0) { launch_fast<64, 64, 3, 0>(p, split_k, stream); return true; } return false;
        case 2: launch_fast<128, 64, 3, CV>(p, split_k, stream); return true;
        case 3:
            if (CV == 1 && g_tune[2] == 1) launch_fast<128, 128, 3, 1, 2, 2, 1>(p, split_k, stream);
            else if (CV == 1 && g_tune[2] == 2) launch_fast<128, 128, 3, 1, 2, 2, 2>(p, split_k, stream);
            else launch_fast<128, 128, 3, CV>(p, split_k, stream);
            return true;
        case 4: launch_fast<128, 128, 2, CV>(p, split_k, stream); return true;
        case 5: launch_fast<256, 128, 3, CV, 4, 2>(p, split_k, stream); return true;
        case 6: launch_fast<256, 128, 3, CV, 2, 2>(p, split_k, stream); return true;
        case 7: launch_fast<128, 64, 2, CV>(p, split_k, stream); return true;
        case 8: launch_fast<256, 64, 3, CV, 4, 2>(p, split_k, stream); return true;
        // deeper rings for the skinny M = B*T GEMMs (one block per CU anyway: bytes in flight per CU = ring depth)
        case 9: if (CV == 0) { launch_fast<64, 64, 5, 0>(p, split_k, stream); return true; } return false;
        case 10: if (CV == 0) { launch_fast<64, 64, 8, 0>(p, split_k, stream); return true; } return false;
        case 11: if (CV == 0) { launch_fast<128, 64, 4, 0>(p, split_k, stream); return true; } return false;
        case 12: if (CV == 0) { launch_fast<128, 64, 6, 0>(p, split_k, stream); return true; } return false;
        case 13: if (CV == 0) { launch_fast<128, 128, 4, 0>(p, split_k, stream); return true; } return false;
        // ablations of the 64x64 GEMM tile (benchmarks only): 14 = no LDS reads / MFMA, 15 = no operand loads after the prologue
        case 14: if (CV == 0) { launch_fast<64, 64, 3, 0, 2, 2, 1>(p, split_k, stream); return true; } return false;
        case 15: if (CV == 0) { launch_fast<64, 64, 3, 0, 2, 2, 2>(p, split_k, stream); return true; } return false;
        // 8 waves = two k groups on one output tile (two waves per SIMD for the one-block-per-CU GEMMs): 16 = 64x64, 17 = 128x64,
        // 18 = 128x128
        case 16: if (CV == 0) { launch_fast<64, 64, 3, 0, 2, 2, 0, 2>(p, split_k, stream); return true; } return false;
        case 17: if (CV == 0) { launch_fast<128, 64, 3, 0, 2, 2, 0, 2>(p, split_k, stream); return true; } return false;
        case 18: if (CV == 0) { launch_fast<128, 128, 3, 0, 2, 2, 0, 2>(p, split_k, stream); return true; } return false;
        default: return false;
    }
}

// f16 operands (forward pass of the mixed mode): the three default tile shapes only
template <int CV>
bool launch_tile_h16(int tile, Params& p, int split_k, hipStream_t stream) {
    switch (tile) {
        case 1: if (CV == 0) { launch_fast<64, 64, 3, 0, 2, 2, 0, 1, 1>(p, split_k, stream); return true; } return false;
        case 4: launch_fast<128, 128, 2, CV, 2, 2, 0, 1, 1>(p, split_k, stream); return true;
        case 7: launch_fast<128, 64, 2, CV, 2, 2, 0, 1, 1>(p, split_k, stream); return true;
        default: return false;
    }
}

// f16 operands with two weight planes (Params::B2).  LDS per block: 64x64 / 3 stages 72 KB (2 blocks per CU) or 2 stages 48 KB
// (3 per CU, code 21); 128x64 / 2 stages 64 KB (2 per CU): the A tile is shared by both planes, so the 128-row tile moves as
// many operand bytes per output as the one-plane 64x64 tile; 128x128 / 2 stages 96 KB (1 per CU, code 4)
template <int CV>
bool launch_tile_h16x2(int tile, Params& p, int split_k, hipStream_t stream) {
    switch (tile) {
        case 1: if (CV == 0) { launch_fast<64, 64, 3, 0, 2, 2, 0, 1, 1, 2>(p, split_k, stream); return true; } return false;
        case 21: if (CV == 0) { launch_fast<64, 64, 2, 0, 2, 2, 0, 1, 1, 2>(p, split_k, stream); return true; } return false;
        case 4: launch_fast<128, 128, 2, CV, 2, 2, 0, 1, 1, 2>(p, split_k, stream); return true;
        case 7: launch_fast<128, 64, 2, CV, 2, 2, 0, 1, 1, 2>(p, split_k, stream); return true;
        case 2: launch_fast<128, 64, 3, CV, 2, 2, 0, 1, 1, 2>(p, split_k, stream); return true;
        // 8 waves (two per SIMD), one block per CU: half (5) / three quarters (8, 22) of the operand bytes per MFMA of the 128x64 tile
        case 5: launch_fast<256, 128, 2, CV, 4, 2, 0, 1, 1, 2>(p, split_k, stream); return true;
        case 8: launch_fast<256, 64, 2, CV, 4, 2, 0, 1, 1, 2>(p, split_k, stream); return true;
        case 22: launch_fast<128, 128, 2, CV, 2, 4, 0, 1, 1, 2>(p, split_k, stream); return true;
        case 23: launch_fast<256, 64, 3, CV, 4, 2, 0, 1, 1, 2>(p, split_k, stream); return true;
        // ablations of the three default two-plane GEMM tiles (benchmarks only, wrong results): 31 / 33 / 35 = operand stream only (no
        // LDS reads, no MFMA) of the 64x64 / 128x64 / 256x128 tile, 32 / 34 / 36 = LDS reads + MFMA only (no operand loads after the prologue)
        case 31: if (CV == 0) { launch_fast<64, 64, 3, 0, 2, 2, 1, 1, 1, 2>(p, split_k, stream); return true; } return false;
        case 32: if (CV == 0) { launch_fast<64, 64, 3, 0, 2, 2, 2, 1, 1, 2>(p, split_k, stream); return true; } return false;
        case 33: if (CV == 0) { launch_fast<128, 64, 2, 0, 2, 2, 1, 1, 1, 2>(p, split_k, stream); return true; } return false;
        case 34: if (CV == 0) { launch_fast<128, 64, 2, 0, 2, 2, 2, 1, 1, 2>(p, split_k, stream); return true; } return false;
        case 35: if (CV == 0) { launch_fast<256, 128, 2, 0, 4, 2, 1, 1, 1, 2>(p, split_k, stream); return true; } return false;
        case 36: if (CV == 0) { launch_fast<256, 128, 2, 0, 4, 2, 2, 1, 1, 2>(p, split_k, stream); return true; } return false;
        default: return false;
    }
}

}  // namespace

int avsr_conv_patch_supported(int N, int H, int W, int Cg, int Cout_eff, int KH, int KW, int stride, int pad_h, int pad_w);
int avsr_conv_patch_launch(int mode, const void* src, const void* w, const void* w_lo, int ldw, const void* resid, void* out, void* out2,
                           const void* zero_page, int N, int H, int W, int Cg, int Cout_eff, hipStream_t stream);
int avsr_conv3x3_c64_supported(int H, int W);
int avsr_conv3x3_c64_launch(int flip, const void* src, const void* wq, const void* resid, void* out, const void* zero_page, int N,
                            int H, int W, hipStream_t stream);

static int gemm16_nt_impl(int f16, const void* A, int lda, const void* B, const void* B_lo, int ldb, int M, int N, int K, const float* bias,
                          int act, const void* gate, int gate_dtype, int ldg, float gate_scale, float drop_p,
                          uint64_t seed, const uint64_t* seed_dev, float alpha, const float* alpha_dev,
                          const void* resid, int resid_dtype, int ldr, void* C, int c_dtype, int ldc, int accumulate,
                          int split_k, int tile, float* colsum, void* c2, int ldc2, hipStream_t stream) {
    AVSR_REQUIRE(!(colsum && accumulate), "gemm_bf16_nt: colsum needs a non-accumulating output");
    AVSR_REQUIRE(K > 0 && K % 64 == 0, "gemm_bf16_nt: K must be a positive multiple of 64");
    AVSR_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "gemm_bf16_nt: lda/ldb must be multiples of 8 elements");
    AVSR_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm_bf16_nt: operands must be 16-byte aligned");
    AVSR_REQUIRE(!(accumulate && c_dtype != 0), "gemm_bf16_nt: accumulate needs an f32 output");
    AVSR_REQUIRE(!(split_k > 1 && !accumulate), "gemm_bf16_nt: split-K needs accumulate=1");
    if (M <= 0 || N <= 0) return 0;
    Params p{};
    p.A = A; p.B = B; p.B2 = B_lo; p.lda = lda; p.ldb = ldb;
    p.M = M; p.N = N; p.K = K;
    p.bias = bias; p.act = act;
    p.gate = gate; p.gate_dtype = gate_dtype; p.ldg = ldg; p.gate_scale = gate_scale;
    p.drop_p = drop_p; p.seed = seed; p.seed_dev = seed_dev;
    p.alpha = alpha; p.alpha_dev = alpha_dev;
    p.resid = reinterpret_cast<const float*>(resid); p.resid_dtype = resid_dtype; p.ldr = ldr;
    p.C = C; p.c_dtype = c_dtype; p.ldc = ldc; p.accumulate = accumulate;
    p.colsum = colsum;
    AVSR_REQUIRE(c2 == nullptr || (!accumulate && c_dtype != 1), "gemm_*_nt: the bf16 twin needs a non-accumulating f32 / f16 output");
    p.C2 = c2; p.ldc2 = ldc2;
    p.nsplit = 1; p.batch_h = 1; p.nbatch = 1;
    if (split_k < 1) split_k = 1;
    // deterministic mode (prims.h avsr_det): no k split (an accumulating output then has ONE block adding to each element), no
    // pairing, and the bias-gradient column sums from an ordered pass over the stored result instead of per-tile atomics
    float* colsum_det = nullptr;
    if (avsr_det()) {
        split_k = 1;
        if (colsum) {
            AVSR_REQUIRE(ldc == N, "gemm_*_nt (deterministic mode): colsum needs a dense output");
            colsum_det = colsum;
            p.colsum = nullptr;
        }
    }
    const bool auto_tile = tile == 0;
    if (tile == 0) {
        const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        const long t12864 = (long)((M + 127) / 128) * ((N + 63) / 64);
        // measured on MI355X (tools/microbench_tiles.py): two or three co-resident blocks per CU beat one block with a
        // deeper ring at every size -- 128x128 / 128x64 with a 2-stage ring (2 / 3 blocks per CU) once the grid fills the
        // chip, 64x64 with 3 stages (3 blocks per CU) for the skinny M = B*T GEMMs of the transformer layers
        tile = t128 >= 1024 ? 4 : (t12864 >= 400 ? 7 : (g_tune[14] > 0 && (long)((M + 63) / 64) * ((N + 63) / 64) > 256 ? g_tune[14] : 1));  // knob 14: A/B of the tile for grids of 257+ 64x64 tiles
    }
    if (f16 && B_lo) {
        AVSR_REQUIRE(((uintptr_t)B_lo % 16) == 0, "gemm_h16_nt: the lo plane must be 16-byte aligned");
        if (auto_tile) {
            // the A tile serves both planes: a 128-row tile once the grid fills the chip with them, else 64x64 (knob 17: A/B)
            const long t12864 = (long)((M + 127) / 128) * ((N + 63) / 64);
            // measured (tools/microbench_h16x2.py, us): 1600x768x768 11.5 (64x64) / 12.9 (128x64); 1600x3072x768 35.2 / 28.2;
            // 1600x768x3072 35.4 / 32.2 with the 3-stage 128x64 ring (a long k loop amortises the fewer, fatter blocks)
            // round 6: 256x128 on 8 waves (two per SIMD, half the operand bytes per MFMA of the 128x64 tile) once the grid has
            // >= 160 of them: 1600x3072x768 29.5 -> 27.6 us, 1600x9216x768 67.6 -> 61.1 (profiles/r6_microbench_h16x2.txt);
            // the 126 tiles of the fused Q/K/V projection are slower (26 vs 19 us)
            const long t256 = (long)((M + 255) / 256) * ((N + 127) / 128);
            // (knob 17 = -1: the round-5 rule, without the 256x128 tile -- same-box A/B runs)
            tile = g_tune[17] > 0 ? g_tune[17] : (t256 >= 160 && g_tune[17] == 0 ? 5 : (t12864 >= 256 ? 7 : (K >= 2048 && t12864 >= 128 ? 2 : 1)));
        }
        AVSR_REQUIRE(launch_tile_h16x2<0>(tile, p, split_k, stream), "gemm_h16_nt: unknown tile code (two weight planes)");
        if (colsum_det) avsr_colsum_det(C, c_dtype, ldc, M, N, colsum_det, stream);
        AVSR_CHECK_LAUNCH("gemm_h16_nt");
        return 0;
    }
    if (f16) {
        if (tile != 1 && tile != 4 && tile != 7) tile = 1;
        AVSR_REQUIRE(launch_tile_h16<0>(tile, p, split_k, stream), "gemm_h16_nt: unknown tile code");
        if (colsum_det) avsr_colsum_det(C, c_dtype, ldc, M, N, colsum_det, stream);
        AVSR_CHECK_LAUNCH("gemm_h16_nt");
        return 0;
    }
    if (!avsr_det() && avsr_pair::stash_nt(p, tile, split_k, stream)) return 0;  // launched by avsr_gemm_pair_end (gemm_pair.hip)
    AVSR_REQUIRE(launch_tile<0>(tile, p, split_k, stream), "gemm_bf16_nt: unknown tile code");
    if (colsum_det) avsr_colsum_det(C, c_dtype, ldc, M, N, colsum_det, stream);
    AVSR_CHECK_LAUNCH("gemm_bf16_nt");
    return 0;
}

extern "C" int avsr_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias,
                                 int act, const void* gate, int gate_dtype, int ldg, float gate_scale, float drop_p,
                                 uint64_t seed, const uint64_t* seed_dev, float alpha, const float* alpha_dev,
                                 const void* resid, int resid_dtype, int ldr, void* C, int c_dtype, int ldc, int accumulate,
                                 int split_k, int tile, float* colsum, hipStream_t stream) {
    return gemm16_nt_impl(0, A, lda, B, nullptr, ldb, M, N, K, bias, act, gate, gate_dtype, ldg, gate_scale, drop_p, seed, seed_dev, alpha,
                          alpha_dev, resid, resid_dtype, ldr, C, c_dtype, ldc, accumulate, split_k, tile, colsum, nullptr, 0, stream);
}

// The same contraction on IEEE-half operands (v_mfma_f32_32x32x16_f16): A and B f16 k-contiguous; C f32, bf16 or f16
// (c_dtype 0 / 1 / 2); c2 (may be NULL): bf16 twin of an f32 / f16 result -- what the backward pass of the mixed mode reads.
// B_lo (may be NULL): the scaled lo plane of the weight, f16((w - B) * 2^11), same pitch -- two MFMAs per product, weight exact.
extern "C" int avsr_gemm_h16_nt(const void* A, int lda, const void* B, const void* B_lo, int ldb, int M, int N, int K, const float* bias,
                                int act, float drop_p, uint64_t seed, const uint64_t* seed_dev, float alpha, const void* resid,
                                int resid_dtype, int ldr, void* C, int c_dtype, int ldc, int tile, void* c2, int ldc2,
                                hipStream_t stream) {
    // knob 26 (probe, tools/microbench_splitk.py): > 1 = k split of that many ways with f32 atomics onto a ZEROED f32 C (no twin)
    const int sk = (g_tune[26] > 1 && c_dtype == 0 && c2 == nullptr) ? g_tune[26] : 1;
    return gemm16_nt_impl(1, A, lda, B, B_lo, ldb, M, N, K, bias, act, nullptr, 0, 0, 1.f, drop_p, seed, seed_dev, alpha, nullptr, resid,
                          resid_dtype, ldr, C, c_dtype, ldc, sk > 1 ? 1 : 0, sk, tile, nullptr, c2, ldc2, stream);
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
    AVSR_REQUIRE(KH * KW <= 32, "conv2d_bf16: at most 32 filter taps");
    AVSR_REQUIRE(stride == 1 || (stride == 2 && KH <= 8 && KW <= 8), "conv2d_bf16: stride must be 1 or 2");
    AVSR_REQUIRE((long)N * H * W < (1l << 31) && (long)N * OH * OW < (1l << 31), "conv2d_bf16: pixel count exceeds int32");
    if (N <= 0) return 0;
    // 64 -> 64 channels, 3x3 / stride 1 / pad 1 (the trunk's first stage): weights in registers, patch-staged persistent kernel
    // (conv3x3_c64.hip); knob 12 = 1 keeps the tiled kernel for A/B runs
    if (Cin == 64 && Cout == 64 && KH == 3 && KW == 3 && stride == 1 && pad_h == 1 && pad_w == 1 && g_tune[12] != 1 &&
        avsr_conv3x3_c64_supported(H, W)) {
        avsr_conv3x3_c64_launch(dgrad, src, wp, resid, out, zero_page, N, H, W, stream);
        AVSR_CHECK_LAUNCH("conv2d_bf16");
        return 0;
    }
    // 128 - 512 channels on small images (stages 2 - 4), 3x3 / stride 1: whole-image tiles with the input PATCH staged once per
    // 64-channel chunk instead of one im2col tile per filter tap (conv_patch.hip, round 6)
    if (g_tune[0] == 0 && avsr_conv_patch_supported(N, H, W, Cg, dgrad ? Cin : Cout, KH, KW, stride, pad_h, pad_w)) {
        avsr_conv_patch_launch(dgrad ? 1 : 0, src, wp, nullptr, 0, resid, out, nullptr, zero_page, N, H, W, Cg, dgrad ? Cin : Cout, stream);
        AVSR_CHECK_LAUNCH("conv2d_bf16");
        return 0;
    }
    Params p{};
    p.A = src; p.B = wp;
    p.K = KH * KW * Cg; p.lda = Cg; p.ldb = p.K;
    p.alpha = 1.f; p.gate_scale = 1.f;
    p.nsplit = 1; p.batch_h = 1; p.nbatch = 1;
    p.gate = zero_page;
    p.c_dtype = 1; p.C = out;
    p.cN = N;
    p.cKH = KH; p.cKW = KW; p.cS = stride; p.cPH = pad_h; p.cPW = pad_w; p.cC = Cg; p.cT = 1; p.cKT = 1;
    if (!dgrad) {
        p.M = N * OH * OW; p.N = Cout; p.ldc = Cout;
        p.cH = H; p.cW = W; p.cOH = OH; p.cOW = OW;
        p.ncls = 1;
        p.cls_h[0] = OH; p.cls_w[0] = OW; p.cls_nkh[0] = KH; p.cls_nkw[0] = KW;
    } else {
        p.M = N * H * W; p.N = Cin; p.ldc = Cin;
        p.cH = OH; p.cW = OW; p.cOH = H; p.cOW = W;
        p.resid = reinterpret_cast<const float*>(resid); p.resid_dtype = 1; p.ldr = Cin;
        // residue classes of (ih + ph, iw + pw) mod stride, heaviest tap list first
        struct Cls { int py, px, y0, x0, h, w, nkh, nkw; } cl[4];
        int nc = 0;
        for (int py = 0; py < stride; py++)
            for (int px = 0; px < stride; px++) {
                Cls c;
                c.py = py; c.px = px;
                c.y0 = ((py - pad_h) % stride + stride) % stride;
                c.x0 = ((px - pad_w) % stride + stride) % stride;
                c.h = c.y0 < H ? (H - c.y0 + stride - 1) / stride : 0;
                c.w = c.x0 < W ? (W - c.x0 + stride - 1) / stride : 0;
                c.nkh = py < KH ? (KH - py + stride - 1) / stride : 0;
                c.nkw = px < KW ? (KW - px + stride - 1) / stride : 0;
                if (c.nkh == 0 || c.nkw == 0) c.nkh = c.nkw = 0;
                if (c.h > 0 && c.w > 0) cl[nc++] = c;
            }
        for (int i = 0; i < nc; i++)
            for (int j = i + 1; j < nc; j++)
                if (cl[j].nkh * cl[j].nkw > cl[i].nkh * cl[i].nkw) { Cls t = cl[i]; cl[i] = cl[j]; cl[j] = t; }
        p.ncls = nc;
        for (int i = 0; i < nc; i++) {
            p.cls_py[i] = cl[i].py; p.cls_px[i] = cl[i].px; p.cls_y0[i] = cl[i].y0; p.cls_x0[i] = cl[i].x0;
            p.cls_h[i] = cl[i].h; p.cls_w[i] = cl[i].w; p.cls_nkh[i] = cl[i].nkh; p.cls_nkw[i] = cl[i].nkw;
        }
    }
    int tile = g_tune[0];
    if (tile == 0) tile = p.N >= 128 ? 4 : 7;  // 2-stage rings: 2 / 3 co-resident blocks per CU (measured best)
    const bool ok = dgrad ? launch_tile<2>(tile, p, 1, stream) : launch_tile<1>(tile, p, 1, stream);
    AVSR_REQUIRE(ok, "conv2d_bf16: unknown tile code");
    AVSR_CHECK_LAUNCH("conv2d_bf16");
    return 0;
}

// f16 forward convolution of the mixed mode on the same tiled kernel (v_mfma_f32_32x32x16_f16): x [N,H,W,Cin] f16,
// wp [Cout][KH][KW][Cin] f16 -> y [N,OH,OW,Cout] f16 and (y2 != NULL) its bf16 twin, which the bf16 backward pass reads.
// ldw: row pitch of wp in elements (0 = KH * KW * Cin); wp_lo (may be NULL): scaled lo plane of the filter, same pitch.
extern "C" int avsr_conv2d_h16(const void* x, const void* wp, const void* wp_lo, int ldw, void* y, void* y2, const void* zero_page, int N, int H, int W, int Cin,
                               int Cout, int KH, int KW, int stride, int pad_h, int pad_w, hipStream_t stream) {
    const int OH = (H + 2 * pad_h - KH) / stride + 1, OW = (W + 2 * pad_w - KW) / stride + 1;
    AVSR_REQUIRE(Cin % 64 == 0, "conv2d_h16: input channel count must be a multiple of 64");
    AVSR_REQUIRE(zero_page != nullptr, "conv2d_h16: zero page required");
    AVSR_REQUIRE(KH * KW <= 32, "conv2d_h16: at most 32 filter taps");
    AVSR_REQUIRE(stride == 1 || (stride == 2 && KH <= 8 && KW <= 8), "conv2d_h16: stride must be 1 or 2");
    AVSR_REQUIRE((long)N * H * W < (1l << 31) && (long)N * OH * OW < (1l << 31), "conv2d_h16: pixel count exceeds int32");
    if (N <= 0) return 0;
    AVSR_REQUIRE((ldw ? ldw : KH * KW * Cin) >= KH * KW * Cin && (ldw % 8) == 0, "conv2d_h16: bad filter pitch");
    if (g_tune[18] == 0 && avsr_conv_patch_supported(N, H, W, Cin, Cout, KH, KW, stride, pad_h, pad_w)) {  // conv_patch.hip (round 6)
        avsr_conv_patch_launch(2, x, wp, wp_lo, ldw, nullptr, y, y2, zero_page, N, H, W, Cin, Cout, stream);
        AVSR_CHECK_LAUNCH("conv2d_h16");
        return 0;
    }
    Params p{};
    p.A = x; p.B = wp; p.B2 = wp_lo;
    p.K = KH * KW * Cin; p.lda = Cin; p.ldb = ldw ? ldw : p.K;
    AVSR_REQUIRE(p.ldb >= p.K && p.ldb % 8 == 0, "conv2d_h16: bad filter pitch");
    p.alpha = 1.f; p.gate_scale = 1.f;
    p.nsplit = 1; p.batch_h = 1; p.nbatch = 1;
    p.gate = zero_page;
    p.c_dtype = 2; p.C = y; p.C2 = y2; p.ldc2 = Cout;
    p.cN = N;
    p.cKH = KH; p.cKW = KW; p.cS = stride; p.cPH = pad_h; p.cPW = pad_w; p.cC = Cin; p.cT = 1; p.cKT = 1;
    p.M = N * OH * OW; p.N = Cout; p.ldc = Cout;
    p.cH = H; p.cW = W; p.cOH = OH; p.cOW = OW;
    p.ncls = 1;
    p.cls_h[0] = OH; p.cls_w[0] = OW; p.cls_nkh[0] = KH; p.cls_nkw[0] = KW;
    if (wp_lo) {
        // two filter planes: 128x64 tiles keep two blocks per CU (64 KB each); knob 18: A/B against 128x128 (one block per CU)
        // round 6: 256x128 tiles on 8 waves (one block per CU, two waves per SIMD) wherever the output is >= 128 channels wide and
        // the grid fills the chip: stage 3 / 4 convolutions 182 -> 159, 164 -> 140, 119 -> 90, 92 -> 80 us
        const long t256 = (long)((p.M + 255) / 256) * ((p.N + 127) / 128);
        const int tile = g_tune[18] > 0 ? g_tune[18] : (p.N % 128 == 0 && t256 >= 200 ? 5 : 7);
        AVSR_REQUIRE(launch_tile_h16x2<1>(tile, p, 1, stream), "conv2d_h16: tile (two filter planes)");
    } else
        AVSR_REQUIRE(launch_tile_h16<1>(p.N >= 128 ? 4 : 7, p, 1, stream), "conv2d_h16: tile");
    AVSR_CHECK_LAUNCH("conv2d_h16");
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
    // deterministic mode: the column sums of the bf16-rounded values from an ordered pass over the copy just written
    float* colsum_det = nullptr;
    if (avsr_det() && colsum) {
        AVSR_REQUIRE(dst != nullptr || dstT != nullptr, "cast_transpose_colsum (deterministic mode): colsum needs one of the copies");
        colsum_det = colsum;
        colsum = nullptr;
    }
    if (src_dtype == 0)
        AVSR_LAUNCH((transpose_cast_kernel<float>), grid, block, 0, stream, (const float*)src, (long)ld_src, (bf16_t*)dst, (bf16_t*)dstT,
                    (long)ld_dstT, colsum, R, C, alpha, alpha_dev, drop_p, seed, seed_dev);
    else
        AVSR_LAUNCH((transpose_cast_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)src, (long)ld_src, (bf16_t*)dst, (bf16_t*)dstT,
                    (long)ld_dstT, colsum, R, C, alpha, alpha_dev, drop_p, seed, seed_dev);
    if (colsum_det) {
        if (dst) avsr_colsum_det(dst, 1, C, R, C, colsum_det, stream);
        else avsr_rowsum_det_bf16(dstT, ld_dstT, C, R, colsum_det, stream);
    }
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
    int R, C, ldT, blk0, tiles_c, limT, pad1, pad2;  // limT (0 = ldT): rows of dstT that are written (zero tail included); pad1 = 2: dst is f16, not bf16
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
            if (e.pad1 == 2) {  // f16 forward copy (mixed mode), [R][2][C]: hi row, scaled lo row (prims.h f2h_lo)
                f16_t* d16 = reinterpret_cast<f16_t*>(e.dst) + (long)gr * 2 * e.C + gc;
                if (vec_ok && gc + 8 <= e.C) {
                    store8(d16, v);
                    store8_lo(d16 + e.C, v);
                } else
                    for (int k = 0; k < 8; k++)
                        if (gc + k < e.C) {
                            d16[k] = f2h(v[k]);
                            d16[e.C + k] = f2h_lo(v[k]);
                        }
            } else if (vec_ok && gc + 8 <= e.C) store8(e.dst + (long)gr * e.C + gc, v);
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
            if (gc < e.C && gr < (e.limT ? e.limT : e.ldT))
                *reinterpret_cast<bf16x8*>(e.dstT + (long)gc * e.ldT + gr) = *reinterpret_cast<const bf16x8*>(tile + c * 72 + rr);
        }
    }
}
}  // namespace

// table: n entries of 64 bytes {src, dst, dstT, R, C, ldT, blk0, tiles_c, limT, dst_dtype (0 / 1 = bf16, 2 = two-plane f16 [R][2][C]), 0}; blk0 = running sum of
// ceil(max(R, limT ? limT : ldT)/64) * ceil(C/64); total_blocks = the final sum.  limT < ldT lets several transposed
// copies share one [C][ldT] buffer side by side (concatenated projection weights).
extern "C" int avsr_multi_cast_transpose(const void* table, int n, int total_blocks, hipStream_t stream) {
    if (n <= 0 || total_blocks <= 0) return 0;
    AVSR_LAUNCH(multi_cast_transpose_kernel, dim3(total_blocks), dim3(256), 0, stream,
                reinterpret_cast<const AvsrCastEntry*>(table), n);
    AVSR_CHECK_LAUNCH("multi_cast_transpose");
    return 0;
}
