// stem.hip -- dedicated kernels for the visual front-end stem  Conv3d(1, 64, (5,7,7), stride (1,2,2), padding (2,3,3))
// (frontend/resnet.py:204-211) in bf16 mode.  C_in = 1 makes the generic im2col gather element-wise; here the 35
// (kt,kh) input rows an output row needs are staged ONCE in LDS (as bf16, zero padded), and the MFMA operands are
// read from that patch: K is re-indexed as (kt*7+kh)*8 + kw with a zero 8th tap, so a lane's 8 consecutive k values
// are 8 consecutive input columns -- four aligned ds_read_b32.
//   forward : one block per output row (n, oh); wave w owns 16 output channels; 3 pixel tiles x 9 k-steps of
//             v_mfma_f32_16x16x32_bf16; weights (64 x 288 bf16) live in registers.
//   wgrad   : persistent blocks loop over output rows, accumulate dW^T tiles in registers (contraction over the 44
//             pixels of a row), write per-block partials, a second kernel sums them (no atomics).
#include "prims.h"
#include "avsr_hip.h"

namespace {

constexpr int KT = 5, KH = 7, KW = 7, ROWS = KT * KH;  // 35 (kt,kh) rows
constexpr int KP = 288;                                // padded K: 36 rows x 8 taps
constexpr int CO = 64;
constexpr int LP = 104;                                // LDS row pitch (bf16): >= 2*47+8, multiple of 8

// wp[co][(kt*7+kh)*8 + kw] = w[co][0][kt][kh][kw]; zero for kw = 7 and rows >= 35
__global__ void stem_weight_kernel(const float* __restrict__ w, bf16_t* __restrict__ wp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= CO * KP) return;
    const int co = i / KP, k = i % KP, r = k >> 3, kw = k & 7;
    wp[i] = (r < ROWS && kw < KW) ? f2bf(w[(co * ROWS + r) * KW + kw]) : (bf16_t)0;
}

// stage the 36 x LP bf16 patch of output row (n, oh): patch[r][3 + iw] = x[b][t+kt-2][2*oh+kh-3][iw]
AVSR_DEV void stage_patch(bf16_t* patch, const float* __restrict__ x, int n, int oh, int T, int H, int W) {
    const int b = n / T, t = n - b * T;
    // zero left/right padding columns and the dummy row 35
    for (int i = threadIdx.x; i < 36 * LP; i += 256) {
        const int r = i / LP, c = i - r * LP;
        if (r == 35 || c < 3 || c >= 3 + W) patch[i] = 0;
    }
    const int nv = W >> 2;  // float4 per row (W % 4 == 0)
    for (int i = threadIdx.x; i < ROWS * nv; i += 256) {
        const int r = i / nv, v = i - r * nv;
        const int kt = r / KH, kh = r - kt * KH;
        const int tt = t + kt - 2, ih = 2 * oh + kh - 3;
        f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tt >= 0 && tt < T && ih >= 0 && ih < H)
            q = *reinterpret_cast<const f32x4*>(x + (((long)b * T + tt) * H + ih) * W + v * 4);
#pragma unroll
        for (int e = 0; e < 4; e++) patch[r * LP + 3 + v * 4 + e] = f2bf(q[e]);
    }
}

AVSR_DEV bf16x8 patch_frag(const bf16_t* patch, int r, int ow) {
    // 8 consecutive columns starting at 2*ow (4-byte aligned): four 32-bit LDS reads
    const uint32_t* p = reinterpret_cast<const uint32_t*>(patch + r * LP + 2 * ow);
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const uint32_t u = p[e];
        f[2 * e] = (short)(u & 0xffff);
        f[2 * e + 1] = (short)(u >> 16);
    }
    return f;
}

__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const bf16_t* __restrict__ wp,
                                                       bf16_t* __restrict__ y, int T, int H, int W, int OH, int OW) {
    __shared__ __attribute__((aligned(16))) bf16_t patch[36 * LP];
    const int row = blockIdx.x;  // (n, oh)
    const int n = row / OH, oh = row - n * OH;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int quad = lane >> 4, lc = lane & 15;
    stage_patch(patch, x, n, oh, T, H, W);
    // this wave's weights: channel 16w + lc, k-step ks: taps [ks*32 + 8*quad, +8)
    bf16x8 fb[9];
#pragma unroll
    for (int ks = 0; ks < 9; ks++)
        fb[ks] = *reinterpret_cast<const bf16x8*>(wp + (16 * w + lc) * KP + ks * 32 + 8 * quad);
    __syncthreads();
    const int ntile = (OW + 15) / 16;
    for (int mt = 0; mt < ntile; mt++) {
        const int ow = min(mt * 16 + lc, OW - 1);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 9; ks++) acc = mfma16(patch_frag(patch, ks * 4 + quad, ow), fb[ks], acc);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int o = mt * 16 + 4 * quad + r;
            if (o < OW) y[(((long)n * OH + oh) * OW + o) * CO + 16 * w + lc] = f2bf(acc[r]);
        }
    }
}

// ---- weight gradient: dWt[k'][co] = sum_pix Xcol[pix][k'] dY[pix][co]; per wave: N-tiles (k' groups of 16) w, w+4, ..
constexpr int DP = 72;  // pitch of the transposed dY tile [co][pixel] (64 pixels + pad)

__global__ __launch_bounds__(256) void stem_wgrad_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                         float* __restrict__ partial, int T, int H, int W, int OH, int OW,
                                                         long total_rows) {
    __shared__ __attribute__((aligned(16))) bf16_t patch[36 * LP];
    __shared__ __attribute__((aligned(16))) bf16_t dyt[CO * DP];  // dyt[co][pixel]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int quad = lane >> 4, lc = lane & 15;
    // output tile (nt, mt): rows = k' in [16 nt, 16 nt + 16), cols = co in [16 mt, +16); wave w: nt = w + 4 j
    f32x4 acc[5][4];
#pragma unroll
    for (int j = 0; j < 5; j++)
#pragma unroll
        for (int m = 0; m < 4; m++) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long row = blockIdx.x; row < total_rows; row += gridDim.x) {
        const int n = (int)(row / OH), oh = (int)(row - (long)n * OH);
        __syncthreads();
        stage_patch(patch, x, n, oh, T, H, W);
        // dY row [OW][64] -> dyt[co][pixel], pixels >= OW zero
        for (int i = threadIdx.x; i < 64 * 8; i += 256) {
            const int pix = i >> 3, cc = (i & 7) * 8;
            bf16x8 v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (pix < OW) v = *reinterpret_cast<const bf16x8*>(dy + (((long)n * OH + oh) * OW + pix) * CO + cc);
#pragma unroll
            for (int e = 0; e < 8; e++) dyt[(cc + e) * DP + pix] = (bf16_t)v[e];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {  // 32 pixels per k-step
            const int p0 = ks * 32 + 8 * quad;  // this lane's 8 consecutive pixels
            bf16x8 fbm[4];                      // B operand: [n = co][k = pixel]
#pragma unroll
            for (int m = 0; m < 4; m++) fbm[m] = *reinterpret_cast<const bf16x8*>(dyt + (16 * m + lc) * DP + p0);
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const int nt = w + 4 * j;
                if (nt >= 18) continue;  // wave-uniform
                // A operand: [m = k' = 16 nt + lc][k = pixel]: Xcol[pix][k'] = patch[r][2*pix + kw], r = k'>>3, kw = k'&7
                const int kq = 16 * nt + lc, r = kq >> 3, kw = kq & 7;
                bf16x8 fa;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int pix = p0 + e;
                    fa[e] = (short)(pix < OW ? patch[r * LP + 2 * pix + kw] : (bf16_t)0);
                }
#pragma unroll
                for (int m = 0; m < 4; m++) acc[j][m] = mfma16(fa, fbm[m], acc[j][m]);
            }
        }
    }
    // partial[block][k'][co]
    float* out = partial + (long)blockIdx.x * KP * CO;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int nt = w + 4 * j;
        if (nt >= 18) continue;
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) out[(16 * nt + 4 * quad + r) * CO + 16 * m + lc] = acc[j][m][r];
    }
}

// dw[co][r][kw] = sum_g partial[g][(r*8+kw)][co]
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ partial, int G,
                                                                float* __restrict__ dw) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // over KP*CO
    if (i >= KP * CO) return;
    const int kq = i / CO, co = i - kq * CO, r = kq >> 3, kw = kq & 7;
    if (r >= ROWS || kw >= KW) return;
    float s = 0.f;
    for (int g = 0; g < G; g++) s += partial[(long)g * KP * CO + i];
    dw[(co * ROWS + r) * KW + kw] = s;
}

}  // namespace

extern "C" int64_t avsr_stem357_workspace_bytes(void) { return (int64_t)512 * KP * CO * 4 + (int64_t)CO * KP * 2; }

// y[B*T, OH, OW, 64] (bf16) = conv3d(x[B,T,H,W] f32, w[64,1,5,7,7] f32), stride (1,2,2), padding (2,3,3).
// workspace: avsr_stem357_workspace_bytes() bytes (holds the re-laid-out bf16 weights)
extern "C" int avsr_stem357_fwd(const float* x, const float* w, void* y, void* workspace, int B, int T, int H, int W,
                                hipStream_t stream) {
    AVSR_REQUIRE(W % 4 == 0 && W <= 96 && H >= 1, "stem357: W must be a multiple of 4 and <= 96");
    if (B <= 0 || T <= 0) return 0;
    const int OH = (H + 6 - KH) / 2 + 1, OW = (W + 6 - KW) / 2 + 1;
    bf16_t* wp = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(workspace) + (size_t)512 * KP * CO * 4);
    AVSR_LAUNCH(stem_weight_kernel, dim3((CO * KP + 255) / 256), dim3(256), 0, stream, w, wp);
    AVSR_LAUNCH(stem_fwd_kernel, dim3((unsigned)((long)B * T * OH)), dim3(256), 0, stream, x, (const bf16_t*)wp, (bf16_t*)y, T, H, W, OH, OW);
    AVSR_CHECK_LAUNCH("stem357_fwd");
    return 0;
}

// dw[64,1,5,7,7] (f32, overwritten) = weight gradient for dy[B*T, OH, OW, 64] (bf16)
extern "C" int avsr_stem357_wgrad(const void* dy, const float* x, float* dw, void* workspace, int B, int T, int H, int W,
                                  hipStream_t stream) {
    AVSR_REQUIRE(W % 4 == 0 && W <= 96, "stem357: W must be a multiple of 4 and <= 96");
    if (B <= 0 || T <= 0) return 0;
    const int OH = (H + 6 - KH) / 2 + 1, OW = (W + 6 - KW) / 2 + 1;
    AVSR_REQUIRE(OW <= 64, "stem357: at most 64 output columns");
    const long rows = (long)B * T * OH;
    const int G = (int)(rows < 512 ? rows : 512);
    float* partial = reinterpret_cast<float*>(workspace);
    AVSR_LAUNCH(stem_wgrad_kernel, dim3(G), dim3(256), 0, stream, (const bf16_t*)dy, x, partial, T, H, W, OH, OW, rows);
    AVSR_LAUNCH(stem_wgrad_reduce_kernel, dim3((KP * CO + 255) / 256), dim3(256), 0, stream, (const float*)partial, G, dw);
    AVSR_CHECK_LAUNCH("stem357_wgrad");
    return 0;
}
