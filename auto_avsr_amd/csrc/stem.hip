// stem.hip -- dedicated kernels for the visual front-end stem  Conv3d(1, 64, (5,7,7), stride (1,2,2), padding (2,3,3))
// (frontend/resnet.py:204-211) in bf16 mode.  C_in = 1 makes the generic im2col gather element-wise; here the 35
// (kt,kh) input rows an output row needs are staged ONCE in LDS (as bf16, zero padded), and the MFMA operands are
// read from that patch: K is re-indexed as (kt*7+kh)*8 + 1 + kw with a zero FIRST tap, so a lane's 8 consecutive k values
// are 8 consecutive input columns starting at an even (4-byte aligned) patch column -- four aligned ds_read_b32 -- while
// the patch itself is staged with its data at column 4 (8-byte aligned): one ds_write_b64 per float4 of input.
//   forward : a block walks 8 output rows (n, oh) with its weights (64 x 288 bf16) in registers; wave w owns 16 output
//             channels; 3 pixel tiles x 9 k-steps of v_mfma_f32_16x16x32_bf16 per row; the row is assembled in LDS and
//             leaves as 128-byte pixels with 16-byte stores.
//   wgrad   : persistent blocks loop over output rows, accumulate dW^T tiles in registers (contraction over the 44
//             pixels of a row), write per-block partials, a second kernel sums them (no atomics).
#include "prims.h"
#include "avsr_hip.h"

namespace {

constexpr int KT = 5, KH = 7, KW = 7, ROWS = KT * KH;  // 35 (kt,kh) rows
constexpr int KP = 288;                                // padded K: 36 rows x 8 taps
constexpr int CO = 64;
constexpr int LP = 104;                                // LDS row pitch (bf16): >= 2*47+8 and >= 4 + W (W <= 96), multiple of 8

// wp[co][(kt*7+kh)*8 + 1 + kw] = w[co][0][kt][kh][kw]; zero for slot 0 and rows >= 35
// NS = 2 (precise mode): wp[plane][...], plane 0 = hi, plane 1 = lo of the split value (prims.h split_bf16)
template <int NS>
__global__ void stem_weight_kernel(const float* __restrict__ w, bf16_t* __restrict__ wp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= CO * KP) return;
    const int co = i / KP, k = i % KP, r = k >> 3, j = k & 7;
    bf16_t pl[NS];
    split_bf16<NS>((r < ROWS && j >= 1) ? w[(co * ROWS + r) * KW + j - 1] : 0.f, pl);
#pragma unroll
    for (int q = 0; q < NS; q++) wp[q * CO * KP + i] = pl[q];
}

// The 36 x LP bf16 patch of output row (n, oh): patch[r][4 + iw] = x[b][t+kt-2][2*oh+kh-3][iw]; output pixel ow reads the
// columns 2*ow .. 2*ow+7 = input columns 2*ow-4 .. 2*ow+3 (the first one under the zero tap).  Padding columns and
// the dummy row 35 are zeroed once per block (patch_init); a row's data is fetched into registers (patch_load) one
// row AHEAD of its use and written to LDS (patch_store) after the previous row's compute -- the global-load latency
// of row i+1 hides behind the MFMAs of row i.
constexpr int PATCH_V = 4;  // float4 per thread: 35 rows x (W/4 <= 24) <= 840 <= 4 * 256
template <int NS = 1>
AVSR_DEV void patch_init(bf16_t* patch, int W) {
    for (int i = threadIdx.x; i < 36 * LP; i += 256) {
        const int r = i / LP, c = i - r * LP;
        if (r == 35 || c < 4 || c >= 4 + W)
#pragma unroll
            for (int q = 0; q < NS; q++) patch[q * 36 * LP + i] = 0;
    }
}
AVSR_DEV void patch_load(f32x4 (&q)[PATCH_V], const float* __restrict__ x, long row, int OH, int T, int H, int W) {
    const int n = (int)(row / OH), oh = (int)(row - (long)n * OH);
    const int b = n / T, t = n - b * T;
    const int nv = W >> 2;  // float4 per row (W % 4 == 0)
#pragma unroll
    for (int j = 0; j < PATCH_V; j++) {
        const int i = threadIdx.x + 256 * j;
        q[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < ROWS * nv) {
            const int r = i / nv, v = i - r * nv;
            const int kt = r / KH, kh = r - kt * KH;
            const int tt = t + kt - 2, ih = 2 * oh + kh - 3;
            if (tt >= 0 && tt < T && ih >= 0 && ih < H)
                q[j] = *reinterpret_cast<const f32x4*>(x + (((long)b * T + tt) * H + ih) * W + v * 4);
        }
    }
}
template <int NS = 1>
AVSR_DEV void patch_store(bf16_t* patch, const f32x4 (&q)[PATCH_V], int W) {
    const int nv = W >> 2;
#pragma unroll
    for (int j = 0; j < PATCH_V; j++) {
        const int i = threadIdx.x + 256 * j;
        if (i < ROWS * nv) {
            const int r = i / nv, v = i - r * nv;
            bf16x4 o[NS];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                bf16_t pl[NS];
                split_bf16<NS>(q[j][e], pl);
#pragma unroll
                for (int s = 0; s < NS; s++) o[s][e] = (short)pl[s];
            }
#pragma unroll
            for (int s = 0; s < NS; s++)  // 8-byte aligned (LP and 4 + 4 v are multiples of 4)
                *reinterpret_cast<bf16x4*>(patch + s * 36 * LP + r * LP + 4 + v * 4) = o[s];
        }
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

constexpr int FWD_ROWS = 8;  // output rows per block: the 36 KB of weights a block holds in registers are fetched once
constexpr int OP = 72;       // pitch (bf16) of the LDS output row [pixel][64 channels]

// NS = 1: bf16 operands (bench mode).  NS = 2: split hi + lo planes of the f32 input and weights, three MFMAs per product
// (precise / hpf modes).  TO: output element type (bf16 / f32).
template <int NS, class TO>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const bf16_t* __restrict__ wp,
                                                       TO* __restrict__ y, int T, int H, int W, int OH, int OW,
                                                       long total_rows, bf16_t* __restrict__ y2 = nullptr,
                                                       float* __restrict__ stats_part = nullptr) {
    constexpr int OPT = sizeof(TO) == 2 ? OP : 68;  // pitch of the LDS output row in elements (16-byte multiple, bank-skewed)
    __shared__ __attribute__((aligned(16))) bf16_t patch[NS * 36 * LP];
    __shared__ __attribute__((aligned(16))) TO orow[64 * OPT];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int quad = lane >> 4, lc = lane & 15;
    // this wave's weights: channel 16w + lc, k-step ks: taps [ks*32 + 8*quad, +8)
    Frag<NS> fb[9];
#pragma unroll
    for (int ks = 0; ks < 9; ks++)
#pragma unroll
        for (int q = 0; q < NS; q++)
            fb[ks].p[q] = *reinterpret_cast<const bf16x8*>(wp + q * CO * KP + (16 * w + lc) * KP + ks * 32 + 8 * quad);
    const int ntile = (OW + 15) / 16;
    const long row_end = min(total_rows, ((long)blockIdx.x + 1) * FWD_ROWS);
    patch_init<NS>(patch, W);
    f32x4 q[PATCH_V];
    patch_load(q, x, (long)blockIdx.x * FWD_ROWS, OH, T, H, W);
    // BatchNorm statistics of the block's output (stats_part, f32 output only): in the copy-out below a thread always handles the
    // same four channels (256 threads, 16 chunks per pixel), so it sums them and their squares over all pixels of all rows
    f32x4 st1 = f32x4{0.f, 0.f, 0.f, 0.f}, st2 = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long row = (long)blockIdx.x * FWD_ROWS; row < row_end; row++) {
        __syncthreads();  // the previous row's patch and output row are no longer read
        patch_store<NS>(patch, q, W);
        if (row + 1 < row_end) patch_load(q, x, row + 1, OH, T, H, W);  // in flight during this row's MFMAs
        __syncthreads();
        for (int mt = 0; mt < ntile; mt++) {
            const int ow = min(mt * 16 + lc, OW - 1);
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            // D^T = W . X^T: the weights are the A operand (rows = this wave's 16 channels), the patch the B operand (columns =
            // 16 pixels) -- the per-lane fragments are the same registers either way, but the accumulator then holds FOUR
            // CONSECUTIVE CHANNELS of one pixel (row 4*quad + r = channel, column lc = pixel): one 8-byte LDS store per tile
            // instead of four 2-byte ones (the 2-byte stores were a quarter of the kernel's LDS instruction issue)
#pragma unroll
            for (int ks = 0; ks < 9; ks++) {
                Frag<NS> fx;
#pragma unroll
                for (int s = 0; s < NS; s++) fx.p[s] = patch_frag(patch + s * 36 * LP, ks * 4 + quad, ow);
                acc = mma16<NS>(fb[ks], fx, acc);
            }
            TO* o = orow + (mt * 16 + lc) * OPT + 16 * w + 4 * quad;
            if (sizeof(TO) == 2) {
                *reinterpret_cast<bf16x4*>(o) = bf16x4{(short)f2bf(acc[0]), (short)f2bf(acc[1]), (short)f2bf(acc[2]), (short)f2bf(acc[3])};
            } else {
                *reinterpret_cast<f32x4*>(o) = acc;
            }
        }
        __syncthreads();
        // the four waves' 16-channel slices are now one [OW][64] row: whole pixels, 16-byte stores
        TO* dst = y + (row * OW) * CO;
        constexpr int EPC = 16 / sizeof(TO), CPP = CO / EPC;  // elements per 16-byte chunk, chunks per pixel
        for (int i = threadIdx.x; i < OW * CPP; i += 256) {
            const int pix = i / CPP, c = (i % CPP) * EPC;
            const f32x4 v = *reinterpret_cast<const f32x4*>(orow + pix * OPT + c);
            *reinterpret_cast<f32x4*>(dst + pix * CO + c) = v;
            if (sizeof(TO) == 4 && stats_part) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    st1[e] += v[e];
                    st2[e] += v[e] * v[e];
                }
            }
            if (sizeof(TO) == 4 && y2)  // bf16 twin of the f32 result (hpf mode)
                *reinterpret_cast<bf16x4*>(y2 + (row * OW + pix) * CO + c) =
                    bf16x4{(short)f2bf(v[0]), (short)f2bf(v[1]), (short)f2bf(v[2]), (short)f2bf(v[3])};
        }
    }
    if (sizeof(TO) == 4 && stats_part) {  // 16 threads per 4-channel chunk meet in LDS: row blockIdx.x of [blocks][2][64]
        __syncthreads();
        float* red = reinterpret_cast<float*>(orow);  // [16 thread groups][2][64]
        const int grp = threadIdx.x >> 4, c = (threadIdx.x & 15) * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            red[(grp * 2 + 0) * CO + c + e] = st1[e];
            red[(grp * 2 + 1) * CO + c + e] = st2[e];
        }
        __syncthreads();
        if (threadIdx.x < 2 * CO) {
            float v = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < 16; g2++) v += red[g2 * 2 * CO + threadIdx.x];
            stats_part[(size_t)blockIdx.x * 2 * CO + threadIdx.x] = v;
        }
    }
}

// ---- weight gradient: dWt[k'][co] = sum_pix Xcol[pix][k'] dY[pix][co]; per wave: M-tiles (k' groups of 16) w, w+4, ..
// The contraction runs over the pixels of an output row, so both MFMA operands need 8 CONSECUTIVE PIXELS per lane:
//   * dY: the row is copied as it lies ([pixel][64 co]) and read with the LDS transpose read (two ds_read_b64_tr_b16);
//   * Xcol[pix][(r,kw)] = xpad[r][2 pix + kw] walks the input row with stride 2.  The patch is therefore staged
//     DEINTERLEAVED and PRE-SHIFTED: EO[parity][shift][r][i] = xpad[r][2 (i + shift) + parity], so that the 8 pixels
//     p0..p0+7 of tap kw are the 16 aligned bytes EO[kw & 1][kw >> 1][r][p0..p0+7] -- one ds_read_b128 instead of eight
//     2-byte reads and their packing.  (kw = 7 and row 35 are padding of the 36 x 8 tap grid: whatever they
//     accumulate is dropped by the reduce kernel.)
constexpr int EP = 56;                  // pitch of an EO row (bf16): 48 pixels + pad, 16-byte multiple
constexpr int EO_ARR = 36 * EP;         // one (parity, shift) array
constexpr int DYP = 64;                 // dY tile: [64 pixels][64 co]

// x columns for thread (r, g) of row (n, oh): xpad columns 16g .. 16g+23 = input columns 16g-3 .. 16g+20, fetched as
// seven aligned float4 starting at 16g-4 (eo_load, one row ahead of use); eo_store converts and writes the eight
// (parity, shift) vectors.
AVSR_DEV void eo_load(float (&v)[28], const float* __restrict__ x, long row, int OH, int T, int H, int W) {
    const int tid = threadIdx.x;
    if (tid >= ROWS * 6) return;
    const int n = (int)(row / OH), oh = (int)(row - (long)n * OH);
    const int b = n / T, t = n - b * T;
    const int r = tid / 6, g = tid - r * 6;
    const int kt = r / KH, kh = r - kt * KH;
    const int tt = t + kt - 2, ih = 2 * oh + kh - 3;
    const bool row_ok = tt >= 0 && tt < T && ih >= 0 && ih < H;
    const float* src = x + (((long)b * T + tt) * H + ih) * W;
#pragma unroll
    for (int q = 0; q < 7; q++) {
        const int c = 16 * g - 4 + 4 * q;
        f32x4 f = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row_ok && c >= 0 && c + 3 < W) f = *reinterpret_cast<const f32x4*>(src + c);
        else if (row_ok)
#pragma unroll
            for (int e = 0; e < 4; e++) f[e] = (c + e >= 0 && c + e < W) ? src[c + e] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) v[4 * q + e] = f[e];
    }
}
AVSR_DEV void eo_store(bf16_t* eo, const float (&v)[28]) {
    const int tid = threadIdx.x;
    if (tid >= ROWS * 6) return;
    const int r = tid / 6, g = tid - r * 6;
    // v[1 + j] = xpad[16g + j]; sequence entry i = 8g + m of parity p is xpad[2 i + p] = v[1 + 2 m + p]
#pragma unroll
    for (int par = 0; par < 2; par++)
#pragma unroll
        for (int sh = 0; sh < 4; sh++) {
            bf16x8 o;
#pragma unroll
            for (int m = 0; m < 8; m++) o[m] = (short)f2bf(v[1 + 2 * (m + sh) + par]);
            *reinterpret_cast<bf16x8*>(eo + (par * 4 + sh) * EO_ARR + r * EP + 8 * g) = o;
        }
}

__global__ __launch_bounds__(256) void stem_wgrad_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                         float* __restrict__ partial, int T, int H, int W, int OH, int OW,
                                                         long total_rows) {
    __shared__ __attribute__((aligned(16))) bf16_t eo[8 * EO_ARR];
    __shared__ __attribute__((aligned(16))) bf16_t dyt[64 * DYP];  // dyt[pixel][co]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int quad = lane >> 4, lc = lane & 15;
    // output tile (nt, mt): rows = k' in [16 nt, 16 nt + 16), cols = co in [16 mt, +16); wave w: nt = w + 4 j
    f32x4 acc[5][4];
#pragma unroll
    for (int j = 0; j < 5; j++)
#pragma unroll
        for (int m = 0; m < 4; m++) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    // pixels >= OW of the dY tile stay zero for the whole kernel; row 35 of the EO arrays is never staged: zero it once
    for (int i = threadIdx.x; i < 64 * DYP / 8; i += 256) reinterpret_cast<bf16x8*>(dyt)[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < 8 * EP / 8; i += 256) {
        const int arr = i / (EP / 8), c8 = (i - arr * (EP / 8)) * 8;
        *reinterpret_cast<bf16x8*>(eo + arr * EO_ARR + 35 * EP + c8) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    // entries 48..55 of every row are read for the (zero) pixels 48..55 of the dY tile but never staged: 0 * NaN = NaN
    for (int i = threadIdx.x; i < 8 * 36; i += 256)
        *reinterpret_cast<bf16x8*>(eo + (i / 36) * EO_ARR + (i % 36) * EP + 48) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    // per-lane operand addresses: A row k' = 16 nt + lc -> (r, kw); B (transpose read) rows 8*quad + (lc>>2) (+4)
    int a_off[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int kq = 16 * min(w + 4 * j, 17) + lc, r = kq >> 3, kw = kq & 7;
        a_off[j] = ((kw & 1) * 4 + (kw >> 1)) * EO_ARR + r * EP + 8 * quad;
    }
    const int b_row = 8 * quad + (lc >> 2), b_col = 4 * (lc & 3);
    // software pipeline over rows: x columns and the dY row of row i+1 are fetched into registers while row i multiplies
    float xv[28];
    bf16x8 dv[2];
    auto dy_load = [&](long row) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int i = threadIdx.x + 256 * j;
            dv[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (i < OW * 8) dv[j] = *reinterpret_cast<const bf16x8*>(dy + (row * OW + (i >> 3)) * CO + (i & 7) * 8);
        }
    };
    if ((long)blockIdx.x < total_rows) {
        eo_load(xv, x, blockIdx.x, OH, T, H, W);
        dy_load(blockIdx.x);
    }
    for (long row = blockIdx.x; row < total_rows; row += gridDim.x) {
        __syncthreads();
        eo_store(eo, xv);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int i = threadIdx.x + 256 * j;
            if (i < OW * 8) *reinterpret_cast<bf16x8*>(dyt + (i >> 3) * DYP + (i & 7) * 8) = dv[j];
        }
        if (row + gridDim.x < total_rows) {
            eo_load(xv, x, row + gridDim.x, OH, T, H, W);
            dy_load(row + gridDim.x);
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {  // 32 pixels per k-step
            bf16x8 fbm[4];                // B operand: [n = co][k = pixel]
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const bf16_t* p0 = dyt + (ks * 32 + b_row) * DYP + 16 * m + b_col;
                const bf16x4 lo = lds_tr16(p0), hi = lds_tr16(p0 + 4 * DYP);
                fbm[m] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < 5; j++) {
                if (w + 4 * j >= 18) continue;  // wave-uniform
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(eo + a_off[j] + ks * 32);
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

// dw[co][r][kw] = sum_g partial[g][(r*8+kw)][co]: 1024 threads = 64 outputs x 16 partial lanes (independent loads in
// flight), lane sums combined through LDS in a fixed order
__global__ __launch_bounds__(1024) void stem_wgrad_reduce_kernel(const float* __restrict__ partial, int G,
                                                                 float* __restrict__ dw) {
    __shared__ float part[16][64];
    const int tx = threadIdx.x & 63, tz = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;  // over KP*CO
    float s = 0.f;
    if (i < KP * CO)
        for (int g = tz; g < G; g += 16) s += partial[(long)g * KP * CO + i];
    part[tz][tx] = s;
    __syncthreads();
    if (tz != 0 || i >= KP * CO) return;
#pragma unroll
    for (int q = 1; q < 16; q++) s += part[q][tx];
    const int kq = i / CO, co = i - kq * CO, r = kq >> 3, kw = kq & 7;
    if (r < ROWS && kw < KW) dw[(co * ROWS + r) * KW + kw] = s;
}

}  // namespace

constexpr int WG_MAX = 1024;  // upper bound of the weight-gradient grid (per-block partials live in the workspace)
extern "C" int64_t avsr_stem357_workspace_bytes(void) { return (int64_t)WG_MAX * KP * CO * 4 + (int64_t)2 * CO * KP * 2; }

// y[B*T, OH, OW, 64] (bf16) = conv3d(x[B,T,H,W] f32, w[64,1,5,7,7] f32), stride (1,2,2), padding (2,3,3).
// workspace: avsr_stem357_workspace_bytes() bytes (holds the re-laid-out bf16 weights)
extern "C" int avsr_stem357_fwd(const float* x, const float* w, void* y, void* workspace, int B, int T, int H, int W,
                                hipStream_t stream) {
    AVSR_REQUIRE(W % 4 == 0 && W <= 96 && H >= 1, "stem357: W must be a multiple of 4 and <= 96");
    static_assert(ROWS * 24 <= PATCH_V * 256, "patch prefetch registers");
    if (B <= 0 || T <= 0) return 0;
    const int OH = (H + 6 - KH) / 2 + 1, OW = (W + 6 - KW) / 2 + 1;
    bf16_t* wp = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(workspace) + (size_t)WG_MAX * KP * CO * 4);
    AVSR_LAUNCH(stem_weight_kernel<1>, dim3((CO * KP + 255) / 256), dim3(256), 0, stream, w, wp);
    const long rows = (long)B * T * OH;
    AVSR_REQUIRE(OW <= 64, "stem357: at most 64 output columns");
    AVSR_LAUNCH((stem_fwd_kernel<1, bf16_t>), dim3((unsigned)((rows + FWD_ROWS - 1) / FWD_ROWS)), dim3(256), 0, stream, x,
                (const bf16_t*)wp, (bf16_t*)y, T, H, W, OH, OW, rows);
    AVSR_CHECK_LAUNCH("stem357_fwd");
    return 0;
}

// The same convolution for the precise / hpf modes: y (f32) from split hi + lo bf16 planes of x and w (three MFMAs per
// product, ~2^-16 relative error -- the arithmetic of avsr_conv_stem_fwd with precise = 1).  Same workspace.
static int stem357_fwd_f32s_impl(const float* x, const float* w, float* y, void* y2, void* workspace, int B, int T, int H, int W,
                                 float* stats_part, int stats_rows, hipStream_t stream) {
    AVSR_REQUIRE(W % 4 == 0 && W <= 96 && H >= 1, "stem357: W must be a multiple of 4 and <= 96");
    if (B <= 0 || T <= 0) return 0;
    const int OH = (H + 6 - KH) / 2 + 1, OW = (W + 6 - KW) / 2 + 1;
    bf16_t* wp = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(workspace) + (size_t)WG_MAX * KP * CO * 4);
    AVSR_LAUNCH(stem_weight_kernel<2>, dim3((CO * KP + 255) / 256), dim3(256), 0, stream, w, wp);
    const long rows = (long)B * T * OH;
    AVSR_REQUIRE(OW <= 64, "stem357: at most 64 output columns");
    const long blocks = (rows + FWD_ROWS - 1) / FWD_ROWS;
    AVSR_REQUIRE(stats_part == nullptr || stats_rows >= blocks, "stem357: statistics buffer too small");
    AVSR_LAUNCH((stem_fwd_kernel<2, float>), dim3((unsigned)blocks), dim3(256), 0, stream, x, (const bf16_t*)wp, y, T, H, W, OH, OW, rows,
                (bf16_t*)y2, stats_part);
    AVSR_CHECK_LAUNCH("stem357_fwd_f32s");
    return 0;
}

// The same convolution for the precise / hpf modes: y (f32) from split hi + lo bf16 planes of x and w (three MFMAs per
// product, ~2^-16 relative error -- the arithmetic of avsr_conv_stem_fwd with precise = 1).  Same workspace.
extern "C" int avsr_stem357_fwd_f32s(const float* x, const float* w, float* y, void* y2, void* workspace, int B, int T, int H,
                                     int W, hipStream_t stream) {
    return stem357_fwd_f32s_impl(x, w, y, y2, workspace, B, T, H, W, nullptr, 0, stream);
}
// ... leaving the BatchNorm statistics of its output behind (frontend/resnet.py:203-219: Conv3d -> BatchNorm3d in batch-statistics
// mode): row j of stats_part [stats_rows >= avsr_stem357_stat_rows(B, T, H)][2][64] = per-channel sums / sums of squares of the
// output rows block j wrote; finish with avsr_bn_finalize_parts / avsr_bn_stats_parts
extern "C" int64_t avsr_stem357_stat_rows(int B, int T, int H) {
    const int OH = (H + 6 - KH) / 2 + 1;
    return ((int64_t)B * T * OH + FWD_ROWS - 1) / FWD_ROWS;
}
extern "C" int avsr_stem357_fwd_f32s_stats(const float* x, const float* w, float* y, void* y2, void* workspace, int B, int T, int H,
                                           int W, float* stats_part, int stats_rows, hipStream_t stream) {
    AVSR_REQUIRE(stats_part != nullptr, "stem357_fwd_f32s_stats: statistics buffer required");
    return stem357_fwd_f32s_impl(x, w, y, y2, workspace, B, T, H, W, stats_part, stats_rows, stream);
}

// dw[64,1,5,7,7] (f32, overwritten) = weight gradient for dy[B*T, OH, OW, 64] (bf16)
extern "C" int avsr_stem357_wgrad(const void* dy, const float* x, float* dw, void* workspace, int B, int T, int H, int W,
                                  hipStream_t stream) {
    AVSR_REQUIRE(W % 4 == 0 && W <= 96, "stem357: W must be a multiple of 4 and <= 96");
    if (B <= 0 || T <= 0) return 0;
    const int OH = (H + 6 - KH) / 2 + 1, OW = (W + 6 - KW) / 2 + 1;
    AVSR_REQUIRE(OW <= 64, "stem357: at most 64 output columns");
    const long rows = (long)B * T * OH;
    int gmax = avsr_tune_knobs[6] > 0 ? avsr_tune_knobs[6] : 512;  // knob 6: persistent blocks (benchmarks)
    if (gmax > WG_MAX) gmax = WG_MAX;
    const int G = (int)(rows < gmax ? rows : gmax);
    float* partial = reinterpret_cast<float*>(workspace);
    AVSR_LAUNCH(stem_wgrad_kernel, dim3(G), dim3(256), 0, stream, (const bf16_t*)dy, x, partial, T, H, W, OH, OW, rows);
    AVSR_LAUNCH(stem_wgrad_reduce_kernel, dim3((KP * CO + 63) / 64), dim3(1024), 0, stream, (const float*)partial, G, dw);
    AVSR_CHECK_LAUNCH("stem357_wgrad");
    return 0;
}
