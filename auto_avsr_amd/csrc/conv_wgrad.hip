// conv_wgrad.hip -- weight gradient of the 3x3 / pad 1 convolutions of the ResNet trunks (resnet.py:10-35), bf16,
// channels-last:     dwp[co][kh][kw][ci] += sum_{n,oh,ow} dy[n,oh,ow,co] * x[n, oh*s+kh-1, ow*s+kw-1, ci]
//
// Why a dedicated kernel: as an implicit GEMM (gemm_tn_fast.hip) every one of the nine taps is its own B operand, so
// the same dy rows and (shifted) x rows are staged nine times and every staged row costs an im2col address decode.
// Here a block owns a 64 (co) x 64 (ci) x ALL NINE TAPS slice of dwp and walks the pixels in tiles of G images x R
// output rows:
//   * one tile = the dy rows of those pixels (k-major, 128 B per pixel = the block's 64 co) and ONE zero-padded x
//     patch (G x ((R-1)s+3) x ((OW-1)s+3) pixels, 128 B each = the block's 64 ci), staged by LDS-DMA into a 2-stage
//     ring; padding pixels and rows past the image come from a page of zeros;
//   * the nine taps are nine SHIFTED VIEWS of that patch: LDS row of pixel k for tap (kh,kw) = view[k] + kh*XW + kw,
//     view[] built once per block (the tile geometry is the same for every tile), so no address decode in the k loop;
//   * both operands have the contraction index (pixels) as the slow LDS dimension: MFMA fragments are fetched with the
//     CDNA4 transpose read ds_read_b64_tr_b16; a 64-byte XOR swizzle keeps the 4-row x 64-byte footprint of a
//     transpose read on distinct banks (rows r and r+2 are 256 bytes apart);
//   * 4 waves = 2 (co halves) x 2 (ci halves), each 9 accumulators of v_mfma_f32_32x32x16_bf16 (144 AGPRs);
//   * split over the pixel tiles across blockIdx.z; every block stores its partial gradient (plain 128-byte-line
//     stores) and a second kernel sums the partials in a fixed order -- deterministic, and cheaper than device-scope
//     float atomics from 512 blocks onto one small tensor (atomics remain available without a workspace).
// Staged bytes per MFMA drop ~9x and the per-load integer work disappears from the inner loop.
#include <math.h>
#include "prims.h"
#include "avsr_hip.h"

namespace {

struct WgParams {
    const bf16_t* dy;
    const bf16_t* x;
    float* dw;         // atomics: dwp ; partial mode: workspace [split][Cout][9][Cin]
    int partial;       // 0: atomicAdd into dw, 1: plain stores of this block's slice into dw + z * |dwp|, 2: none
    int xcd_order;     // 1: XCD-aware work order
    int abl;           // benchmarks only (wrong results): 1 no MFMA, 2 no B reads, 4 no staging after tile 0, 8 no view reads
    const bf16_t* zero;
    int N, H, W, OH, OW, Cin, Cout, S;
    int G, R;          // images / output rows per tile
    int KP;            // dy rows per tile (G*R*OW rounded up to 16)
    int XW, XR;        // patch width / rows per image: (OW-1)*S+3, (R-1)*S+3
    int XROWS;         // patch rows per tile (G*XR*XW rounded up to 8)
    int nbands;        // ceil(OH / R)
    int ntiles;        // ceil(N / G) * nbands
    int tiles_per_block;
};

constexpr int STAGES = 2;
// staging descriptor of one LDS row: byte offset from the tile's base pointer; image | row << 8 | swizzle << 24
// (row = kNever for rows that are always zero)
struct RowDesc { int off, meta; };
constexpr int kNever = 0xfff;
// patch-view table entry of pixel k: byte address (relative to the patch) of its row for kw = 0, 1, 2 with the row's
// swizzle bit already placed in bit 6, so that the address of byte column c is simply v[kw] ^ c
struct ViewEnt { int v[4]; };

// LDS image.  Every row is 128 bytes (64 channels of one pixel) in eight 16-byte chunks; chunk c of a row is stored at
// chunk c ^ 4*s where s is one bit of the row's identity, chosen so that the four rows a transpose read touches
// (4 consecutive pixels) alternate s in pairs: rows 256 bytes apart would otherwise hit the same banks.
//   dy rows: s = bit 1 of the row index k;   x patch rows: s = bit 1 of the patch COLUMN (invariant under the kh
//   shift of a tap, so a pixel needs only three pre-swizzled addresses, one per kw).
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(WgParams p) {
    AVSR_DYN_SMEM(smem);
    const int lane = threadIdx.x & 63, wave = wave_id(), tid = threadIdx.x;
    const int wm = wave >> 1, wn = wave & 1;
    // Work item w = (pixel-tile range z, ci block, co block).  All (ci, co) blocks of one z stage the same dy / x rows
    // (a different 128-byte slice each), so they should meet in one L2: block b runs on XCD b % 8 (observed), hence
    // XCD x is handed the contiguous run of work items [x * total/8, (x+1) * total/8) with z slowest.
    int w = blockIdx.x;
    if (p.xcd_order) {
        const int total = gridDim.x, xcd = w & 7, slot = w >> 3, q = total >> 3, r = total & 7;
        w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int pairs = (p.Cin >> 6) * (p.Cout >> 6);
    const int zid = w / pairs, pr = w - zid * pairs;
    const int ci0 = (pr % (p.Cin >> 6)) * 64, co0 = (pr / (p.Cin >> 6)) * 64;
    const int KP = p.KP, XROWS = p.XROWS, nbands = p.nbands, G = p.G, R = p.R, S = p.S, N = p.N, H = p.H, OH = p.OH;
    const int stage_bytes = (KP + XROWS) * 128;
    ViewEnt* view = reinterpret_cast<ViewEnt*>(smem + STAGES * stage_bytes);  // [KP]
    RowDesc* desc = reinterpret_cast<RowDesc*>(view + KP);                    // [KP + XROWS]
    const int kvalid = G * R * p.OW;

    // ---- once per block: the tile geometry (identical for every tile)
    for (int k = tid; k < KP; k += 256) {
        int g = 0, y = 0, xx = 0;
        const bool ok = k < kvalid;
        if (ok) {
            g = k / (R * p.OW);
            const int rem = k - g * R * p.OW;
            y = rem / p.OW;
            xx = rem - y * p.OW;
        }
        const int row = ok ? (g * p.XR + y * S) * p.XW + xx * S : 0, col = ok ? xx * S : 0;
        ViewEnt e;
#pragma unroll
        for (int kw = 0; kw < 3; kw++) e.v[kw] = (row + kw) * 128 | (((col + kw) & 2) << 5);
        e.v[3] = 0;
        view[k] = e;
        // dy row k: byte offset relative to pixel (n0, r0, 0), image g, row y inside the band
        desc[k] = ok ? RowDesc{((g * OH + y) * p.OW + xx) * p.Cout * 2, g | (y << 8) | ((k & 2) << 23)}
                     : RowDesc{0, kNever << 8};
    }
    for (int j = tid; j < XROWS; j += 256) {
        const int g = j / (p.XR * p.XW);
        const int rem = j - g * p.XR * p.XW;
        const int yy = rem / p.XW, xx = rem - yy * p.XW;
        const bool ok = g < G && xx >= 1 && xx <= p.W;  // column padding is static, row padding depends on the band
        desc[KP + j] = ok ? RowDesc{((g * H + yy) * p.W + xx) * p.Cin * 2, g | (yy << 8) | ((xx & 2) << 23)}
                          : RowDesc{0, kNever << 8};
    }
    __syncthreads();

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    const int t_begin = zid * p.tiles_per_block;
    const int t_end = min(p.ntiles, t_begin + p.tiles_per_block);

    // stage tile t: every wave instruction moves 8 LDS rows (64 lanes x 16 B); rows are padded to whole instructions.
    // Fixed trip counts (KP <= 256, XROWS <= 288) with wave-uniform guards: the descriptors of all of a lane's rows are
    // fetched first, then turned into addresses -- one LDS round trip per tile instead of one per row.
    constexpr int DY_PASSES = 8, X_PASSES = 9;
    const char* const zero = reinterpret_cast<const char*>(p.zero);
    const int rsub = lane >> 3;
    const int chunk0 = (lane & 7) * 16, chunk1 = ((lane & 7) ^ 4) * 16;  // source chunk of this lane for s = 0 / 1
    const char* const dy_blk = reinterpret_cast<const char*>(p.dy + co0);
    const char* const x_blk = reinterpret_cast<const char*>(p.x + ci0);
    auto issue = [&](int t, char* stage) {
        const int grp = t / nbands;
        const int n0 = grp * G, r0 = (t - grp * nbands) * R;
        const int gmax = N - n0;      // images g >= gmax do not exist
        const int ymax = OH - r0;     // band rows y >= ymax do not exist
        const int ytop = r0 * S - 1;  // image row of patch row 0
        const char* const dy_tile = dy_blk + (((long)n0 * OH + r0) * p.OW) * p.Cout * 2;
        const char* const x_tile = x_blk + (((long)n0 * H + ytop) * p.W - 1) * p.Cin * 2;
        RowDesc dd[DY_PASSES], dx[X_PASSES];
#pragma unroll
        for (int j = 0; j < DY_PASSES; j++)
            if (wave * 8 + j * 32 < KP) dd[j] = desc[wave * 8 + j * 32 + rsub];
#pragma unroll
        for (int j = 0; j < X_PASSES; j++)
            if (wave * 8 + j * 32 < XROWS) dx[j] = desc[KP + wave * 8 + j * 32 + rsub];
#pragma unroll
        for (int j = 0; j < DY_PASSES; j++) {
            const int row0 = wave * 8 + j * 32;
            if (row0 < KP) {
                const int m = dd[j].meta;
                const bool ok = (m & 255) < gmax && ((m >> 8) & 0xfff) < ymax;
                glds16(ok ? dy_tile + (dd[j].off + ((m >> 24) ? chunk1 : chunk0)) : zero, stage + row0 * 128);
            }
        }
        char* xs = stage + KP * 128;
#pragma unroll
        for (int j = 0; j < X_PASSES; j++) {
            const int row0 = wave * 8 + j * 32;
            if (row0 < XROWS) {
                const int m = dx[j].meta;
                const bool ok = (m & 255) < gmax && (unsigned)(ytop + ((m >> 8) & 0xfff)) < (unsigned)H;
                glds16(ok ? x_tile + (dx[j].off + ((m >> 24) ? chunk1 : chunk0)) : zero, xs + row0 * 128);
            }
        }
    };

    if (t_begin < t_end) issue(t_begin, smem);
    // per-lane constants of the transpose reads (prims.h lds_tr16): lane (g4, i) addresses row 8*(g4>>1) + (i>>2) (+4)
    // of a 16-row k-step, 4 consecutive columns starting at 16*(g4&1) + 4*(i&3) of the wave's 32-column slice
    const int g4 = lane >> 4, li = lane & 15;
    const int krow = 8 * (g4 >> 1) + (li >> 2);  // bit 1 of krow == bit 1 of krow + 4 == bit 1 of the dy row
    const int acol = ((wm * 32 + 16 * (g4 & 1) + 4 * (li & 3)) * 2) ^ ((krow & 2) << 5);  // swizzled byte column, dy tile
    const int bcol = (wn * 32 + 16 * (g4 & 1) + 4 * (li & 3)) * 2;                         // byte column, x patch
    const int kh_bytes = p.XW * 128;                                                       // one patch row of pixels

    for (int t = t_begin; t < t_end; t++) {
        wait_vmcnt<0>();
        __syncthreads();  // tile t has landed for every wave; everyone is done with the other buffer
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end && !(p.abl & 4)) issue(t + 1, smem + (buf ^ 1) * stage_bytes);
        const char* dys = smem + buf * stage_bytes + krow * 128 + acol;
        const char* xs = smem + buf * stage_bytes + KP * 128;
        for (int ks = 0; ks < KP / 16; ks++) {
            const ViewEnt vlo = view[(p.abl & 8) ? krow : ks * 16 + krow], vhi = view[(p.abl & 8) ? krow + 4 : ks * 16 + krow + 4];
            // Transpose reads in the asm form (they must not wait for the next tile's LDS-DMA, see prims.h), software
            // pipelined six taps ahead of the MFMAs: the LGKM counter is 4 bits, so at most 15 reads may be in flight.
            // Each MFMA waits only for its own fragments: lgkmcnt(n) = reads issued after them that may still be pending.
            bf16x4 alo = lds_tr16_async(reinterpret_cast<const bf16_t*>(dys + ks * 2048));
            bf16x4 ahi = lds_tr16_async(reinterpret_cast<const bf16_t*>(dys + ks * 2048 + 512));
            const char* plo[3];
            const char* phi[3];
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                plo[kw] = xs + (vlo.v[kw] ^ bcol);
                phi[kw] = xs + (vhi.v[kw] ^ bcol);
            }
            bf16x4 blo[9], bhi[9];
            auto read_tap = [&](int tap) {
                if (p.abl & 2) return;
                blo[tap] = lds_tr16_async(reinterpret_cast<const bf16_t*>(plo[tap % 3] + (tap / 3) * kh_bytes));
                bhi[tap] = lds_tr16_async(reinterpret_cast<const bf16_t*>(phi[tap % 3] + (tap / 3) * kh_bytes));
            };
            constexpr int AHEAD = 6;
#pragma unroll
            for (int tap = 0; tap < AHEAD; tap++) read_tap(tap);
            bf16x8 a;
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                // issued after this tap's pair: taps tap+1 .. min(tap+AHEAD-1, 8)
                switch (2 * ((tap + AHEAD - 1 < 8 ? tap + AHEAD - 1 : 8) - tap)) {
                    case 10: lds_wait<10>(); break;
                    case 8: lds_wait<8>(); break;
                    case 6: lds_wait<6>(); break;
                    case 4: lds_wait<4>(); break;
                    case 2: lds_wait<2>(); break;
                    default: lds_wait<0>(); break;
                }
                if (tap == 0) {
                    lds_tie(alo);
                    lds_tie(ahi);
                    a = bf16x8{alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
                }
                lds_tie(blo[tap]);
                lds_tie(bhi[tap]);
                const bf16x8 b{blo[tap][0], blo[tap][1], blo[tap][2], blo[tap][3], bhi[tap][0], bhi[tap][1], bhi[tap][2], bhi[tap][3]};
                if (!(p.abl & 1)) acc[tap] = mfma32(a, b, acc[tap]);
                if (tap + AHEAD < 9) read_tap(tap + AHEAD);
                sched_fence();  // keep each MFMA right behind its own wait
            }
        }
    }

    // ---- dwp[co][tap][ci] (+)= acc.  A wave instruction covers 2 rows x 32 consecutive ci = whole 128-byte lines.
    if (p.partial == 2) return;
    float* out = p.dw + (p.partial ? (size_t)zid * p.Cout * 9 * p.Cin : 0);
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int ci = ci0 + wn * 32 + (lane & 31);
            float* dst = out + ((size_t)co * 9 + tap) * p.Cin + ci;
            if (p.partial) *dst = acc[tap][r];
            else atomicAdd(dst, acc[tap][r]);
        }
}

// dwp[i] = sum_z ws[z][i]: the blocks' partial gradients are combined in a fixed order (deterministic, no atomics).
// 1024 threads = 64 float4 columns x 16 z-lanes; a z-lane sums every 16th partial with independent loads in flight,
// then the 16 lane sums are combined through LDS in a fixed order.
// torch_layout: dw is [Cout][Cin][3][3] (what autograd hands to the optimizer) instead of [Cout][3][3][Cin]
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n4, int split,
                                                            int Cin, int torch_layout) {
    __shared__ f32x4 part[16][64];
    const int tx = threadIdx.x & 63, tz = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + tx;
    f32x4 s{0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        const f32x4* src = reinterpret_cast<const f32x4*>(ws) + i;
        int z = tz;
        for (; z + 48 < split; z += 64) {
            const f32x4 v0 = src[(long)z * n4], v1 = src[(long)(z + 16) * n4], v2 = src[(long)(z + 32) * n4],
                        v3 = src[(long)(z + 48) * n4];
#pragma unroll
            for (int e = 0; e < 4; e++) s[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
        }
        for (; z < split; z += 16) {
            const f32x4 v = src[(long)z * n4];
#pragma unroll
            for (int e = 0; e < 4; e++) s[e] += v[e];
        }
    }
    part[tz][tx] = s;
    __syncthreads();
    if (tz == 0 && i < n4) {
#pragma unroll
        for (int q = 1; q < 16; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) s[e] += part[q][tx][e];
        if (!torch_layout) {
            reinterpret_cast<f32x4*>(dw)[i] = s;
        } else {
            const long e0 = i * 4;  // = (co * 9 + tap) * Cin + ci
            const int ci = (int)(e0 % Cin);
            const long ct = e0 / Cin;
            const int tap = (int)(ct % 9);
            const long co = ct / 9;
#pragma unroll
            for (int e = 0; e < 4; e++) dw[(co * Cin + ci + e) * 9 + tap] = s[e];
        }
    }
}

struct Plan { WgParams p; int split; size_t lds; };

// tile geometry: G whole images (small images) or one band of R output rows; at most 256 pixels and 38 KiB per stage
// (two stages, two blocks per CU); maximise useful pixels per unit of max(MFMA time, staging time)
bool make_plan(int N, int H, int W, int Cin, int Cout, int stride, Plan& pl) {
    const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
    WgParams& p = pl.p;
    p = WgParams{};
    p.N = N; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.Cin = Cin; p.Cout = Cout; p.S = stride;
    p.XW = (OW - 1) * stride + 3;
    double best = -1.0;
    for (int mode = 0; mode < 2; mode++)
        for (int v = 1; v <= (mode == 0 ? OH : 64); v++) {
            const int G = mode == 0 ? 1 : v, R = mode == 0 ? v : OH;
            if (G > N || G > 255 || R > 1000) break;
            const int XR = (R - 1) * stride + 3;
            const int KP = (G * R * OW + 15) / 16 * 16, XROWS = (G * XR * p.XW + 7) / 8 * 8;
            if (KP > 256 || XROWS > 288 || (KP + XROWS) * 128 > 37 * 1024) break;
            const int nb = (OH + R - 1) / R;
            const double useful = (double)G * OH * OW / nb;                                  // real pixels per tile
            const double cost = fmax(KP / 16 * 9 * 32.0, (KP + XROWS) * 128 / 20.0) + 300.0;  // clocks per tile
            if (useful / cost > best) {
                best = useful / cost;
                p.G = G; p.R = R; p.XR = XR; p.KP = KP; p.XROWS = XROWS; p.nbands = nb;
            }
        }
    if (best <= 0.0) return false;
    p.ntiles = (N + p.G - 1) / p.G * p.nbands;
    const int pairs = (Cin / 64) * (Cout / 64);
    const int target = avsr_tune_knobs[15] > 0 ? avsr_tune_knobs[15] : 512;  // knob 15: block-count target (A/B runs)
    int split = (target + pairs - 1) / pairs;  // default: two blocks per CU
    if (split > p.ntiles) split = p.ntiles;
    p.tiles_per_block = (p.ntiles + split - 1) / split;
    pl.split = (p.ntiles + p.tiles_per_block - 1) / p.tiles_per_block;
    pl.lds = (size_t)STAGES * (p.KP + p.XROWS) * 128 + (size_t)p.KP * 16 + (size_t)(p.KP + p.XROWS) * 8;
    return true;
}

}  // namespace

// Bytes of workspace with which avsr_conv3x3_wgrad_bf16 runs in its deterministic (partial sums + ordered reduce) mode.
extern "C" int64_t avsr_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int stride) {
    Plan pl;
    if (N <= 0 || Cin % 64 || Cout % 64 || (stride != 1 && stride != 2) || !make_plan(N, H, W, Cin, Cout, stride, pl)) return 0;
    return (int64_t)pl.split * Cout * 9 * Cin * 4;
}

// dwp[Cout][3][3][Cin] (f32) = dy[N,OH,OW,Cout]^T (x) shifted x[N,H,W,Cin]; 3x3, pad 1, stride 1 or 2; Cin % 64 == 0,
// Cout % 64 == 0; zero_page: >= 16 zero bytes of device memory.  With a workspace of at least
// avsr_conv3x3_wgrad_workspace_bytes() dwp is OVERWRITTEN with the ordered sum of the blocks' partial gradients;
// without one (workspace = NULL) the blocks atomicAdd into dwp, which the caller must have zeroed.
extern "C" int avsr_conv3x3_wgrad_bf16(const void* dy, const void* x, float* dwp, const void* zero_page, void* workspace,
                                       int64_t workspace_bytes, int N, int H, int W, int Cin, int Cout, int stride,
                                       int torch_layout, hipStream_t stream) {
    AVSR_REQUIRE(!torch_layout || workspace != nullptr, "conv3x3_wgrad_bf16: the torch layout needs the workspace mode");
    AVSR_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, "conv3x3_wgrad_bf16: channel counts must be multiples of 64");
    AVSR_REQUIRE(stride == 1 || stride == 2, "conv3x3_wgrad_bf16: stride must be 1 or 2");
    AVSR_REQUIRE(zero_page != nullptr, "conv3x3_wgrad_bf16: zero page required");
    AVSR_REQUIRE((long)N * H * W < (1l << 31), "conv3x3_wgrad_bf16: pixel count exceeds int32");
    if (N <= 0) return 0;
    Plan pl;
    AVSR_REQUIRE(make_plan(N, H, W, Cin, Cout, stride, pl), "conv3x3_wgrad_bf16: image row too wide for one LDS tile");
    WgParams& p = pl.p;
    p.dy = reinterpret_cast<const bf16_t*>(dy); p.x = reinterpret_cast<const bf16_t*>(x);
    p.zero = reinterpret_cast<const bf16_t*>(zero_page);
    const int64_t need = (int64_t)pl.split * Cout * 9 * Cin * 4;
    const bool partial = workspace != nullptr && (avsr_tune_knobs[3] != 1 || torch_layout);
    AVSR_REQUIRE(!partial || workspace_bytes >= need, "conv3x3_wgrad_bf16: workspace too small");
    p.dw = partial ? reinterpret_cast<float*>(workspace) : dwp;
    p.partial = avsr_tune_knobs[3] == 2 ? 2 : (partial ? 1 : 0);
    p.xcd_order = avsr_tune_knobs[4] != 1;
    p.abl = avsr_tune_knobs[5];
    dim3 grid((Cin / 64) * (Cout / 64) * pl.split), block(256);
    AVSR_LAUNCH(conv3x3_wgrad_kernel, grid, block, pl.lds, stream, p);
    if (partial) {
        const long n4 = (long)Cout * 9 * Cin / 4;
        AVSR_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(1024), 0, stream,
                    reinterpret_cast<const float*>(workspace), dwp, n4, pl.split, Cin, torch_layout);
    }
    AVSR_CHECK_LAUNCH("conv3x3_wgrad_bf16");
    return 0;
}
