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
//   * the nine taps are nine SHIFTED VIEWS of that patch: LDS row of pixel k for tap (kh,kw) = lut[k] + kh*XW + kw,
//     lut[] built once per block (the tile geometry is the same for every tile), so no address decode in the k loop;
//   * both operands have the contraction index (pixels) as the slow LDS dimension: MFMA fragments are fetched with the
//     CDNA4 transpose read ds_read_b64_tr_b16; a 16-byte-chunk XOR by bit 1 of the LDS row keeps the 4-row x 64-byte
//     footprint of a transpose read on distinct banks (rows r and r+2 are 256 bytes apart);
//   * 4 waves = 2 (co halves) x 2 (ci halves), each 9 accumulators of v_mfma_f32_32x32x16_bf16 (144 AGPRs);
//   * split over the pixel tiles across blockIdx.z, f32 atomics straight from the accumulators (a wave's atomic
//     instruction covers 2 rows x 32 consecutive ci = whole 128-byte lines).
// Staged bytes per MFMA drop ~9x and the per-load integer work disappears from the inner loop.
#include <math.h>
#include "prims.h"
#include "avsr_hip.h"

namespace {

struct WgParams {
    const bf16_t* dy;
    const bf16_t* x;
    float* dw;
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
struct RowDesc { int off, meta; };  // staging descriptor of one LDS row: byte offset from the tile's base pointer,
                                    // image | row << 8 (row = kNever for rows that are always zero)
constexpr int kNever = 0x400000;

AVSR_DEV int swz_byte(int row, int col_byte) { return row * 128 + (col_byte ^ ((row & 2) << 5)); }

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(WgParams p) {
    AVSR_DYN_SMEM(smem);
    const int lane = threadIdx.x & 63, wave = wave_id(), tid = threadIdx.x;
    const int wm = wave >> 1, wn = wave & 1;
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
    const int stage_bytes = (p.KP + p.XROWS) * 128;
    int* lut = reinterpret_cast<int*>(smem + STAGES * stage_bytes);  // [KP]: patch row of pixel k, tap (0,0)
    RowDesc* desc = reinterpret_cast<RowDesc*>(lut + p.KP);                // [KP + XROWS]: staging descriptor of every LDS row
    const int kvalid = p.G * p.R * p.OW;

    // ---- once per block: the tile geometry (identical for every tile)
    for (int k = tid; k < p.KP; k += 256) {
        int g = 0, y = 0, xx = 0;
        const bool ok = k < kvalid;
        if (ok) {
            g = k / (p.R * p.OW);
            const int rem = k - g * p.R * p.OW;
            y = rem / p.OW;
            xx = rem - y * p.OW;
        }
        lut[k] = ok ? (g * p.XR + y * p.S) * p.XW + xx * p.S : 0;
        // dy row k: pixel offset relative to (n0, r0), image g, row y inside the band; .y < 0 = always zeros
        desc[k] = ok ? RowDesc{((g * p.OH + y) * p.OW + xx) * p.Cout * 2, g | (y << 8)} : RowDesc{0, kNever << 8};
    }
    for (int j = tid; j < p.XROWS; j += 256) {
        const int g = j / (p.XR * p.XW);
        const int rem = j - g * p.XR * p.XW;
        const int yy = rem / p.XW, xx = rem - yy * p.XW;
        const bool ok = g < p.G && xx >= 1 && xx <= p.W;  // column padding is static, row padding depends on the band
        desc[p.KP + j] = ok ? RowDesc{((g * p.H + yy) * p.W + xx) * p.Cin * 2, g | (yy << 8)} : RowDesc{0, kNever << 8};
    }
    __syncthreads();

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    const int t_begin = blockIdx.z * p.tiles_per_block;
    const int t_end = min(p.ntiles, t_begin + p.tiles_per_block);

    // stage tile t: every wave instruction moves 8 LDS rows (64 lanes x 16 B); rows are padded to whole instructions.
    // Fixed trip counts (KP <= 256, XROWS <= 288) with wave-uniform guards: the descriptors of all of a lane's rows are
    // fetched first, then turned into addresses -- one LDS round trip per tile instead of one per row.
    constexpr int DY_PASSES = 8, X_PASSES = 9;
    const int KP = p.KP, XROWS = p.XROWS, nbands = p.nbands, G = p.G, R = p.R, S = p.S, N = p.N, H = p.H, OH = p.OH;
    const char* const zero = reinterpret_cast<const char*>(p.zero);
    // the 16-byte chunk a lane fetches: physical chunk pc of row (row0 + rsub) holds source chunk pc ^ 4*bit1(row), and
    // row0 is a multiple of 8, so the chunk is a per-lane constant
    const int rsub = lane >> 3;
    const int chunk_bytes = ((lane & 7) ^ ((rsub & 2) << 1)) * 16;
    const char* const dy_lane = reinterpret_cast<const char*>(p.dy + co0) + chunk_bytes;
    const char* const x_lane = reinterpret_cast<const char*>(p.x + ci0) + chunk_bytes;
    auto issue = [&](int t, char* stage) {
        const int grp = t / nbands;
        const int n0 = grp * G, r0 = (t - grp * nbands) * R;
        const int gmax = N - n0;                 // images g >= gmax do not exist
        const int ymax = OH - r0;                // band rows y >= ymax do not exist
        const int ytop = r0 * S - 1;             // image row of patch row 0
        const char* const dy_tile = dy_lane + (((long)n0 * OH + r0) * p.OW) * p.Cout * 2;
        const char* const x_tile = x_lane + (((long)n0 * H + ytop) * p.W - 1) * p.Cin * 2;
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
                const bool ok = (dd[j].meta & 255) < gmax && (dd[j].meta >> 8) < ymax;
                glds16(ok ? dy_tile + dd[j].off : zero, stage + row0 * 128);
            }
        }
        char* xs = stage + KP * 128;
#pragma unroll
        for (int j = 0; j < X_PASSES; j++) {
            const int row0 = wave * 8 + j * 32;
            if (row0 < XROWS) {
                const bool ok = (dx[j].meta & 255) < gmax && (unsigned)(ytop + (dx[j].meta >> 8)) < (unsigned)H;
                glds16(ok ? x_tile + dx[j].off : zero, xs + row0 * 128);
            }
        }
    };

    if (t_begin < t_end) issue(t_begin, smem);
    // per-lane constants of the transpose reads (prims.h lds_tr16): lane (g4, i) addresses row 8*(g4>>1) + (i>>2) (+4)
    // of a 16-row k-step, 4 consecutive columns starting at 16*(g4&1) + 4*(i&3) of the wave's 32-column slice
    const int g4 = lane >> 4, li = lane & 15;
    const int krow = 8 * (g4 >> 1) + (li >> 2);
    const int acol = (wm * 32 + 16 * (g4 & 1) + 4 * (li & 3)) * 2;  // byte column in the dy tile
    const int bcol = (wn * 32 + 16 * (g4 & 1) + 4 * (li & 3)) * 2;  // byte column in the x patch
    const int bflip = (bcol & 64) ? -64 : 64;                        // what the row-bit-1 chunk XOR does to bcol

    for (int t = t_begin; t < t_end; t++) {
        wait_vmcnt<0>();
        __syncthreads();  // tile t has landed for every wave; everyone is done with the other buffer
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end) issue(t + 1, smem + (buf ^ 1) * stage_bytes);
        const char* dys = smem + buf * stage_bytes;
        const char* xs = dys + p.KP * 128;
        for (int ks = 0; ks < p.KP / 16; ks++) {
            const int k_lo = ks * 16 + krow, k_hi = k_lo + 4;
            const int xlo = lut[k_lo] * 128 + bcol, xhi = lut[k_hi] * 128 + bcol;
            // Transpose reads in the asm form (they must not wait for the next tile's LDS-DMA, see prims.h), software
            // pipelined six taps ahead of the MFMAs: the LGKM counter is 4 bits, so at most 15 reads may be in flight.
            // Each MFMA waits only for its own fragments: lgkmcnt(n) = reads issued after them that may still be pending.
            bf16x4 alo = lds_tr16_async(reinterpret_cast<const bf16_t*>(dys + swz_byte(k_lo, acol)));
            bf16x4 ahi = lds_tr16_async(reinterpret_cast<const bf16_t*>(dys + swz_byte(k_hi, acol)));
            bf16x4 blo[9], bhi[9];
            auto read_tap = [&](int tap) {
                const int shift = ((tap / 3) * p.XW + tap % 3) * 128;  // wave-uniform
                const int rl = xlo + shift, rh = xhi + shift;          // bit 8 = bit 1 of the patch row
                blo[tap] = lds_tr16_async(reinterpret_cast<const bf16_t*>(xs + rl + ((rl >> 8) & 1) * bflip));
                bhi[tap] = lds_tr16_async(reinterpret_cast<const bf16_t*>(xs + rh + ((rh >> 8) & 1) * bflip));
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
                acc[tap] = mfma32(a, b, acc[tap]);
                if (tap + AHEAD < 9) read_tap(tap + AHEAD);
                sched_fence();  // keep each MFMA right behind its own wait
            }
        }
    }

    // ---- dwp[co][tap][ci] += acc
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int ci = ci0 + wn * 32 + (lane & 31);
            atomicAdd(p.dw + ((size_t)co * 9 + tap) * p.Cin + ci, acc[tap][r]);
        }
}

}  // namespace

// dwp[Cout][3][3][Cin] (f32, caller zeroes) += dy[N,OH,OW,Cout]^T (x) shifted x[N,H,W,Cin]; 3x3, pad 1, stride 1 or 2;
// Cin % 64 == 0, Cout % 64 == 0; zero_page: >= 16 zero bytes of device memory
extern "C" int avsr_conv3x3_wgrad_bf16(const void* dy, const void* x, float* dwp, const void* zero_page, int N, int H, int W,
                                       int Cin, int Cout, int stride, hipStream_t stream) {
    AVSR_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, "conv3x3_wgrad_bf16: channel counts must be multiples of 64");
    AVSR_REQUIRE(stride == 1 || stride == 2, "conv3x3_wgrad_bf16: stride must be 1 or 2");
    AVSR_REQUIRE(zero_page != nullptr, "conv3x3_wgrad_bf16: zero page required");
    AVSR_REQUIRE((long)N * H * W < (1l << 31), "conv3x3_wgrad_bf16: pixel count exceeds int32");
    if (N <= 0) return 0;
    const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
    WgParams p{};
    p.dy = reinterpret_cast<const bf16_t*>(dy); p.x = reinterpret_cast<const bf16_t*>(x); p.dw = dwp;
    p.zero = reinterpret_cast<const bf16_t*>(zero_page);
    p.N = N; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.Cin = Cin; p.Cout = Cout; p.S = stride;
    p.XW = (OW - 1) * stride + 3;
    // tile geometry: G whole images (small images) or one band of R output rows; at most 256 pixels and 38 KiB per
    // stage (two stages, two blocks per CU); maximise useful pixels per unit of max(MFMA time, staging time)
    double best = -1.0;
    for (int mode = 0; mode < 2; mode++)
        for (int v = 1; v <= (mode == 0 ? OH : 64); v++) {
            const int G = mode == 0 ? 1 : v, R = mode == 0 ? v : OH;
            if (G > N || G > 255) break;
            const int XR = (R - 1) * stride + 3;
            const int KP = (G * R * OW + 15) / 16 * 16, XROWS = (G * XR * p.XW + 7) / 8 * 8;
            if (KP > 256 || XROWS > 288 || (KP + XROWS) * 128 > 38 * 1024) break;
            const int nb = (OH + R - 1) / R;
            const double useful = (double)G * OH * OW / nb;                      // real pixels per tile (average)
            const double cost = fmax(KP / 16 * 9 * 32.0, (KP + XROWS) * 128 / 20.0) + 300.0;  // clocks per tile
            if (useful / cost > best) {
                best = useful / cost;
                p.G = G; p.R = R; p.XR = XR; p.KP = KP; p.XROWS = XROWS; p.nbands = nb;
            }
        }
    AVSR_REQUIRE(best > 0.0, "conv3x3_wgrad_bf16: image row too wide for one LDS tile");
    p.ntiles = (N + p.G - 1) / p.G * p.nbands;
    const int pairs = (Cin / 64) * (Cout / 64);
    int split = (512 + pairs - 1) / pairs;  // two blocks per CU
    if (split > p.ntiles) split = p.ntiles;
    p.tiles_per_block = (p.ntiles + split - 1) / split;
    split = (p.ntiles + p.tiles_per_block - 1) / p.tiles_per_block;
    const size_t lds = (size_t)STAGES * (p.KP + p.XROWS) * 128 + (size_t)p.KP * 4 + (size_t)(p.KP + p.XROWS) * 8;
    dim3 grid(Cin / 64, Cout / 64, split), block(256);
    AVSR_LAUNCH(conv3x3_wgrad_kernel, grid, block, lds, stream, p);
    AVSR_CHECK_LAUNCH("conv3x3_wgrad_bf16");
    return 0;
}
