// conv_patch.hip (round 6) -- 3x3 / stride 1 / pad 1 convolutions of the ResNet trunk's 128 - 512-channel stages (small images:
// 11x11, 6x6, 3x3 per video frame; frontend/resnet.py:10-35,82-98 under autograd), forward and data gradient, with the A operand
// PATCH-STAGED instead of im2col-tiled.
//
// Why.  On the tiled implicit-GEMM kernel (gemm_fast_kernel.h) every k-tile = (filter tap, 64 input channels) re-fetches its
// 128 / 256 pixel rows from L2: each input pixel crosses L2 -> LDS nine times, and that delivery (18 - 25 B/clk per CU however it
// is fed, profiles/r2_feed_probe.txt) is what those launches are bound by -- stage 3 with two weight planes moves 2.1 GB per launch
// (DESIGN.md section 5).  Here a block owns WHOLE images: 256-row tiles of IPT = floor(256 / (H W)) images (252, 252, 242 rows
// for 3x3, 6x6, 11x11 images), and for each 64-channel chunk of the input the tile's pixels are staged ONCE (32 KB, LDS-DMA,
// 128-byte pixel rows, XOR-swizzled chunks) and the nine taps are nine views of that patch: a lane's A fragment of tap t is
// read from the LDS row of pixel (y + ty - 1, x + tx - 1) of its image -- or from a row of zeros when that falls outside -- through a
// per-lane table of nine offsets per accumulator row block (18 registers), one XOR per read.  Only the weights stream through the
// operand ring.  Per block and k-tile the L2 -> LDS traffic drops from 32 + 32 KB (two planes) / 32 + 16 KB (one) to 3.6 + 32 / 3.6 + 16.
//
// Structure: 8 waves (4 x 2; two per SIMD), each a 64 x (BN / 2) block of 32x32x16 MFMA accumulators; k loop chunk-major
// (chunk c, tap t): B tiles [BN][64] (x 2 planes) in a 2-stage LDS-DMA ring, one raw barrier per k-tile, fragments double
// buffered in registers as in gemm_fast_kernel.h; the patch is double buffered per chunk, the four DMA instructions per wave of
// chunk c + 1 ride in the first four taps of chunk c; epilogue through LDS (gemm_core.h epilogue_lds: f16 + bf16 twin / bf16 +
// residual) with a row map that drops the tile rows beyond IPT images.
// FLIP = data gradient: the same kernel on dy and the [Cin][3][3][Cout] weight copy, tap offsets mirrored.
#include "gemm_core.h"
#include "avsr_hip.h"

namespace {

using avsr_gemm_impl::Params;

template <int BN, int F16, int WP, bool FLIP, int STAGES_ = 2>
struct PatchKernel {
    static constexpr int BM = 256, BK = 64, WGM = 4, WGN = 2, NW = 8, NTHR = 512;
    static constexpr int WM = 64, WN = BN / WGN, TM = 2, TN = WN / 32;
    static constexpr int PATCH_BYTES = BM * 128;  // 256 pixel slots of 64 channels; slot 255 is never a pixel: a row of zeros
    static constexpr int B_BYTES = BN * 128, STAGE_BYTES = WP * B_BYTES, STAGES = STAGES_;
    static_assert(STAGES == 2 || STAGES == 3, "weight ring: two or three stages");
    static constexpr int B_LOADS = BN / (8 * NW);    // wave-instructions per wave, stage and plane
    static constexpr int P_LOADS = BM / (8 * NW);    // ... per wave and patch chunk: 4
    static constexpr int LPT = WP * B_LOADS;         // weight LDS-DMA instructions per wave and k-tile
    static constexpr size_t RING_OFF = 2 * PATCH_BYTES, RING_END = RING_OFF + (size_t)STAGES * STAGE_BYTES;
    static constexpr size_t EPI_BYTES = (size_t)BM * (BN + 4) * 4;
    static constexpr size_t MAP_OFF = RING_END > EPI_BYTES ? RING_END : EPI_BYTES;
    static constexpr size_t TAB_OFF = MAP_OFF + BM * 4;             // int [9][BM]: per (tap, tile row) the patch offset of its source pixel
    static constexpr size_t LDS_BYTES = TAB_OFF + 9 * BM * 4;
    static_assert(B_LOADS >= 1 && TN >= 1, "tile / wave-count mismatch");
    static constexpr int ZSLOT = BM - 1;

    static AVSR_DEV bf16x8 frag_b(const char* base, int r, int chunk) {
        return *reinterpret_cast<const bf16x8*>(base + r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4));
    }

    // p.cN images of p.cH x p.cW pixels, p.cC gathered channels (multiple of 64); p.cT = IPT (images per tile); p.gate = zero page
    static AVSR_DEV void run(const Params& p, char* smem) {
        const bf16_t* X = reinterpret_cast<const bf16_t*>(p.A);
        const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
        const int lane = threadIdx.x & 63, wave = wave_id();
        const int wm = wave / WGN, wn = wave % WGN;
        const int HW = p.cH * p.cW, IPT = p.cT, rows_valid = IPT * HW;
        const int n0 = blockIdx.x * BN;
        const int img0 = blockIdx.y * IPT;
        const int m0 = img0 * HW;  // first output row of this tile (rows are pixels, image-major)
        const int nch = p.cC / BK, nt = 9 * nch;

        // ---- output row of every tile row (rows beyond the tile's images / beyond the tensor: none)
        int* rowmap = reinterpret_cast<int*>(smem + MAP_OFF);
        for (int r = threadIdx.x; r < BM; r += NTHR) rowmap[r] = (r < rows_valid && m0 + r < p.M) ? m0 + r : -1;

        // ---- patch staging: wave-instruction j = wave * P_LOADS + q covers pixel slots 8 j .. 8 j + 7; lane -> (slot 8 j + (lane >> 3),
        // physical chunk lane & 7); the slot's pixel is decoded once
        // (decoded per instruction: four per wave and 64-channel chunk -- a per-lane pointer table indexed by the run-time piece
        // number lands in scratch memory, and a scratch reload waits for every LDS-DMA in flight)
        auto stage_patch_piece = [&](int chunk, int q, char* buf) {
            const int sp = (wave * P_LOADS + q) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((sp >> 1) & 7);  // source chunk that lands in physical chunk lane & 7
            const bool ok = sp < rows_valid && m0 + sp < p.M;
            const void* src = ok ? (const void*)(X + (size_t)(m0 + sp) * p.cC + c * 8 + chunk * BK) : p.gate;  // zero page
            glds16(src, buf + (wave * P_LOADS + q) * 1024);
        };
        // ---- weights: rows n0 + ..., as FastKernel::decode_rows / issue
        const bf16_t* brow[B_LOADS];
#pragma unroll
        for (int i = 0; i < B_LOADS; i++) {
            const int r = (wave * B_LOADS + i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            brow[i] = B + (size_t)min(n0 + r, p.N - 1) * p.ldb + c * 8;
        }
        const long lo_off = WP == 2 ? reinterpret_cast<const bf16_t*>(p.B2) - B : 0;
        auto issue_b = [&](int t, char* stage) {
            const int chunk = t / 9, tap = t - chunk * 9;
            const long db = (long)tap * p.cC + chunk * BK;
#pragma unroll
            for (int i = 0; i < B_LOADS; i++) glds16(brow[i] + db, stage + (wave * B_LOADS + i) * 1024);
            if (WP == 2) {
#pragma unroll
                for (int i = 0; i < B_LOADS; i++) glds16(brow[i] + db + lo_off, stage + B_BYTES + (wave * B_LOADS + i) * 1024);
            }
        };

        // ---- per (tap, tile row): byte offset of the source pixel's row inside a patch buffer with that row's swizzle key folded in
        // -- pixel (y + ty - 1, x + tx - 1) of the row's own image, or the row of zeros.  A lane's fragment address for k-step ks is
        // (table entry ^ (khalf << 4)) ^ (ks << 5): one LDS read per (k-tile, row block), requested a k-tile ahead.  (As a per-lane
        // register table -- 18 values, the nine taps unrolled -- the two-plane variant spilled 34 registers at the 256 cap.)
        const int khalf = lane >> 5;
        int* ftab = reinterpret_cast<int*>(smem + TAB_OFF);
        for (int idx = threadIdx.x; idx < 9 * BM; idx += NTHR) {
            const int tap = idx / BM, r = idx - tap * BM;
            const bool live = r < rows_valid;
            const int nl = live ? r / HW : 0, pix = live ? r - nl * HW : 0;
            const int y = pix / p.cW, x = pix - y * p.cW;
            const int ty = tap / 3, tx = tap - ty * 3;
            const int yy = FLIP ? y + 1 - ty : y + ty - 1, xx = FLIP ? x + 1 - tx : x + tx - 1;
            const bool in = live && yy >= 0 && yy < p.cH && xx >= 0 && xx < p.cW;
            const int sp = in ? nl * HW + yy * p.cW + xx : ZSLOT;
            ftab[idx] = sp * 128 + (((sp >> 1) & 7) << 4);
        }
        const int arow0 = wm * WM + (lane & 31);

        f32x16 acc[TM][TN];
        f32x16 acc_lo[WP == 2 ? TM : 1][WP == 2 ? TN : 1];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    acc[i][j][r] = 0.f;
                    if (WP == 2) acc_lo[i][j][r] = 0.f;
                }

        // prologue: patch chunk 0 and weight tile 0
#pragma unroll
        for (int q = 0; q < P_LOADS; q++) stage_patch_piece(0, q, smem);
        issue_b(0, smem + RING_OFF);
        if (STAGES == 3 && nt > 1) issue_b(1, smem + RING_OFF + STAGE_BYTES);

        __syncthreads();  // the table (and the row map) are complete
        int fcur[TM], fnext[TM];
#pragma unroll
        for (int i = 0; i < TM; i++) fcur[i] = ftab[arow0 + i * 32] ^ (khalf << 4);
        // k loop, chunk-major
        int chunk = 0, tap = 0;
        for (int t = 0; t < nt; t++) {
            const int st = STAGES == 2 ? (t & 1) : t % 3;
            // weight tile t (and every patch piece issued before it) has landed for this wave; with three stages the loads of tile
            // t + 1 -- the youngest in the queue: a step issues its patch piece BEFORE its weight tile -- may stay in flight
            if (STAGES == 3 && t + 1 < nt) wait_vmcnt<LPT>();
            else wait_vmcnt<0>();
            block_barrier_raw();  // ... for every wave; everybody is done with tile t - 1's weight buffer and the other patch buffer
            const char* Ps = smem + (chunk & 1) * PATCH_BYTES;
            const char* Bs = smem + RING_OFF + st * STAGE_BYTES;
            bf16x8 fa[2][TM], fb[2][TN], fl[2][WP == 2 ? TN : 1];
            const int brow0 = wn * WN + (lane & 31);
            auto load_frags = [&](int set, int ks) {
                const int ch = ks * 2 + khalf;
#pragma unroll
                for (int i = 0; i < TM; i++) fa[set][i] = *reinterpret_cast<const bf16x8*>(Ps + (fcur[i] ^ (ks << 5)));
#pragma unroll
                for (int j = 0; j < TN; j++) fb[set][j] = frag_b(Bs, brow0 + j * 32, ch);
                if (WP == 2) {
#pragma unroll
                    for (int j = 0; j < TN; j++) fl[set][j] = frag_b(Bs + B_BYTES, brow0 + j * 32, ch);
                }
            };
            load_frags(0, 0);
            load_frags(1, 1);
            const int tap_n = tap == 8 ? 0 : tap + 1;
#pragma unroll
            for (int i = 0; i < TM; i++) fnext[i] = ftab[tap_n * BM + arow0 + i * 32] ^ (khalf << 4);  // the next k-tile's view
            sched_fence();
            if (tap < P_LOADS && chunk + 1 < nch) stage_patch_piece(chunk + 1, tap, smem + ((chunk + 1) & 1) * PATCH_BYTES);
            if (t + STAGES - 1 < nt) issue_b(t + STAGES - 1, smem + RING_OFF + ((t + STAGES - 1) % STAGES) * STAGE_BYTES);
            sched_fence();
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        acc[i][j] = mfma32x<F16>(fa[ks & 1][i], fb[ks & 1][j], acc[i][j]);
                        if (WP == 2) acc_lo[i][j] = mfma32x<F16>(fa[ks & 1][i], fl[ks & 1][j], acc_lo[i][j]);
                    }
                if (ks + 2 < 4) load_frags(ks & 1, ks + 2);
                sched_fence();
            }
#pragma unroll
            for (int i = 0; i < TM; i++) fcur[i] = fnext[i];
            if (++tap == 9) {
                tap = 0;
                chunk++;
            }
        }
        if (WP == 2) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[i][j][r] += acc_lo[i][j][r] * (1.0f / AVSR_H16_LO_SCALE);
        }
        Params q = p;
        q.gate = nullptr;  // (the field carried the zero page)
        avsr_gemm_impl::epilogue_lds<BM, BN, TM, TN, NTHR, 1>(acc, q, m0, n0, wm * WM, wn * WN, 0, 0, smem, rowmap, 0);
    }
};

template <int BN, int F16, int WP, bool FLIP, int STAGES>
__global__ __launch_bounds__(512) void conv_patch_kernel(Params p) {
    AVSR_DYN_SMEM(smem);
    PatchKernel<BN, F16, WP, FLIP, STAGES>::run(p, smem);
}

template <int BN, int F16, int WP, bool FLIP, int STAGES = 2>
void launch_patch(Params& p, hipStream_t stream) {
    using K = PatchKernel<BN, F16, WP, FLIP, STAGES>;
    const int tiles = (p.cN + p.cT - 1) / p.cT;
    dim3 grid((p.N + BN - 1) / BN, tiles), block(K::NTHR);
    AVSR_LAUNCH((conv_patch_kernel<BN, F16, WP, FLIP, STAGES>), grid, block, K::LDS_BYTES, stream, p);
}

}  // namespace

// 1 when the geometry is one the patch-staged kernel takes: 3x3 / stride 1 / pad 1, whole images of <= 128 pixels (at least two per
// 256-row tile), gathered channels % 64 == 0, output channels % 128 == 0, and enough tiles to fill the chip
int avsr_conv_patch_supported(int N, int H, int W, int Cg, int Cout_eff, int KH, int KW, int stride, int pad_h, int pad_w) {
    if (avsr_tune_knobs[20] == 1) return 0;  // knob 20 = 1: keep the tiled kernel (A/B runs)
    if (KH != 3 || KW != 3 || stride != 1 || pad_h != 1 || pad_w != 1) return 0;
    const int HW = H * W;
    if (HW < 1 || HW > 128 || Cg % 64 != 0 || Cout_eff % 128 != 0) return 0;
    const int ipt = 256 / HW;
    if (ipt * HW > 255) return 0;  // (pixel slot 255 of the patch must stay free: it is the row of zeros the padding taps read)
    const long tiles = (long)((N + ipt - 1) / ipt) * (Cout_eff / 128);
    return (tiles >= 200 || avsr_tune_knobs[20] >= 2) ? 1 : 0;  // (knob 20 = 2 / 3: whatever the grid size -- tests)
}

// mode: 0 = bf16 forward / 1 = bf16 data gradient (+ bf16 residual) -> bf16 out; 2 = f16 forward with one / two weight planes ->
// f16 out + optional bf16 twin.  src [N][H][W][Cg], w [Cout_eff][3][3][Cg] (pitch ldw elements), out [N][H][W][Cout_eff].
int avsr_conv_patch_launch(int mode, const void* src, const void* w, const void* w_lo, int ldw, const void* resid, void* out, void* out2,
                           const void* zero_page, int N, int H, int W, int Cg, int Cout_eff, hipStream_t stream) {
    Params p{};
    p.A = src; p.B = w; p.B2 = w_lo;
    p.K = 9 * Cg; p.lda = Cg; p.ldb = ldw ? ldw : p.K;
    p.alpha = 1.f; p.gate_scale = 1.f;
    p.nsplit = 1; p.batch_h = 1; p.nbatch = 1;
    p.gate = zero_page;
    p.cN = N; p.cH = H; p.cW = W; p.cC = Cg; p.cOH = H; p.cOW = W;
    p.cKH = 3; p.cKW = 3; p.cS = 1; p.cPH = 1; p.cPW = 1;
    p.cT = 256 / (H * W);
    p.M = N * H * W; p.N = Cout_eff; p.ldc = Cout_eff;
    p.C = out;
    if (mode == 2) {
        p.c_dtype = 2; p.C2 = out2; p.ldc2 = Cout_eff;
        if (w_lo) launch_patch<128, 1, 2, false>(p, stream);
        else if (avsr_tune_knobs[20] == 3) launch_patch<128, 1, 1, false, 3>(p, stream);
        else launch_patch<128, 1, 1, false>(p, stream);
    } else {
        p.c_dtype = 1;
        if (mode == 1) {
            p.resid = reinterpret_cast<const float*>(resid); p.resid_dtype = 1; p.ldr = Cout_eff;
            if (avsr_tune_knobs[20] == 3) launch_patch<128, 0, 1, true, 3>(p, stream);  // (knob 20 = 3: three weight stages, A/B)
            else launch_patch<128, 0, 1, true>(p, stream);
        } else {
            if (avsr_tune_knobs[20] == 3) launch_patch<128, 0, 1, false, 3>(p, stream);
            else launch_patch<128, 0, 1, false>(p, stream);
        }
    }
    return 0;
}
