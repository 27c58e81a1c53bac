// conv3x3_c64.hip -- 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels on channels-last bf16
// images, forward and data gradient: the four BasicBlock convolutions of the ResNet-18 trunk's first stage and their data
// gradients (frontend/resnet.py:10-35,82-98 under autograd; 22 x 22 x 64 images, one per video frame).
//
// Why a kernel of its own.  On the tiled implicit-GEMM kernel (gemm_fast_kernel.h, 128 x 64 tile) these launches are the
// slowest convolutions of the trunk per FLOP (125-135 us for 57 GFLOP, 17 % of the MFMA peak): K = 9 * 64 is only nine
// k-tiles, every one of them re-fetches its 128 pixel rows from L2 (each input pixel crosses L2 -> LDS nine times) together
// with the whole 64 x 64 weight slice, 221 KB of operand stream per 16 KB of output -- the launch is bound by L2 -> CU
// delivery (12 TB/s chip-wide), not by the MFMA pipe.  With C = 64 everything that is re-read fits on chip:
//  * WEIGHTS IN REGISTERS: the whole filter is 64 x 576 bf16 = 72 KB; wave (mg, ng) of the FOUR-wave block (one wave per SIMD,
//    so each may use the 512-entry register file) keeps the 32 output channels x 576 slice it multiplies with (144 VGPRs per
//    lane) for the life of the block and owns 128 of the tile's pixels (four accumulator tiles) -- no LDS reads and no
//    re-fetch for the B operand at all (first version: 8 waves on a 256-register budget: the staging plan spilled next to the
//    weights, and a scratch reload waits on vmcnt(0), i.e. on the LDS-DMA in flight);
//  * PATCH IN LDS: a tile is a band of R full image rows (R * W <= 256 output pixels); its zero-padded (R+2) x (W+2) x 64
//    input patch is staged ONCE by LDS-DMA (128-byte pixels, XOR-swizzled chunks) and the nine taps are nine shifted views
//    of it: 40 KB in for 31 KB out;
//  * PERSISTENT BLOCKS, one per CU, loop over tiles with the next patch in flight (two patch buffers) while the current one
//    is multiplied;
//  * the product is formed TRANSPOSED (D^T = W X^T: accumulator rows = channels, columns = pixels), so a lane ends up with four
//    consecutive channels of one pixel per register quad -> 8-byte LDS stores into the [pixel][64] staging row, 16-byte
//    coalesced stores (+ the residual of the data-gradient chain) from there.
// Data gradient: the same kernel on the [Cin][tap][Cout] weight copy with the tap order reversed (dx[p] = sum_t dy[p - off_t] W_t^T).
#include "prims.h"
#include "avsr_hip.h"

namespace {

template <class T> AVSR_DEV void opaque(T& v) {  // the value is unchanged, but the compiler may not reason about it
#ifndef AVSR_EMU
    asm volatile("" : "+v"(v));
#endif
}

constexpr int C = 64, KTOT = 9 * C;
constexpr int OST_BYTES = 256 * 128;  // output staging tile: [256 pixels][64] bf16, 128-byte rows, chunks XOR-swizzled like the patch
constexpr int NTHR = 256, NWAVE = 4;  // one wave per SIMD: each may use the whole 512-register file
constexpr int PR_LIMIT = 384;                                       // patch rows (pixels) per buffer
constexpr int DMA_PER_WAVE = (PR_LIMIT / 8 + NWAVE - 1) / NWAVE;    // LDS-DMA wave-instructions per wave and patch: 12

struct C64Params {
    const bf16_t* src;    // [N][H][W][64]
    const bf16_t* wq;     // [64 n][9 taps][64 k]  (forward: n = co, k = ci; data gradient: n = ci, k = co)
    const bf16_t* resid;  // [N][H][W][64] or null: added to the result
    bf16_t* out;          // [N][H][W][64]
    const void* zero;     // >= 16 zero bytes
    int N, H, W, R, bands, ntiles;
    int patch_bytes;      // bytes of one patch buffer ((R+2) * (W+2) pixels of 128 bytes, rounded up to 1 KiB)
    int abl;              // benchmarks only (avsr_tune knob 13): bit 0 = no MFMA loop, bit 1 = no patch staging after the first, bit 2 = no copy-out
};

// FLIP (data gradient: taps reversed) is a template parameter so that every tap offset is a compile-time multiple of the patch pitch.
// XABL (benchmarks only, compile-time so that the measured loop carries no switches): 1 = no fragment reads after the first two
// quads (MFMAs on stale registers), 2 = no MFMAs (fragment reads only)
template <bool FLIP, int XABL = 0>
__global__ __launch_bounds__(NTHR) void conv3x3_c64_kernel(C64Params p) {
    AVSR_DYN_SMEM(smem);
    char* ost0 = smem + 2 * p.patch_bytes;
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int mg = wave >> 1, ng = wave & 1;  // this wave's 128 pixels (four 32-pixel accumulator tiles) and 32 channels
    const int W2 = p.W + 2, prows = (p.R + 2) * W2;

    // ---- this wave's half of the filter in registers: n = 32 ng + (lane & 31), k-step s: k = 16 s + 8 (lane >> 5) .. + 8
    bf16x8 bw[KTOT / 16];
    {
        const bf16_t* wrow = p.wq + (size_t)(ng * 32 + (lane & 31)) * KTOT + 8 * (lane >> 5);
#pragma unroll
        for (int s = 0; s < KTOT / 16; s++) bw[s] = *reinterpret_cast<const bf16x8*>(wrow + 16 * s);
    }
    // the weights are complete BEFORE the tile loop as far as the compiler is concerned: otherwise it waits for them with a
    // vmcnt(0) at their first use inside the loop -- which also waits for the next patch's LDS-DMA, every iteration
#pragma unroll
    for (int s = 0; s < KTOT / 16; s++) opaque(bw[s]);

    // ---- LDS-DMA of a patch: wave-instruction j = wave + 8 i stages patch rows 8 j .. 8 j + 7, lane -> (row 8 j + (lane >> 3),
    // physical chunk lane & 7).  The lane's first patch row is decoded once (py0, px0); successive instructions of a wave are
    // 8 * NWAVE rows apart, so (py, px) advance by that distance's (quotient, remainder) by W2 with one carry -- no division in the tile loop (the first
    // version re-divided per instruction: 200 instructions per tile, 11 us of an 82 us launch).
    const int step_y = (8 * NWAVE) / W2, step_x = 8 * NWAVE - step_y * W2;
    int py0, px0;
    {
        const int pr = wave * 8 + (lane >> 3);
        py0 = pr / W2;
        px0 = pr - py0 * W2;
    }
    struct Stager {  // walks one patch's wave-instructions; one() issues the i-th of this wave
        const bf16_t* base;
        char* buf;
        int y0, py, px;
    };
    auto stage_begin = [&](int tile, char* buf) {
        const int n = tile / p.bands, y0 = (tile - n * p.bands) * p.R;
        return Stager{p.src + ((size_t)n * p.H + y0) * p.W * C, buf, y0, py0, px0};
    };
    auto stage_one = [&](Stager& st, int i) {
        const int j = wave + NWAVE * i;
        if (j * 8 >= prows) return;  // wave-uniform
        const int pr = j * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((pr >> 1) & 7);  // source chunk that lands in physical chunk lane & 7
        const int yy = st.y0 - 1 + st.py;
        const bool ok = pr < prows && st.px >= 1 && st.px <= p.W && yy >= 0 && yy < p.H;
        const void* src = ok ? (const void*)(st.base + ((st.py - 1) * p.W + (st.px - 1)) * C + c * 8) : p.zero;
        glds16(src, st.buf + j * 1024);
        st.px += step_x;
        st.py += step_y;
        if (st.px >= W2) {  // (W2 >= 6 and a step of 32 rows: the remainder is < W2, one carry suffices)
            st.px -= W2;
            st.py++;
        }
    };

    // ---- this lane's four accumulator columns (pixels) m_i = mg * 128 + 32 i + (lane & 31), and for each of them and each tap
    // the LDS offset of its fragment inside a patch buffer, with the swizzle key and the lane's chunk parity already folded in:
    //     offset(tap, i, ks) = fbase[tap][i] ^ (ks << 5)         (row * 128 has zeros where the XOR terms live)
    // -- ONE VALU instruction per fragment read in the tile loop.  (Computed per fragment from the pixel coordinates, the address
    // arithmetic was 6-7 instructions per ds_read and the loop was bound by VALU issue, not by the MFMA pipe: 38 us of a 90 us
    // launch with the MFMAs removed.)  36 registers, tile-invariant; a block is one wave per SIMD, the register file is there.
    const int khalf = lane >> 5;
    int fbase[9][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int pix = mg * 128 + 32 * i + (lane & 31);
        if (pix >= p.R * p.W) pix = 0;  // idle column: reads a valid pixel, never stored
        const int y = pix / p.W, x = pix - y * p.W;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int tt = FLIP ? 8 - tap : tap;  // patch offset of weight tap `tap`
            const int pr = y * W2 + x + (tt / 3) * W2 + (tt % 3);
            fbase[tap][i] = (pr * 128) ^ (((pr >> 1) & 7) << 4) ^ (khalf << 4);
        }
    }

    // copy-out of a finished tile: the band's pixels are contiguous in the output; thread -> 8 x 16 bytes (+ the residual of the
    // data-gradient chain, which load_resid requested one MFMA phase earlier: 32 of the 512 registers of a one-wave-per-SIMD block)
    constexpr int NCP = 2048 / NTHR;
    struct Band {  // where a tile's pixels live in the output / residual
        size_t g0;
        int npix8;
    };
    auto band_of = [&](int tile) {
        const int n = tile / p.bands, y0 = (tile - n * p.bands) * p.R;
        return Band{((size_t)n * p.H + y0) * p.W * C, min(p.R, p.H - y0) * p.W * 8};
    };
    // one 16-byte piece of the copy-out in two halves: the LDS read, and -- a k-step later, when its result has arrived under the
    // MFMAs in between -- the residual add and the store
    auto copy_read = [&](const Band& b, const char* ost, int j) {
        const int idx = threadIdx.x + NTHR * j;
        const int pix = idx < b.npix8 ? idx >> 3 : 0, c = idx & 7;
        return *reinterpret_cast<const bf16x8*>(ost + pix * 128 + ((c ^ ((pix >> 1) & 7)) << 4));
    };
    auto copy_store = [&](const Band& b, bf16x8 v, const bf16x8 (&rr)[NCP], int j) {
        const int idx = threadIdx.x + NTHR * j;
        if (idx >= b.npix8) return;
        if (p.resid) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = (short)f2bf(bf2f((bf16_t)v[e]) + bf2f((bf16_t)rr[j][e]));
        }
        *reinterpret_cast<bf16x8*>(p.out + b.g0 + (size_t)(idx >> 3) * C + (idx & 7) * 8) = v;
    };
    auto resid_one = [&](const Band& b, bf16x8 (&rr)[NCP], int j) {
        const int idx = threadIdx.x + NTHR * j;
        if (idx < b.npix8) rr[j] = *reinterpret_cast<const bf16x8*>(p.resid + b.g0 + (size_t)(idx >> 3) * C + (idx & 7) * 8);
    };

    // Tile loop.  Per iteration: [vmcnt(0) + barrier] the patch of this tile has landed, the previous tile's result is in the
    // staging tile and its residual in registers -> 36 k-steps of four MFMAs each; INTO those steps are woven, one piece per
    // step, everything else a tile needs (a block is one wave per SIMD: nothing else could hide it): the 12 LDS-DMA instructions
    // of the NEXT patch, the 8 x 16-byte copy-out pieces of the PREVIOUS tile (its stores then have the rest of the MFMA phase
    // to complete before the next vmcnt(0): stores count in vmcnt on this chip) and, after those, the 8 residual loads of THIS
    // tile -> [barrier: the staging tile has been copied out by everybody] -> accumulators -> staging tile.
    int tile = blockIdx.x;
    if (tile < p.ntiles) {
        Stager st = stage_begin(tile, smem);
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; i++) stage_one(st, i);
    }
    int it = 0, prev = -1;
    char* ost = ost0;
    bf16x8 rr[NCP];
#pragma unroll
    for (int j = 0; j < NCP; j++) rr[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    static_assert(DMA_PER_WAVE + 2 * NCP + 1 <= KTOT / 16, "the woven pieces must fit the k-steps");
    for (; tile < p.ntiles; tile += gridDim.x, it++) {
        char* patch = smem + (it & 1) * p.patch_bytes;
        wait_vmcnt<0>();
#pragma unroll
        for (int j = 0; j < NCP; j++) opaque(rr[j]);  // the residual registers are complete HERE as far as the compiler is concerned
                                                      // (its own wait would come after the DMA below was issued, and cover it)
        block_barrier_raw();
        const bool do_stage = tile + (int)gridDim.x < p.ntiles && !(p.abl & 2);
        const bool do_copy = prev >= 0 && !(p.abl & 4);
        Stager st = stage_begin(do_stage ? tile + (int)gridDim.x : tile, smem + ((it + 1) & 1) * p.patch_bytes);
        const Band bprev = band_of(prev >= 0 ? prev : tile), bcur = band_of(tile);

        f32x16 acc[4];  // [pixel tile i]
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
        // 36 k-steps (tap, 16 channels); the four pixel fragments of step s + 2 are requested right after the MFMAs of step s
        // were issued (into the registers those MFMAs read: they are issued, hence have read their operands)
        auto frag_quad = [&](int s, i32x4 (&xf)[4]) {
            const int tap = s >> 2, ks = s & 3;
#pragma unroll
            for (int i = 0; i < 4; i++) xf[i] = lds_read16_async(patch + (fbase[tap][i] ^ (ks << 5)));
        };
        // Order inside a step: the four MFMAs FIRST, then the fragment requests of step s + 2 and the woven piece.  The
        // fragment reads are in the caller-ordered asm form (prims.h): with plain LDS loads the compiler put a full lgkmcnt(0)
        // in front of every step's MFMAs (an LDS-DMA is always in flight here), i.e. each step also waited for the fragments of
        // step s + 2 it had just requested -- one exposed LDS round trip per 128 MFMA cycles.  Now a step waits for its own
        // four fragments only: lgkmcnt(4) = the four requests of the following step may still be out (reads return in order;
        // a woven LDS read issued in between only makes the wait slightly conservative).
        i32x4 xf[2][4];
        bf16x8 cv = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        frag_quad(0, xf[0]);
        frag_quad(1, xf[1]);
#pragma unroll
        for (int s = 0; s < KTOT / 16; s++) {
            if (s + 1 < KTOT / 16) lds_wait<4>();
            else lds_wait<0>();
#pragma unroll
            for (int i = 0; i < 4; i++) {
                lds_tie(xf[s & 1][i]);
                if (XABL != 2) acc[i] = mfma32(bw[s], __builtin_bit_cast(bf16x8, xf[s & 1][i]), acc[i]);  // rows = channels, columns = pixels
            }
            sched_fence();
            if (s + 2 < KTOT / 16 && XABL != 1) frag_quad(s + 2, xf[s & 1]);
            // the woven piece of this step
            if (s < DMA_PER_WAVE) {
                if (do_stage) stage_one(st, s);
            } else if (s <= DMA_PER_WAVE + NCP) {
                const int j = s - DMA_PER_WAVE;
                if (do_copy && j > 0) copy_store(bprev, cv, rr, j - 1);
                if (do_copy && j < NCP) cv = copy_read(bprev, ost, j);
            } else if (s <= DMA_PER_WAVE + 2 * NCP) {
                if (p.resid) resid_one(bcur, rr, s - DMA_PER_WAVE - NCP - 1);
            }
            sched_fence();
        }
        block_barrier_raw();  // every wave has copied its share of the previous tile out of the staging tile
        // ---- accumulators -> staging tile [pixel][64] bf16: register quad q of accumulator i = channels
        // 32 ng + 8 q + 4 khalf .. +3 of pixel mg*128 + 32 i + (lane & 31)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int pix = mg * 128 + 32 * i + (lane & 31);
            char* orow = ost + pix * 128 + (4 * khalf) * 2;
            const int key = (pix >> 1) & 7;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                bf16x4* dst = reinterpret_cast<bf16x4*>(orow + (((ng * 4 + q) ^ key) << 4));
                *dst = bf16x4{(short)f2bf(acc[i][4 * q]), (short)f2bf(acc[i][4 * q + 1]), (short)f2bf(acc[i][4 * q + 2]),
                              (short)f2bf(acc[i][4 * q + 3])};
            }
        }
        lds_wait<0>();
        sched_fence();
        prev = tile;
    }
    if (prev >= 0) {
        wait_vmcnt<0>();  // (the last tile's residual)
        block_barrier_raw();
        if (!(p.abl & 4)) {
            const Band b = band_of(prev);
#pragma unroll
            for (int j = 0; j < NCP; j++) copy_store(b, copy_read(b, ost, j), rr, j);
        }
    }
}

}  // namespace

static int c64_patch_bytes(int H, int W, int* R_out) {
    int R = 256 / W;
    if (R > H) R = H;
    if (R_out) *R_out = R;
    return ((R + 2) * (W + 2) * 128 + 1023) / 1024 * 1024;
}

// 1 when the geometry is one this kernel takes (the caller falls back to the tiled kernel otherwise)
int avsr_conv3x3_c64_supported(int H, int W) {
    if (W < 4 || W > 254 || H < 1) return 0;
    int R;
    const int pb = c64_patch_bytes(H, W, &R);
    return R >= 1 && (R + 2) * (W + 2) <= PR_LIMIT && 2 * pb + OST_BYTES <= 160 * 1024;
}

// out[N,H,W,64] = conv3x3(src[N,H,W,64], wq) (+ resid); flip = 0: forward with wq = [Cout][3][3][Cin];
// flip = 1: data gradient with wq = [Cin][3][3][Cout] (avsr_conv_weight_permute to_dgrad = 1), src = dy
int avsr_conv3x3_c64_launch(int flip, const void* src, const void* wq, const void* resid, void* out, const void* zero_page, int N,
                            int H, int W, hipStream_t stream) {
    C64Params p{};
    p.src = (const bf16_t*)src; p.wq = (const bf16_t*)wq; p.resid = (const bf16_t*)resid; p.out = (bf16_t*)out;
    p.zero = zero_page;
    p.N = N; p.H = H; p.W = W;
    p.patch_bytes = c64_patch_bytes(H, W, &p.R);
    p.bands = (H + p.R - 1) / p.R;
    p.ntiles = N * p.bands;
    p.abl = avsr_tune_knobs[13];
    const int grid = p.ntiles < 256 ? p.ntiles : 256;  // one persistent block per CU
    const size_t lds = 2 * (size_t)p.patch_bytes + OST_BYTES;
    if (p.abl & 8) AVSR_LAUNCH((conv3x3_c64_kernel<false, 1>), dim3(grid), dim3(NTHR), lds, stream, p);
    else if (p.abl & 16) AVSR_LAUNCH((conv3x3_c64_kernel<false, 2>), dim3(grid), dim3(NTHR), lds, stream, p);
    else if (flip) AVSR_LAUNCH(conv3x3_c64_kernel<true>, dim3(grid), dim3(NTHR), lds, stream, p);
    else AVSR_LAUNCH(conv3x3_c64_kernel<false>, dim3(grid), dim3(NTHR), lds, stream, p);
    return 0;
}
