// gemm_split.hip -- the tuned path of the PRECISE numerical mode's forward contractions:
//     C[M,N] = epi( A[M,K] . B[N,K]^T )      A, B f32 in HBM, k-contiguous, K % 64 == 0
// with every product formed on split hi + lo bf16 planes (3 x v_mfma_f32_32x32x16_bf16 per product, ~2^-16 relative
// error) -- the arithmetic of gemm_core.h's NS = 2 kernel, on the operand path of gemm_fast.hip:
//  * the f32 tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4) into a STAGES-deep ring, counted s_waitcnt vmcnt in
//    front of a raw s_barrier; no VGPR staging, no ds_write pass;
//  * a k-tile is 32 f32 deep: LDS image per operand stage [rows][32] f32 = 128-byte rows, the same image, swizzle ((row >> 1) & 7
//    on the 16-byte chunk index, applied to the per-lane SOURCE address on the way in and to the ds_read_b128 address on the way
//    out) and bytes per stage as the bf16 kernel's 64-deep tiles -- so the same two or three blocks stay co-resident per CU
//    (first version: 64-deep f32 tiles, 96 KiB rings, ONE block per CU: 300 tiles of a 1600 x 768 GEMM ran in two rounds,
//    38 us per launch against 7.8 us in bf16);
//  * the hi / lo split happens in registers on the MFMA fragments (8 consecutive k of one row = two ds_read_b128):
//    hi = bf16(x), lo = bf16(x - hi), exactly split_bf16<2>() of prims.h, so the result is bit-compatible with the generic
//    precise kernel up to summation order;
//  * 2 x 2 waves, each a (BM/2) x (BN/2) grid of 32x32 accumulators; epilogue through LDS (gemm_core.h epilogue_lds).
//  * "split8" layout: every group of 8 consecutive f32 (32 bytes) replaced by its 8 hi bf16 (16 bytes) followed by its 8 lo bf16
//    -- the same bytes, pitches and DMA addressing as the f32 matrix, and an MFMA fragment (8 consecutive k of a row) is then
//    two plain ds_read_b128 with no VALU work.  The B operand (weights) arrives pre-split from HBM (avsr_split_pack, once per
//    step); the A tile (activations, f32 in HBM) is converted IN PLACE in LDS once per stage by the whole block (ACV: thread
//    (row, group) rewrites its own 32 bytes) instead of on every fragment of every wave that reads it.  The split costs ~3 VALU
//    instructions per element (v_cvt_pk_bf16_f32 twice, a shift, a subtract): done on the fragments, the 64x64 tile issued 115
//    VALU instructions per 6 MFMAs (each element split by two waves, both operands) and the K = 3072 GEMMs were VALU-bound
//    (30 us against 20 us in bf16);
// CV = 1: A is the im2col view of a channels-last f32 image (convolution forward, channels % 64 == 0), gathered like
// gemm_fast_kernel.h does: one pointer + tap-validity mask per staged row, a zero page for the padding taps.
//
// Replaces, in the precise / hpf modes: every nn.Linear / k=1 Conv1d forward of the encoder, decoder and heads
// (positionwise_feed_forward.py:24-30, attention.py:31-34,123, conformer_encoder.py:24,27, e2e_asr_conformer.py:31, ctc.py:21,
// transformer_decoder.py:225) and the ResNet trunk's forward convolutions (frontend/resnet.py:10-35).
// AVSR_CXXFLAGS: -fno-slp-vectorize
// (SLP vectorisation packs the per-element subtractions of the split into v_pk_add_f32, which is slower than two v_sub_f32
//  beside MFMAs on this chip)
#include "gemm_core.h"
#include "avsr_hip.h"

namespace {

using avsr_gemm_impl::Params;

// BSP: B is in the split8 layout.  ACV = 1: the staged A tile is converted to split8 in place, once per stage, by all threads;
// ACV = 2 (round 5): A ARRIVES in the split8 layout -- its producer (a BatchNorm + activation pass) wrote it that way, the same
// bytes as the f32 tensor -- so the stage needs no conversion pass and no second barrier.
// WGM x WGN: wave grid -- 4 waves, or (round 6) 8 waves = two per SIMD on 256-row tiles: the A tile of a convolution is the
// im2col gather (every input pixel crosses L2 -> LDS once per filter tap), so a 256 x 64 tile moves 40 KB per k-tile where two
// 128 x 64 tiles move 48, a 256 x 128 tile 48 KB where two 128 x 128 tiles move 64 -- and these launches are bound by exactly
// that delivery (gemm_fast.hip, round-6 note).
template <int BM, int BN, int STAGES, int CV, bool BSP = false, int WGM = 2, int WGN = 2, int ACV = 0>
struct SplitKernel {
    static constexpr int BK = 32, NW = WGM * WGN, NTHR = 64 * NW;
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    static constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32 x 32");
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int A_LOADS = BM / (8 * NW), B_LOADS = BN / (8 * NW);  // wave-instructions (8 rows each) per wave per stage
    static_assert(A_LOADS >= 1 && B_LOADS >= 1 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile / wave-count mismatch");
    static constexpr int LPT = A_LOADS + B_LOADS;
    static constexpr size_t RING_BYTES = (size_t)STAGES * STAGE_BYTES, EPI_BYTES = (size_t)BM * (BN + 4) * 4;
    static constexpr size_t LDS_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
    static_assert(BM % 64 == 0 && BN % 64 == 0, "tile must be a multiple of 64 x 64");

    struct Rows {
        const float* a[A_LOADS];  // CV 0: &A[row][4c] ; CV 1: &x[pixel of tap (0,0)][4c] (may lie outside the tensor)
        uint32_t mask[A_LOADS];   // CV 1: bit t set <=> tap t reads inside the image
        const float* b[B_LOADS];
    };

    static AVSR_DEV Rows decode_rows(const Params& p, const float* A, const float* B, int m0, int n0, int wave, int lane) {
        Rows ri;
        const int rsub = lane >> 3, pc = lane & 7;
#pragma unroll
        for (int i = 0; i < A_LOADS; i++) {
            const int r = (wave * A_LOADS + i) * 8 + rsub;  // row inside the tile
            const int c = pc ^ ((r >> 1) & 7);              // source chunk that lands in physical chunk pc
            if (CV == 0) {
                const int gr = min(m0 + r, p.M - 1);
                ri.a[i] = A + (size_t)gr * p.lda + c * 4;
                ri.mask[i] = 0;
            } else {
                const int m = min(m0 + r, p.M - 1);
                const int pix = p.cOH * p.cOW;
                const int n = m / pix, rem = m - n * pix;
                const int oh = rem / p.cOW, ow = rem - oh * p.cOW;
                const int ya = oh * p.cS - p.cPH, xa = ow * p.cS - p.cPW;
                ri.a[i] = A + ((long)n * p.cH * p.cW + (long)ya * p.cW + xa) * p.cC + c * 4;
                uint32_t mk = 0;
                for (int t = 0; t < p.cKH * p.cKW; t++) {
                    const int ta = t / p.cKW, tb = t - ta * p.cKW;
                    const int y = ya + ta, x = xa + tb;
                    if (y >= 0 && y < p.cH && x >= 0 && x < p.cW) mk |= 1u << t;
                }
                ri.mask[i] = mk;
            }
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; i++) {
            const int r = (wave * B_LOADS + i) * 8 + rsub;
            const int c = pc ^ ((r >> 1) & 7);
            const int gr = min(n0 + r, p.N - 1);
            ri.b[i] = B + (size_t)gr * p.ldb + c * 4;
        }
        return ri;
    }

    // stage k-tile t of this block
    static AVSR_DEV void issue(const Params& p, const Rows& ri, int kbeg, int t, char* stage, int wave) {
        long da, db;
        int tap = 0;
        if (CV == 0) {
            da = db = kbeg + t * BK;
        } else {
            const int cpt = p.cC / BK;
            tap = t / cpt;
            const int cb = (t - tap * cpt) * BK;
            const int ta = tap / p.cKW, tb = tap - ta * p.cKW;
            da = (long)(ta * p.cW + tb) * p.cC + cb;
            db = (long)tap * p.cC + cb;
        }
#pragma unroll
        for (int i = 0; i < A_LOADS; i++) {
            const float* src = ri.a[i] + da;
            if (CV != 0) src = ((ri.mask[i] >> tap) & 1u) ? src : reinterpret_cast<const float*>(p.gate);  // zero page
            glds16(src, stage + (wave * A_LOADS + i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; i++) glds16(ri.b[i] + db, stage + A_BYTES + (wave * B_LOADS + i) * 1024);
    }

    // the 8 consecutive k values [8 * half8, +8) of row r as split hi / lo planes (half8 = index of the 32-byte group in the row)
    static AVSR_DEV Frag<2> frag(const char* base, int r, int half8) {
        const int sw = (r >> 1) & 7;
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + r * 128 + (((2 * half8) ^ sw) << 4));
        const f32x4 b = *reinterpret_cast<const f32x4*>(base + r * 128 + (((2 * half8 + 1) ^ sw) << 4));
        Frag<2> f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            bf16_t s0[2], s1[2];
            split_bf16<2>(a[e], s0);
            split_bf16<2>(b[e], s1);
            f.p[0][e] = (short)s0[0];
            f.p[1][e] = (short)s0[1];
            f.p[0][e + 4] = (short)s1[0];
            f.p[1][e + 4] = (short)s1[1];
        }
        return f;
    }

    // the same 8 k values of row r of a split8 operand: hi in chunk 2 * half8, lo in chunk 2 * half8 + 1
    static AVSR_DEV Frag<2> frag_presplit(const char* base, int r, int half8) {
        const int sw = (r >> 1) & 7;
        Frag<2> f;
        f.p[0] = *reinterpret_cast<const bf16x8*>(base + r * 128 + (((2 * half8) ^ sw) << 4));
        f.p[1] = *reinterpret_cast<const bf16x8*>(base + r * 128 + (((2 * half8 + 1) ^ sw) << 4));
        return f;
    }
    // f32 -> split8 of a staged A tile, in place: thread (row, group) reads its 32 bytes and rewrites them
    static AVSR_DEV void convert_a(char* As) {
#pragma unroll
        for (int j = 0; j < BM * 4 / NTHR; j++) {
            const int idx = threadIdx.x + NTHR * j;
            const int r = idx >> 2, g = idx & 3;
            const Frag<2> f = frag(As, r, g);
            const int sw = (r >> 1) & 7;
            *reinterpret_cast<bf16x8*>(As + r * 128 + (((2 * g) ^ sw) << 4)) = f.p[0];
            *reinterpret_cast<bf16x8*>(As + r * 128 + (((2 * g + 1) ^ sw) << 4)) = f.p[1];
        }
    }

    static AVSR_DEV void run(const Params& p, char* smem) {
        const float* A = reinterpret_cast<const float*>(p.A);
        const float* B = reinterpret_cast<const float*>(p.B);
        const int lane = threadIdx.x & 63, wave = wave_id();
        const int wm = wave / WGN, wn = wave % WGN;
        const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
        const int zs = blockIdx.z;
        int kbeg = 0, nt;
        if (CV == 0) {
            kbeg = zs * p.k_chunk;
            nt = (min(p.K, kbeg + p.k_chunk) - kbeg) / BK;
        } else {
            nt = p.cKH * p.cKW * (p.cC / BK);
        }
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

        const Rows ri = decode_rows(p, A, B, m0, n0, wave, lane);
#pragma unroll
        for (int s = 0; s < STAGES - 1; s++)
            if (s < nt) issue(p, ri, kbeg, s, smem + s * STAGE_BYTES, wave);

        auto step = [&](int t, auto issue_flag) {
            constexpr bool ISSUE = decltype(issue_flag)::value;
            if (ISSUE) {
                wait_vmcnt<(STAGES - 2) * LPT>();
            } else {
                switch (nt - 1 - t) {
                    case 0: wait_vmcnt<0>(); break;
                    default: wait_vmcnt<(STAGES > 2 ? LPT : 0)>(); break;
                }
            }
            block_barrier_raw();  // tile t is in LDS for every wave; everyone is done reading tile t-1's buffer
            char* As = smem + (t % STAGES) * STAGE_BYTES;
            const char* Bs = As + A_BYTES;
            if (ACV == 1) {
                convert_a(As);
                lds_wait<0>();        // this wave's LDS writes have landed (LDS-DMA of later tiles stays in flight: no vmcnt wait)
                sched_fence();
                block_barrier_raw();  // ... and so have everybody else's
            }
            const int arow = wm * WM + (lane & 31), brow = wn * WN + (lane & 31);
            Frag<2> fa[TM], fb[TN];
            auto load_frags = [&](int ks) {
                const int h8 = ks * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < TM; i++) fa[i] = ACV ? frag_presplit(As, arow + i * 32, h8) : frag(As, arow + i * 32, h8);
#pragma unroll
                for (int j = 0; j < TN; j++) fb[j] = BSP ? frag_presplit(Bs, brow + j * 32, h8) : frag(Bs, brow + j * 32, h8);
            };
            load_frags(0);
            sched_fence();
            if (ISSUE) issue(p, ri, kbeg, t + STAGES - 1, smem + ((t + STAGES - 1) % STAGES) * STAGE_BYTES, wave);
            sched_fence();
#pragma unroll
            for (int ks = 0; ks < BK / 16; ks++) {
                // the small cross terms first, into the same accumulator, as mma32<2> of the generic precise kernel
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acc[i][j] = mfma32(fa[i].p[1], fb[j].p[0], acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acc[i][j] = mfma32(fa[i].p[0], fb[j].p[1], acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acc[i][j] = mfma32(fa[i].p[0], fb[j].p[0], acc[i][j]);
                if (ks + 1 < BK / 16) load_frags(ks + 1);
            }
        };
        int t = 0;
        for (; t + STAGES - 1 < nt; t++) step(t, std::true_type{});
        for (; t < nt; t++) step(t, std::false_type{});
        if (CV != 0) {
            Params q = p;
            q.gate = nullptr;  // in conv mode the field carries the zero page, not an activation gate
            avsr_gemm_impl::epilogue_lds<BM, BN, TM, TN, NTHR, 1>(acc, q, m0, n0, wm * WM, wn * WN, zs, 0, smem);
            // the caller sized the statistics buffer for 128-row tiles: a taller tile fills fewer rows, and the rows behind them
            // must read as zero for the fold (block (0, by) clears row gridDim.y + by: every tail row exactly once)
            if (BM > 128 && p.colstat && blockIdx.x == 0) {
                const int r2 = (int)gridDim.y + (int)blockIdx.y;
                if (r2 < p.colstat_rows)
                    for (int i = threadIdx.x; i < 2 * p.N; i += NTHR) p.colstat[(size_t)r2 * 2 * p.N + i] = 0.f;
            }
        } else {
            avsr_gemm_impl::epilogue_lds<BM, BN, TM, TN, NTHR, 1>(acc, p, m0, n0, wm * WM, wn * WN, zs, 0, smem);
        }
    }
};

template <int BM, int BN, int STAGES, int CV, bool BSP, int WGM, int WGN, int ACV>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_split_kernel(Params p) {
    AVSR_DYN_SMEM(smem);
    SplitKernel<BM, BN, STAGES, CV, BSP, WGM, WGN, ACV>::run(p, smem);
}

template <int BM, int BN, int STAGES, int CV, bool BSP, int WGM = 2, int WGN = 2, int ACV = 0>
void launch_split(Params& p, int split_k, hipStream_t stream) {
    using K = SplitKernel<BM, BN, STAGES, CV, BSP, WGM, WGN, ACV>;
    if (CV == 0) {
        int kc = (p.K + split_k - 1) / split_k;
        kc = ((kc + 31) / 32) * 32;
        split_k = (p.K + kc - 1) / kc;
        p.k_chunk = kc;
    } else {
        split_k = 1;
    }
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, split_k), block(K::NTHR);
    AVSR_LAUNCH((gemm_split_kernel<BM, BN, STAGES, CV, BSP, WGM, WGN, ACV>), grid, block, K::LDS_BYTES, stream, p);
}

// tile codes: 1 = 64x64 / 3 stages, 2 = 64x64 / 2 stages, 3 = 128x64 / 2 stages, 4 = 128x128 / 2 stages (48 / 32 / 48 / 64 KiB
// of LDS: three / four / three / two blocks per CU), 5 = 128x64 / 3 stages, 6 = 128x128 / 3 stages
// (7 = 128x64 / 2 stages with a 4 x 1 wave grid: no A fragment is split twice; 11 .. 16 = 1 .. 6 with the A tile converted in
// place once per stage)
template <int CV, bool BSP>
bool launch_tile(int tile, Params& p, int split_k, hipStream_t stream) {
    switch (tile) {
        case 11: launch_split<64, 64, 3, CV, BSP, 2, 2, 1>(p, split_k, stream); return true;
        case 12: launch_split<64, 64, 2, CV, BSP, 2, 2, 1>(p, split_k, stream); return true;
        case 13: launch_split<128, 64, 2, CV, BSP, 2, 2, 1>(p, split_k, stream); return true;
        case 14: launch_split<128, 128, 2, CV, BSP, 2, 2, 1>(p, split_k, stream); return true;
        case 15: launch_split<128, 64, 3, CV, BSP, 2, 2, 1>(p, split_k, stream); return true;
        case 16: launch_split<128, 128, 3, CV, BSP, 2, 2, 1>(p, split_k, stream); return true;
        // 23 .. 26: A pre-split in HBM (ACV = 2): 128x64 / 2, 128x128 / 2, 128x64 / 3, 128x128 / 3 stages
        case 23: if constexpr (BSP) { launch_split<128, 64, 2, CV, BSP, 2, 2, 2>(p, split_k, stream); return true; } return false;
        case 24: if constexpr (BSP) { launch_split<128, 128, 2, CV, BSP, 2, 2, 2>(p, split_k, stream); return true; } return false;
        case 25: if constexpr (BSP) { launch_split<128, 64, 3, CV, BSP, 2, 2, 2>(p, split_k, stream); return true; } return false;
        case 26: if constexpr (BSP) { launch_split<128, 128, 3, CV, BSP, 2, 2, 2>(p, split_k, stream); return true; } return false;
        // 27 .. 30 (round 6): 256-row tiles on 8 waves, A pre-split: 256x64 / 2 stages (80 KB: two blocks per CU), 256x64 / 3,
        // 256x128 / 3 stages (144 KB), 256x128 / 2
        case 27: if constexpr (BSP) { launch_split<256, 64, 2, CV, BSP, 4, 2, 2>(p, split_k, stream); return true; } return false;
        case 28: if constexpr (BSP) { launch_split<256, 64, 3, CV, BSP, 4, 2, 2>(p, split_k, stream); return true; } return false;
        case 29: if constexpr (BSP) { launch_split<256, 128, 3, CV, BSP, 4, 2, 2>(p, split_k, stream); return true; } return false;
        case 30: if constexpr (BSP) { launch_split<256, 128, 2, CV, BSP, 4, 2, 2>(p, split_k, stream); return true; } return false;
        case 1: launch_split<64, 64, 3, CV, BSP>(p, split_k, stream); return true;
        case 2: launch_split<64, 64, 2, CV, BSP>(p, split_k, stream); return true;
        case 3: launch_split<128, 64, 2, CV, BSP>(p, split_k, stream); return true;
        case 4: launch_split<128, 128, 2, CV, BSP>(p, split_k, stream); return true;
        case 5: launch_split<128, 64, 3, CV, BSP>(p, split_k, stream); return true;
        case 6: launch_split<128, 128, 3, CV, BSP>(p, split_k, stream); return true;
        case 7: launch_split<128, 64, 2, CV, BSP, 4, 1>(p, split_k, stream); return true;
        default: return false;
    }
}

// split8 packing: dst group g (16 bf16 = 32 bytes) = {hi(src[8 g .. 8 g + 7]), lo(src[8 g .. 8 g + 7])}; n % 8 == 0
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float v[8];
        load8(src + i * 8, v);
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            bf16_t pl[2];
            split_bf16<2>(v[e], pl);
            hi[e] = (short)pl[0];
            lo[e] = (short)pl[1];
        }
        *reinterpret_cast<bf16x8*>(dst + i * 16) = hi;
        *reinterpret_cast<bf16x8*>(dst + i * 16 + 8) = lo;
    }
}
struct AvsrSplitEntry {
    const float* src;
    bf16_t* dst;
    long n8;    // groups of 8 elements
    long blk0;  // first block of this entry (2048 elements per block)
};
__global__ __launch_bounds__(256) void multi_split_pack_kernel(const AvsrSplitEntry* __restrict__ table, int n) {
    int lo_ = 0, hi_ = n - 1;
    while (lo_ < hi_) {  // last entry with blk0 <= blockIdx.x
        const int mid = (lo_ + hi_ + 1) >> 1;
        if (table[mid].blk0 <= (long)blockIdx.x) lo_ = mid; else hi_ = mid - 1;
    }
    const AvsrSplitEntry e = table[lo_];
    const long i = ((long)blockIdx.x - e.blk0) * 256 + threadIdx.x;
    if (i >= e.n8) return;
    float v[8];
    load8(e.src + i * 8, v);
    bf16x8 hi, lo;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        bf16_t pl[2];
        split_bf16<2>(v[k], pl);
        hi[k] = (short)pl[0];
        lo[k] = (short)pl[1];
    }
    *reinterpret_cast<bf16x8*>(e.dst + i * 16) = hi;
    *reinterpret_cast<bf16x8*>(e.dst + i * 16 + 8) = lo;
}

}  // namespace

extern "C" int avsr_gemm_f32s_nt(const float* A, int lda, const float* B, int ldb, int M, int N, int K, const float* bias,
                                 int act, const void* gate, int gate_dtype, int ldg, float gate_scale, float drop_p,
                                 uint64_t seed, const uint64_t* seed_dev, float alpha, const float* alpha_dev,
                                 const void* resid, int resid_dtype, int ldr, void* C, int c_dtype, int ldc, int accumulate,
                                 int split_k, int tile, float* colsum, int b_split, void* c2, int ldc2, hipStream_t stream) {
    AVSR_REQUIRE(!(colsum && accumulate), "gemm_f32s_nt: colsum needs a non-accumulating output");
    AVSR_REQUIRE(!(c2 && (accumulate || c_dtype != 0)), "gemm_f32s_nt: the bf16 twin needs a non-accumulating f32 output");
    AVSR_REQUIRE(K > 0 && K % 64 == 0, "gemm_f32s_nt: K must be a positive multiple of 64");
    AVSR_REQUIRE(lda % 4 == 0 && ldb % 4 == 0, "gemm_f32s_nt: lda/ldb must be multiples of 4 elements");
    AVSR_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm_f32s_nt: operands must be 16-byte aligned");
    AVSR_REQUIRE(!(accumulate && c_dtype != 0), "gemm_f32s_nt: accumulate needs an f32 output");
    AVSR_REQUIRE(!(split_k > 1 && !accumulate), "gemm_f32s_nt: split-K needs accumulate=1");
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
    p.colsum = colsum;
    p.C2 = c2; p.ldc2 = ldc2;
    p.nsplit = 1; p.batch_h = 1; p.nbatch = 1;
    if (split_k < 1) split_k = 1;
    float* colsum_det = nullptr;  // deterministic mode: as avsr_gemm_bf16_nt
    if (avsr_det()) {
        split_k = 1;
        if (colsum) {
            AVSR_REQUIRE(ldc == N, "gemm_f32s_nt (deterministic mode): colsum needs a dense output");
            colsum_det = colsum;
            p.colsum = nullptr;
        }
    }
    if (tile == 0) {
        const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        const long t12864 = (long)((M + 127) / 128) * ((N + 63) / 64);
        tile = t128 >= 1024 ? 14 : (t12864 >= 400 ? 13 : 11);
    }
    const bool ok = b_split ? launch_tile<0, true>(tile, p, split_k, stream) : launch_tile<0, false>(tile, p, split_k, stream);
    AVSR_REQUIRE(ok, "gemm_f32s_nt: unknown tile code");
    if (colsum_det) avsr_colsum_det(C, c_dtype, ldc, M, N, colsum_det, stream);
    AVSR_CHECK_LAUNCH("gemm_f32s_nt");
    return 0;
}

// f32 implicit-GEMM convolution forward on split hi / lo bf16 planes: x[N,H,W,Cin] * wp[Cout][KH][KW][Cin] -> y[N,OH,OW,Cout]
// (all f32, channels-last; Cin % 64 == 0; zero_page: >= 16 zero bytes in device memory)
static int conv2d_f32s_impl(const float* x, const float* wp, float* y, const void* zero_page, int N, int H, int W, int Cin, int Cout, int KH,
                            int KW, int stride, int pad_h, int pad_w, int tile, int w_split, void* y2, float* stats_part, int stats_tiles,
                            hipStream_t stream) {
    const int OH = (H + 2 * pad_h - KH) / stride + 1, OW = (W + 2 * pad_w - KW) / stride + 1;
    AVSR_REQUIRE(Cin % 64 == 0, "conv2d_f32s: input channel count must be a multiple of 64");
    AVSR_REQUIRE(zero_page != nullptr, "conv2d_f32s: zero page required");
    AVSR_REQUIRE(KH * KW <= 32, "conv2d_f32s: at most 32 filter taps");
    AVSR_REQUIRE((long)N * H * W < (1l << 31) && (long)N * OH * OW < (1l << 31), "conv2d_f32s: pixel count exceeds int32");
    AVSR_REQUIRE(stats_part == nullptr || (long)stats_tiles * 128 >= (long)N * OH * OW, "conv2d_f32s: statistics buffer too small");
    if (N <= 0) return 0;
    Params p{};
    p.A = x; p.B = wp;
    p.K = KH * KW * Cin; p.lda = Cin; p.ldb = p.K;
    p.alpha = 1.f; p.gate_scale = 1.f;
    p.nsplit = 1; p.batch_h = 1; p.nbatch = 1;
    p.gate = zero_page;
    p.c_dtype = 0; p.C = y;
    p.C2 = y2; p.ldc2 = Cout;
    p.cN = N;
    p.cKH = KH; p.cKW = KW; p.cS = stride; p.cPH = pad_h; p.cPW = pad_w; p.cC = Cin;
    p.M = N * OH * OW; p.N = Cout; p.ldc = Cout;
    p.cH = H; p.cW = W; p.cOH = OH; p.cOW = OW;
    p.colstat = stats_part;
    p.colstat_rows = stats_tiles;
    if (tile == 0) tile = Cout >= 128 ? 14 : 13;
    // pre-split A (codes 23 / 24 = "the caller's activation is in the split8 layout, 128-row default"): knobs 21 (Cout < 128) /
    // 22 (Cout >= 128) force a tile for A/B runs
    // round 6: stage-1 geometry (Cout = 64, >= 65 k output rows): 256 x 64 tiles on 8 waves, two blocks per CU: 293 -> 264 us
    // (profiles/r6_microbench_presplit.txt); the 256 x 128 tiles of the 128-channel stage measured slower (200 vs 171 us)
    if (tile == 23 && avsr_tune_knobs[21] == 0 && Cout == 64 && (long)N * OH * OW >= 65536) tile = 27;
    if (tile == 23 && avsr_tune_knobs[21] > 0) tile = avsr_tune_knobs[21];
    if (tile == 24 && avsr_tune_knobs[22] > 0) tile = avsr_tune_knobs[22];
    AVSR_REQUIRE(stats_part == nullptr || tile == 13 || tile == 14 || (tile >= 23 && tile <= 30), "conv2d_f32s: statistics need a 128- or 256-row tile");
    const bool ok = w_split ? launch_tile<1, true>(tile, p, 1, stream) : launch_tile<1, false>(tile, p, 1, stream);
    AVSR_REQUIRE(ok, "conv2d_f32s: unknown tile code");
    AVSR_CHECK_LAUNCH("conv2d_f32s");
    return 0;
}

extern "C" int avsr_conv2d_f32s(const float* x, const float* wp, float* y, const void* zero_page, int N, int H, int W, int Cin,
                                int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int tile, int w_split, void* y2,
                                hipStream_t stream) {
    return conv2d_f32s_impl(x, wp, y, zero_page, N, H, W, Cin, Cout, KH, KW, stride, pad_h, pad_w, tile, w_split, y2, nullptr, 0, stream);
}

// the same convolution leaving the BatchNorm statistics of its output behind: row t of stats_part [stats_tiles >= ceil(rows / 128)]
// [2][Cout] = per-column sums / sums of squares of the stored f32 values of output rows [128 t, 128 t + 128), from the epilogue
// (frontend/resnet.py:82-98: every trunk convolution feeds a BatchNorm in batch-statistics mode); finish with
// avsr_bn_finalize_parts / avsr_bn_stats_parts
extern "C" int avsr_conv2d_f32s_stats(const float* x, const float* wp, float* y, const void* zero_page, int N, int H, int W, int Cin,
                                      int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int tile, int w_split, void* y2,
                                      float* stats_part, int stats_tiles, hipStream_t stream) {
    AVSR_REQUIRE(stats_part != nullptr, "conv2d_f32s_stats: statistics buffer required");
    return conv2d_f32s_impl(x, wp, y, zero_page, N, H, W, Cin, Cout, KH, KW, stride, pad_h, pad_w, tile, w_split, y2, stats_part, stats_tiles, stream);
}

// dst (split8 layout, n * 4 bytes) <- src (f32, n elements, n % 8 == 0): every group of 8 consecutive elements becomes its
// 8 hi bf16 followed by its 8 lo bf16 (prims.h split_bf16).  For a [rows][K] matrix with K % 8 == 0 the row pitch in bytes
// is unchanged, which is what lets avsr_gemm_f32s_nt / avsr_conv2d_f32s address it like the f32 matrix (b_split / w_split = 1).
extern "C" int avsr_split_pack(const float* src, void* dst, int64_t n, hipStream_t stream) {
    AVSR_REQUIRE(n % 8 == 0, "split_pack: element count must be a multiple of 8");
    AVSR_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "split_pack: 16-byte aligned buffers");
    if (n <= 0) return 0;
    const long n8 = n / 8, nb = (n8 + 255) / 256;
    AVSR_LAUNCH(split_pack_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, stream, src, (bf16_t*)dst, n8);
    AVSR_CHECK_LAUNCH("split_pack");
    return 0;
}

// the same for many tensors in ONE launch (the per-step refresh of every pre-split weight): table of 32-byte entries
// {src, dst, n / 8, blk0}, blk0 = running sum of ceil(n / 2048); total_blocks = the final sum
extern "C" int avsr_multi_split_pack(const void* table, int n, int total_blocks, hipStream_t stream) {
    if (n <= 0 || total_blocks <= 0) return 0;
    AVSR_LAUNCH(multi_split_pack_kernel, dim3(total_blocks), dim3(256), 0, stream, reinterpret_cast<const AvsrSplitEntry*>(table), n);
    AVSR_CHECK_LAUNCH("multi_split_pack");
    return 0;
}
