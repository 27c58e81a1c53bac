// gemm_split.hip -- the tuned path of the PRECISE numerical mode's forward contractions:
//     C[M,N] = epi( A[M,K] . B[N,K]^T )      A, B f32 in HBM, k-contiguous, K % 64 == 0
// with every product formed on split hi + lo bf16 planes (3 x v_mfma_f32_32x32x16_bf16 per product, ~2^-16 relative
// error) -- the arithmetic of gemm_core.h's NS = 2 kernel, on the operand path of gemm_fast.hip:
//  * the f32 tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4) into a STAGES-deep ring, counted s_waitcnt vmcnt in
//    front of a raw s_barrier; no VGPR staging, no ds_write pass;
//  * LDS image per operand stage: [rows][64] f32 = 256-byte rows with the 16-byte chunk index XOR-swizzled by (row & 15) --
//    applied to the per-lane SOURCE address on the way in and to the ds_read_b128 address on the way out;
//  * the hi / lo split happens in registers on the MFMA fragments (8 consecutive k of one row = two ds_read_b128):
//    hi = bf16(x), lo = bf16(x - hi), exactly split_bf16<2>() of prims.h, so the result is bit-compatible with the generic
//    precise kernel up to summation order;
//  * 2 x 2 waves, each a (BM/2) x (BN/2) grid of 32x32 accumulators; epilogue through LDS (gemm_core.h epilogue_lds).
// CV = 1: A is the im2col view of a channels-last f32 image (convolution forward, channels % 64 == 0), gathered like
// gemm_fast_kernel.h does: one pointer + tap-validity mask per staged row, a zero page for the padding taps.
//
// Replaces, in the precise / hpf modes: every nn.Linear / k=1 Conv1d forward of the encoder, decoder and heads
// (positionwise_feed_forward.py:24-30, attention.py:31-34,123, conformer_encoder.py:24,27, e2e_asr_conformer.py:31, ctc.py:21,
// transformer_decoder.py:225) and the ResNet trunk's forward convolutions (frontend/resnet.py:10-35).
#include "gemm_core.h"
#include "avsr_hip.h"

namespace {

using avsr_gemm_impl::Params;

template <int BM, int BN, int STAGES, int CV>
struct SplitKernel {
    static constexpr int BK = 64, NW = 4, NTHR = 256;
    static constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    static constexpr int A_BYTES = BM * 256, B_BYTES = BN * 256, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int A_LOADS = BM / 16, B_LOADS = BN / 16;  // wave-instructions (4 rows each) per wave per stage
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
        const int rsub = lane >> 4, pc = lane & 15;
#pragma unroll
        for (int i = 0; i < A_LOADS; i++) {
            const int r = (wave * A_LOADS + i) * 4 + rsub;  // row inside the tile
            const int c = pc ^ (r & 15);                    // source chunk that lands in physical chunk pc
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
            const int r = (wave * B_LOADS + i) * 4 + rsub;
            const int c = pc ^ (r & 15);
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
        const int sw = r & 15;
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + r * 256 + (((2 * half8) ^ sw) << 4));
        const f32x4 b = *reinterpret_cast<const f32x4*>(base + r * 256 + (((2 * half8 + 1) ^ sw) << 4));
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

    static AVSR_DEV void run(const Params& p, char* smem) {
        const float* A = reinterpret_cast<const float*>(p.A);
        const float* B = reinterpret_cast<const float*>(p.B);
        const int lane = threadIdx.x & 63, wave = wave_id();
        const int wm = wave >> 1, wn = wave & 1;
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
            const char* As = smem + (t % STAGES) * STAGE_BYTES;
            const char* Bs = As + A_BYTES;
            const int arow = wm * WM + (lane & 31), brow = wn * WN + (lane & 31);
            Frag<2> fa[TM], fb[TN];
            auto load_frags = [&](int ks) {
                const int h8 = ks * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < TM; i++) fa[i] = frag(As, arow + i * 32, h8);
#pragma unroll
                for (int j = 0; j < TN; j++) fb[j] = frag(Bs, brow + j * 32, h8);
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
        } else {
            avsr_gemm_impl::epilogue_lds<BM, BN, TM, TN, NTHR, 1>(acc, p, m0, n0, wm * WM, wn * WN, zs, 0, smem);
        }
    }
};

template <int BM, int BN, int STAGES, int CV>
__global__ __launch_bounds__(256) void gemm_split_kernel(Params p) {
    AVSR_DYN_SMEM(smem);
    SplitKernel<BM, BN, STAGES, CV>::run(p, smem);
}

template <int BM, int BN, int STAGES, int CV>
void launch_split(Params& p, int split_k, hipStream_t stream) {
    using K = SplitKernel<BM, BN, STAGES, CV>;
    if (CV == 0) {
        int kc = (p.K + split_k - 1) / split_k;
        kc = ((kc + 63) / 64) * 64;
        split_k = (p.K + kc - 1) / kc;
        p.k_chunk = kc;
    } else {
        split_k = 1;
    }
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, split_k), block(K::NTHR);
    AVSR_LAUNCH((gemm_split_kernel<BM, BN, STAGES, CV>), grid, block, K::LDS_BYTES, stream, p);
}

// tile codes: 1 = 64x64 / 3 stages (96 KiB: one block per CU -- the skinny M = B*T GEMMs give every CU one tile anyway),
// 2 = 64x64 / 2 stages (two blocks per CU), 3 = 128x64 / 2 stages, 4 = 128x128 / 2 stages
template <int CV>
bool launch_tile(int tile, Params& p, int split_k, hipStream_t stream) {
    switch (tile) {
        case 1: launch_split<64, 64, 3, CV>(p, split_k, stream); return true;
        case 2: launch_split<64, 64, 2, CV>(p, split_k, stream); return true;
        case 3: launch_split<128, 64, 2, CV>(p, split_k, stream); return true;
        case 4: launch_split<128, 128, 2, CV>(p, split_k, stream); return true;
        default: return false;
    }
}

}  // namespace

extern "C" int avsr_gemm_f32s_nt(const float* A, int lda, const float* B, int ldb, int M, int N, int K, const float* bias,
                                 int act, const void* gate, int gate_dtype, int ldg, float gate_scale, float drop_p,
                                 uint64_t seed, const uint64_t* seed_dev, float alpha, const float* alpha_dev,
                                 const void* resid, int resid_dtype, int ldr, void* C, int c_dtype, int ldc, int accumulate,
                                 int split_k, int tile, float* colsum, hipStream_t stream) {
    AVSR_REQUIRE(!(colsum && accumulate), "gemm_f32s_nt: colsum needs a non-accumulating output");
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
    p.nsplit = 1; p.batch_h = 1; p.nbatch = 1;
    if (split_k < 1) split_k = 1;
    if (tile == 0) {
        const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        const long t12864 = (long)((M + 127) / 128) * ((N + 63) / 64);
        tile = t128 >= 512 ? 4 : (t12864 >= 400 ? 3 : 1);
    }
    AVSR_REQUIRE(launch_tile<0>(tile, p, split_k, stream), "gemm_f32s_nt: unknown tile code");
    AVSR_CHECK_LAUNCH("gemm_f32s_nt");
    return 0;
}

// f32 implicit-GEMM convolution forward on split hi / lo bf16 planes: x[N,H,W,Cin] * wp[Cout][KH][KW][Cin] -> y[N,OH,OW,Cout]
// (all f32, channels-last; Cin % 64 == 0; zero_page: >= 16 zero bytes in device memory)
extern "C" int avsr_conv2d_f32s(const float* x, const float* wp, float* y, const void* zero_page, int N, int H, int W, int Cin,
                                int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int tile, hipStream_t stream) {
    const int OH = (H + 2 * pad_h - KH) / stride + 1, OW = (W + 2 * pad_w - KW) / stride + 1;
    AVSR_REQUIRE(Cin % 64 == 0, "conv2d_f32s: input channel count must be a multiple of 64");
    AVSR_REQUIRE(zero_page != nullptr, "conv2d_f32s: zero page required");
    AVSR_REQUIRE(KH * KW <= 32, "conv2d_f32s: at most 32 filter taps");
    AVSR_REQUIRE((long)N * H * W < (1l << 31) && (long)N * OH * OW < (1l << 31), "conv2d_f32s: pixel count exceeds int32");
    if (N <= 0) return 0;
    Params p{};
    p.A = x; p.B = wp;
    p.K = KH * KW * Cin; p.lda = Cin; p.ldb = p.K;
    p.alpha = 1.f; p.gate_scale = 1.f;
    p.nsplit = 1; p.batch_h = 1; p.nbatch = 1;
    p.gate = zero_page;
    p.c_dtype = 0; p.C = y;
    p.cN = N;
    p.cKH = KH; p.cKW = KW; p.cS = stride; p.cPH = pad_h; p.cPW = pad_w; p.cC = Cin;
    p.M = N * OH * OW; p.N = Cout; p.ldc = Cout;
    p.cH = H; p.cW = W; p.cOH = OH; p.cOW = OW;
    if (tile == 0) tile = Cout >= 128 ? 4 : 3;
    AVSR_REQUIRE(launch_tile<1>(tile, p, 1, stream), "conv2d_f32s: unknown tile code");
    AVSR_CHECK_LAUNCH("conv2d_f32s");
    return 0;
}
