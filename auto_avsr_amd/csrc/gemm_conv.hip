// gemm_conv.hip -- implicit-GEMM convolutions of the front-ends on channels-last activations, built on the
// MFMA GEMM core (gemm_core.h) with gathering operand loaders; no im2col buffer is ever materialised.
//
// Replaces the ATen/MIOpen convolutions of
//   frontend/resnet.py:10-17,20-35   conv3x3 / 1x1 downsample of the ResNet-18 trunk (and resnet1d.py's k=3 / k=1
//                                     1-D versions, run here as H = 1 images)
//   frontend/resnet.py:204-211       the Conv3d(1, 64, (5,7,7), stride (1,2,2)) stem
//   frontend/resnet1d.py:124-131     the Conv1d(1, 64, 80, stride 4) audio stem (KT = KH = 1)
// and their data / weight gradients.  Weights are consumed in a [Cout][KH][KW][Cin] (forward, weight gradient)
// or [Cin][KH][KW][Cout] (data gradient) permutation of torch's [Cout][Cin][KH][KW], produced by
// avsr_conv_weight_permute in the activation dtype once per step.
#include "gemm_core.h"
#include "avsr_hip.h"

namespace {

using avsr_gemm_impl::Params;

template <int CV, int LA, int LB>
int conv_launch(const Params& p, int a_dtype, int b_dtype, int precise, int split_k, hipStream_t stream) {
    using namespace avsr_gemm_impl;
    if (precise) {
        if (a_dtype != 0 || b_dtype != 0) return -1;
        return launch<float, float, 2, LA, LB, CV>(p, 0, split_k, stream);
    }
    if (a_dtype == 1 && b_dtype == 1) return launch<bf16_t, bf16_t, 1, LA, LB, CV>(p, 0, split_k, stream);
    if (a_dtype == 0 && b_dtype == 1) return launch<float, bf16_t, 1, LA, LB, CV>(p, 0, split_k, stream);
    if (a_dtype == 1 && b_dtype == 0) return launch<bf16_t, float, 1, LA, LB, CV>(p, 0, split_k, stream);
    return launch<float, float, 1, LA, LB, CV>(p, 0, split_k, stream);
}

Params base_params() {
    Params p{};
    p.alpha = 1.f;
    p.gate_scale = 1.f;
    p.nsplit = 1;
    p.batch_h = 1;
    p.nbatch = 1;
    return p;
}

int pick_split(long tiles, long rows) {
    long s = 1024 / (tiles < 1 ? 1 : tiles);
    if (s > rows / 256) s = rows / 256;
    if (s > 256) s = 256;
    return (int)(s < 1 ? 1 : s);
}

// out[a][tap][b] (pitch ld_out over (tap,b)) = w[co][ci][tap]; to_dgrad: a=ci,b=co else a=co,b=ci
template <class TO>
__global__ __launch_bounds__(256) void weight_permute_kernel(const float* __restrict__ w, TO* __restrict__ out, int Cout,
                                                             int Cin, int taps, int to_dgrad, long ld_out) {
    const long total = (long)Cout * Cin * taps;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        // iterate in output order for coalesced stores
        const int A = to_dgrad ? Cin : Cout, Bc = to_dgrad ? Cout : Cin;
        const int b = (int)(i % Bc);
        const int tap = (int)((i / Bc) % taps);
        const int a = (int)(i / ((long)Bc * taps));
        (void)A;
        const int co = to_dgrad ? b : a, ci = to_dgrad ? a : b;
        Elem<TO>::st(out + (long)a * ld_out + (long)tap * Bc + b, w[((long)co * Cin + ci) * taps + tap]);
    }
}
// the same permutation into the split8 layout (gemm_split.hip: every 8 consecutive elements of the dense [a][tap][b] order
// as 8 hi bf16 + 8 lo bf16) -- the pre-split weight operand of the precise-mode forward convolution
__global__ __launch_bounds__(256) void weight_permute_split_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout,
                                                                   int Cin, int taps, int to_dgrad) {
    const long total = (long)Cout * Cin * taps;
    const int Bc = to_dgrad ? Cout : Cin;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i % Bc);
        const int tap = (int)((i / Bc) % taps);
        const int a = (int)(i / ((long)Bc * taps));
        const int co = to_dgrad ? b : a, ci = to_dgrad ? a : b;
        bf16_t pl[2];
        split_bf16<2>(w[((long)co * Cin + ci) * taps + tap], pl);
        bf16_t* blk = out + (i >> 3) * 16 + (i & 7);
        blk[0] = pl[0];
        blk[8] = pl[1];
    }
}
// every conv weight of the model in ONE launch (the per-step refresh of the bf16 [Cout][taps][Cin] / [Cin][taps][Cout]
// copies): table entries {w, out, Cout, Cin, taps, to_dgrad, blk0}, 2048 output elements per block
struct AvsrPermEntry {
    const float* w;
    bf16_t* out;
    int Cout, Cin, taps, to_dgrad, blk0, pad0, pad1, pad2;  // pad0 = 2: `out` is the two-plane IEEE-half image [Cout][2][taps][Cin] (mixed mode forward copies), 3: the split8 layout of the dense order (split-plane forward copies), else bf16
};
// One block = one (co tile, ci tile) x all taps, transposed through LDS: the source w[co][ci][tap] is read in runs of
// TCI * taps consecutive floats per co, the output [a][tap][b] is written in runs of 64 consecutive bf16 (128 bytes).
// forward copy (b = ci): tile 8 co x 64 ci; data-gradient copy (b = co): tile 64 co x 8 ci.  (Round 3: the element-wise
// version gathered the source at a stride of taps -- or Cin * taps -- floats and took 190 us per step for 22 M elements.)
constexpr int PERM_PITCH = 513;  // LDS floats per tap plane (512 + 1: the taps of one (co, ci) pair land on different banks)
__global__ __launch_bounds__(256) void multi_weight_permute_kernel(const AvsrPermEntry* __restrict__ table, int n) {
    AVSR_DYN_SMEM(smem);  // taps * PERM_PITCH floats
    float* lds = reinterpret_cast<float*>(smem);
    int lo = 0, hi = n - 1;
    while (lo < hi) {  // last entry with blk0 <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const AvsrPermEntry e = table[lo];
    const int TCO = e.to_dgrad ? 64 : 8, TCI = e.to_dgrad ? 8 : 64;
    const int nci_t = (e.Cin + TCI - 1) / TCI;
    const int blk = (int)blockIdx.x - e.blk0;
    const int co0 = (blk / nci_t) * TCO, ci0 = (blk % nci_t) * TCI;
    const int seg = TCI * e.taps, total = 512 * e.taps;
    for (int i = threadIdx.x; i < total; i += 256) {  // source order
        const int co_l = i / seg, r = i - co_l * seg;
        const int ci_l = r / e.taps, tap = r - ci_l * e.taps;
        const int co = co0 + co_l, ci = ci0 + ci_l;
        const float v = (co < e.Cout && ci < e.Cin) ? e.w[((long)co * e.Cin + ci) * e.taps + tap] : 0.f;
        lds[tap * PERM_PITCH + (e.to_dgrad ? ci_l * 64 + co_l : co_l * 64 + ci_l)] = v;
    }
    __syncthreads();
    const int Bc = e.to_dgrad ? e.Cout : e.Cin;
    for (int i = threadIdx.x; i < total; i += 256) {  // output order: [a_l][tap][b_l], b_l over 64
        const int b_l = i & 63, t2 = i >> 6;
        const int tap = t2 % e.taps, a_l = t2 / e.taps;
        const int a = (e.to_dgrad ? ci0 : co0) + a_l, b = (e.to_dgrad ? co0 : ci0) + b_l;
        if (a < (e.to_dgrad ? e.Cin : e.Cout) && b < Bc) {
            const float v = lds[tap * PERM_PITCH + a_l * 64 + b_l];
            const long o = ((long)a * e.taps + tap) * Bc + b;
            if (e.pad0 == 2) {  // two-plane f16 image [a][2][taps][b] (prims.h f2h_lo)
                f16_t* o16 = reinterpret_cast<f16_t*>(e.out) + ((long)(2 * a) * e.taps + tap) * Bc + b;
                o16[0] = f2h(v);
                o16[(long)e.taps * Bc] = f2h_lo(v);
            } else if (e.pad0 == 3) {  // split8 layout (gemm_split.hip): every 8 consecutive elements of the dense order as 8 hi + 8 lo bf16
                bf16_t pl[2];
                split_bf16<2>(v, pl);
                bf16_t* blk8 = e.out + (o >> 3) * 16 + (o & 7);
                blk8[0] = pl[0];
                blk8[8] = pl[1];
            } else e.out[o] = f2bf(v);
        }
    }
}
// dw[co][ci][tap] = dwp[co][tap][ci]
__global__ __launch_bounds__(256) void weight_unpermute_kernel(const float* __restrict__ dwp, float* __restrict__ dw,
                                                               int Cout, int Cin, int taps) {
    const long total = (long)Cout * Cin * taps;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int tap = (int)(i % taps);
        const int ci = (int)((i / taps) % Cin);
        const int co = (int)(i / ((long)taps * Cin));
        dw[i] = dwp[((long)co * taps + tap) * Cin + ci];
    }
}

}  // namespace

extern "C" int avsr_conv_weight_permute(const float* w, void* out, int out_dtype, int Cout, int Cin, int taps,
                                        int to_dgrad, int64_t ld_out, hipStream_t stream) {
    const long total = (long)Cout * Cin * taps;
    if (total <= 0) return 0;
    long nb = (total + 255) / 256;
    dim3 grid((unsigned)(nb > 2048 ? 2048 : nb)), block(256);
    if (out_dtype == 2) {
        AVSR_REQUIRE(total % 8 == 0 && ld_out == (int64_t)taps * (to_dgrad ? Cout : Cin), "conv_weight_permute: split8 output must be dense");
        AVSR_LAUNCH(weight_permute_split_kernel, grid, block, 0, stream, w, (bf16_t*)out, Cout, Cin, taps, to_dgrad);
    } else if (out_dtype == 3)  // IEEE half (dtype code 2 is taken by the split8 layout here)
        AVSR_LAUNCH((weight_permute_kernel<f16_t>), grid, block, 0, stream, w, (f16_t*)out, Cout, Cin, taps, to_dgrad, (long)ld_out);
    else if (out_dtype == 0)
        AVSR_LAUNCH((weight_permute_kernel<float>), grid, block, 0, stream, w, (float*)out, Cout, Cin, taps, to_dgrad, (long)ld_out);
    else
        AVSR_LAUNCH((weight_permute_kernel<bf16_t>), grid, block, 0, stream, w, (bf16_t*)out, Cout, Cin, taps, to_dgrad, (long)ld_out);
    AVSR_CHECK_LAUNCH("conv_weight_permute");
    return 0;
}

// table: n entries of 48 bytes {w, out, Cout, Cin, taps, to_dgrad, blk0, 0, 0, 0}; blk0 = running sum of
// avsr_weight_permute_blocks(Cout, Cin, to_dgrad); total_blocks = the final sum; max_taps = the largest `taps` of the table.
// Outputs are dense bf16 [a][taps][b].
extern "C" int64_t avsr_weight_permute_blocks(int Cout, int Cin, int to_dgrad) {
    return to_dgrad ? ((Cout + 63) / 64) * ((Cin + 7) / 8) : ((Cout + 7) / 8) * ((Cin + 63) / 64);
}
extern "C" int avsr_multi_weight_permute(const void* table, int n, int total_blocks, int max_taps, hipStream_t stream) {
    if (n <= 0 || total_blocks <= 0) return 0;
    AVSR_REQUIRE(max_taps >= 1 && max_taps <= 64, "multi_weight_permute: at most 64 taps per filter");
    AVSR_LAUNCH(multi_weight_permute_kernel, dim3(total_blocks), dim3(256), (size_t)max_taps * PERM_PITCH * sizeof(float), stream,
                reinterpret_cast<const AvsrPermEntry*>(table), n);
    AVSR_CHECK_LAUNCH("multi_weight_permute");
    return 0;
}

extern "C" int avsr_conv_weight_unpermute(const float* dwp, float* dw, int Cout, int Cin, int taps, hipStream_t stream) {
    const long total = (long)Cout * Cin * taps;
    if (total <= 0) return 0;
    long nb = (total + 255) / 256;
    AVSR_LAUNCH(weight_unpermute_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, stream, dwp, dw, Cout, Cin, taps);
    AVSR_CHECK_LAUNCH("conv_weight_unpermute");
    return 0;
}

static void fill_conv(Params& p, int H, int W, int C, int OH, int OW, int KH, int KW, int stride, int ph, int pw) {
    p.cH = H; p.cW = W; p.cC = C; p.cOH = OH; p.cOW = OW;
    p.cKH = KH; p.cKW = KW; p.cS = stride; p.cPH = ph; p.cPW = pw;
    p.cT = 1; p.cKT = 1; p.cPT = 0;
}

// y[N,OH,OW,Cout] = conv(x[N,H,W,Cin], wp[Cout][KH][KW][Cin])
extern "C" int avsr_conv2d_fwd(const void* x, int dtype, const void* wp, int w_dtype, void* y, int N, int H, int W,
                               int Cin, int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int precise,
                               hipStream_t stream) {
    AVSR_REQUIRE(Cin % 8 == 0, "conv2d: Cin must be a multiple of 8");
    const int OH = (H + 2 * pad_h - KH) / stride + 1, OW = (W + 2 * pad_w - KW) / stride + 1;
    if (N <= 0) return 0;
    Params p = base_params();
    p.A = x; p.B = wp; p.lda = Cin; p.ldb = KH * KW * Cin;
    p.M = N * OH * OW; p.N = Cout; p.K = KH * KW * Cin; p.k_chunk = p.K;
    p.C = y; p.c_dtype = dtype; p.ldc = Cout;
    fill_conv(p, H, W, Cin, OH, OW, KH, KW, stride, pad_h, pad_w);
    AVSR_REQUIRE((conv_launch<1, 0, 0>(p, dtype, w_dtype, precise, 1, stream)) == 0, "conv2d_fwd: dtype combination");
    AVSR_CHECK_LAUNCH("conv2d_fwd");
    return 0;
}

// dx[N,H,W,Cin] = conv_transpose(dy[N,OH,OW,Cout], wpd[Cin][KH][KW][Cout]) (+ resid[N,H,W,Cin], same dtype)
extern "C" int avsr_conv2d_dgrad(const void* dy, int dtype, const void* wpd, int w_dtype, const void* resid, void* dx,
                                 int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_h, int pad_w,
                                 int precise, hipStream_t stream) {
    AVSR_REQUIRE(Cout % 8 == 0, "conv2d: Cout must be a multiple of 8");
    const int OH = (H + 2 * pad_h - KH) / stride + 1, OW = (W + 2 * pad_w - KW) / stride + 1;
    if (N <= 0) return 0;
    Params p = base_params();
    p.A = dy; p.B = wpd; p.lda = Cout; p.ldb = KH * KW * Cout;
    p.M = N * H * W; p.N = Cin; p.K = KH * KW * Cout; p.k_chunk = p.K;
    p.C = dx; p.c_dtype = dtype; p.ldc = Cin;
    p.resid = reinterpret_cast<const float*>(resid); p.ldr = Cin; p.resid_dtype = dtype;
    // the gathered tensor is dy (OH x OW x Cout); rows run over the input pixel grid (H x W)
    fill_conv(p, OH, OW, Cout, H, W, KH, KW, stride, pad_h, pad_w);
    AVSR_REQUIRE((conv_launch<2, 0, 0>(p, dtype, w_dtype, precise, 1, stream)) == 0, "conv2d_dgrad: dtype combination");
    AVSR_CHECK_LAUNCH("conv2d_dgrad");
    return 0;
}

// dwp[Cout][KH][KW][Cin] (f32, zero-initialised by the caller) += dy^T im2col(x)
extern "C" int avsr_conv2d_wgrad(const void* dy, const void* x, int dtype, float* dwp, int N, int H, int W, int Cin,
                                 int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int precise,
                                 hipStream_t stream) {
    AVSR_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, "conv2d: channels must be multiples of 8");
    const int OH = (H + 2 * pad_h - KH) / stride + 1, OW = (W + 2 * pad_w - KW) / stride + 1;
    if (N <= 0) return 0;
    Params p = base_params();
    p.A = dy; p.B = x; p.lda = Cout; p.ldb = Cin;
    p.M = Cout; p.N = KH * KW * Cin; p.K = N * OH * OW; p.k_chunk = p.K;
    p.C = dwp; p.c_dtype = 0; p.ldc = p.N; p.accumulate = 1;
    fill_conv(p, H, W, Cin, OH, OW, KH, KW, stride, pad_h, pad_w);
    const long tiles = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
    AVSR_REQUIRE((conv_launch<3, 1, 1>(p, dtype, dtype, precise, pick_split(tiles, p.K), stream)) == 0,
                 "conv2d_wgrad: dtype combination");
    AVSR_CHECK_LAUNCH("conv2d_wgrad");
    return 0;
}

static void fill_stem(Params& p, int T, int H, int W, int OH, int OW, int KT, int KH, int KW, int stride, int pt, int ph,
                      int pw) {
    p.cH = H; p.cW = W; p.cC = 1; p.cOH = OH; p.cOW = OW;
    p.cKH = KH; p.cKW = KW; p.cS = stride; p.cPH = ph; p.cPW = pw;
    p.cT = T; p.cKT = KT; p.cPT = pt;
}

// y[B,T,OH,OW,Cout] = conv3d(x[B,T,H,W] (one channel, f32), wp[Cout][ldw >= KT*KH*KW]), temporal stride 1
extern "C" int avsr_conv_stem_fwd(const float* x, const void* wp, int w_dtype, int ldw, void* y, int y_dtype, int B, int T,
                                  int H, int W, int Cout, int KT, int KH, int KW, int stride, int pad_t, int pad_h,
                                  int pad_w, int precise, hipStream_t stream) {
    AVSR_REQUIRE(ldw % 8 == 0 && ldw >= KT * KH * KW, "conv_stem: weight pitch must be a multiple of 8");
    const int OH = (H + 2 * pad_h - KH) / stride + 1, OW = (W + 2 * pad_w - KW) / stride + 1;
    if (B <= 0 || T <= 0) return 0;
    Params p = base_params();
    p.A = x; p.B = wp; p.lda = 8; p.ldb = ldw;
    p.M = B * T * OH * OW; p.N = Cout; p.K = KT * KH * KW; p.k_chunk = p.K;
    p.C = y; p.c_dtype = y_dtype; p.ldc = Cout;
    fill_stem(p, T, H, W, OH, OW, KT, KH, KW, stride, pad_t, pad_h, pad_w);
    AVSR_REQUIRE((conv_launch<4, 0, 0>(p, 0, w_dtype, precise, 1, stream)) == 0, "conv_stem_fwd: dtype combination");
    AVSR_CHECK_LAUNCH("conv_stem_fwd");
    return 0;
}

// dw[Cout][KT*KH*KW] (f32, zero-initialised, torch layout since Cin = 1) += dy^T im2col(x)
extern "C" int avsr_conv_stem_wgrad(const void* dy, int dy_dtype, const float* x, float* dw, int B, int T, int H, int W,
                                    int Cout, int KT, int KH, int KW, int stride, int pad_t, int pad_h, int pad_w,
                                    int precise, hipStream_t stream) {
    AVSR_REQUIRE(Cout % 8 == 0, "conv_stem: Cout must be a multiple of 8");
    const int OH = (H + 2 * pad_h - KH) / stride + 1, OW = (W + 2 * pad_w - KW) / stride + 1;
    if (B <= 0 || T <= 0) return 0;
    Params p = base_params();
    p.A = dy; p.B = x; p.lda = Cout; p.ldb = 8;
    p.M = Cout; p.N = KT * KH * KW; p.K = B * T * OH * OW; p.k_chunk = p.K;
    p.C = dw; p.c_dtype = 0; p.ldc = p.N; p.accumulate = 1;
    fill_stem(p, T, H, W, OH, OW, KT, KH, KW, stride, pad_t, pad_h, pad_w);
    const long tiles = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
    AVSR_REQUIRE((conv_launch<5, 1, 1>(p, dy_dtype, 0, precise, pick_split(tiles, p.K), stream)) == 0,
                 "conv_stem_wgrad: dtype combination");
    AVSR_CHECK_LAUNCH("conv_stem_wgrad");
    return 0;
}
