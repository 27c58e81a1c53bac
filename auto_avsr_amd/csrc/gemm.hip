// gemm.hip -- C-ABI entry point of the MFMA GEMM family (kernels in gemm_core.h).
//
// Replaces every torch.nn.Linear / k=1 Conv1d contraction of the hot path and
// their autograd counterparts:
//   FFN w_1/w_2            positionwise_feed_forward.py:24-30
//   linear_q/k/v/out/pos   attention.py:31-34,123
//   pointwise_cov1/2       conformer_encoder.py:24,27
//   proj_encoder           e2e_asr_conformer.py:31
//   ctc_lo / output_layer  ctc.py:21, transformer_decoder.py:225
#include "gemm_core.h"
#include "avsr_hip.h"

namespace avsr_gemm_impl {
int run_nt(const Params&, int, int, int, int, int, hipStream_t);
int run_nn(const Params&, int, int, int, int, int, hipStream_t);
int run_tn(const Params&, int, int, int, int, int, hipStream_t);
int run_tn_multi(const Params*, int, int, int, int, hipStream_t);
}
int avsr_attention_bwd_kv_fast(const void* pd, const void* ds, int lds, const void* dout, int ldo, int64_t sbo, const void* qu,
                               const void* qv, int ldq, int64_t sbq, void* dk, int ldk, int64_t sbk, void* dv, int ldv,
                               int64_t sbv, float* dpos, int ldpos, int B, int H, int Tq, int Tk, hipStream_t stream);
namespace avsr_gemm_impl {
}  // namespace avsr_gemm_impl

extern "C" int avsr_gemm(int layout, const void* A, int a_dtype, int lda, const void* B, int b_dtype,
                         int ldb, int M, int N, int K, int precise, const float* bias, int act,
                         const void* gate, int gate_dtype, int ldg, float gate_scale, float drop_p,
                         uint64_t seed, const uint64_t* seed_dev, float alpha, const float* alpha_dev,
                         const float* resid, int ldr, void* C, int c_dtype, int ldc, int accumulate, int split_k,
                         int force_tile, hipStream_t stream) {
    AVSR_REQUIRE(layout >= 0 && layout <= 2, "gemm: layout must be 0 (NT), 1 (NN) or 2 (TN)");
    AVSR_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "gemm: lda/ldb must be multiples of 8 elements");
    AVSR_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm: operands must be 16-byte aligned");
    AVSR_REQUIRE(!(accumulate && c_dtype != 0), "gemm: accumulate needs an f32 output");
    AVSR_REQUIRE(!(split_k > 1 && !accumulate), "gemm: split-K needs accumulate=1");
    AVSR_REQUIRE(!(precise && (a_dtype != 0 || b_dtype != 0)), "gemm: precise mode needs f32 operands");
    if (M <= 0 || N <= 0) return 0;
    AVSR_REQUIRE(K > 0, "gemm: K must be positive");
    avsr_gemm_impl::Params p{};
    p.A = A; p.B = B; p.lda = lda; p.ldb = ldb;
    p.M = M; p.N = N; p.K = K; p.k_chunk = K;
    p.bias = bias; p.act = act;
    p.gate = gate; p.gate_dtype = gate_dtype; p.ldg = ldg; p.gate_scale = gate_scale;
    p.drop_p = drop_p; p.seed = seed; p.seed_dev = seed_dev; p.alpha_dev = alpha_dev;
    p.alpha = alpha; p.resid = resid; p.ldr = ldr;
    p.C = C; p.c_dtype = c_dtype; p.ldc = ldc; p.accumulate = accumulate;
    p.nsplit = 1; p.batch_h = 1; p.nbatch = 1;
    p.sAb = p.sAh = p.sBb = p.sBh = p.sCb = p.sCh = 0;
    p.a_skew = 0; p.skew_off = 0; p.skew_lim = 0;
    int rc;
    if (avsr_det()) split_k = 1;  // deterministic mode: one block per output element
    if (layout == 0) rc = avsr_gemm_impl::run_nt(p, a_dtype, b_dtype, precise, force_tile, split_k, stream);
    else if (layout == 1) rc = avsr_gemm_impl::run_nn(p, a_dtype, b_dtype, precise, force_tile, split_k, stream);
    else rc = avsr_gemm_impl::run_tn(p, a_dtype, b_dtype, precise, force_tile, split_k, stream);
    AVSR_REQUIRE(rc == 0, "gemm: unsupported dtype combination");
    AVSR_CHECK_LAUNCH("gemm");
    return 0;
}

// Batched TN contraction over (b, h) used by the attention backward:
//   C[b,h] (MxN) (+)= A[b,h]^T-view (KxM, m contiguous) . B[b,h] (KxN, n contiguous)
// with two-level element strides per operand.  a_skew != 0 reads A through the inverse rel_shift index map
// (A[m][k] = src[k*lda + m + k - skew_off]), which turns dS into the gradient of the (q+v) p^T band matrix.
extern "C" int avsr_gemm_tn_batched(const void* A, int a_dtype, int lda, int64_t sAb, int64_t sAh, const void* B,
                                    int b_dtype, int ldb, int64_t sBb, int64_t sBh, void* C, int c_dtype, int ldc,
                                    int64_t sCb, int64_t sCh, int nb, int nh, int M, int N, int K, int precise,
                                    int accumulate, int a_skew, int skew_off, int skew_lim, hipStream_t stream) {
    AVSR_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "gemm_tn_batched: lda/ldb must be multiples of 8 elements");
    AVSR_REQUIRE(!(accumulate && c_dtype != 0), "gemm_tn_batched: accumulate needs an f32 output");
    if (M <= 0 || N <= 0 || nb <= 0 || nh <= 0) return 0;
    AVSR_REQUIRE(K > 0, "gemm_tn_batched: K must be positive");
    avsr_gemm_impl::Params p{};
    p.A = A; p.B = B; p.lda = lda; p.ldb = ldb;
    p.M = M; p.N = N; p.K = K; p.k_chunk = K;
    p.alpha = 1.f; p.gate_scale = 1.f;
    p.C = C; p.c_dtype = c_dtype; p.ldc = ldc; p.accumulate = accumulate;
    p.nsplit = 1; p.batch_h = nh; p.nbatch = nb * nh;
    p.sAb = sAb; p.sAh = sAh; p.sBb = sBb; p.sBh = sBh; p.sCb = sCb; p.sCh = sCh;
    p.a_skew = a_skew; p.skew_off = skew_off; p.skew_lim = skew_lim;
    int rc = avsr_gemm_impl::run_tn(p, a_dtype, b_dtype, precise, 64, 1, stream);
    AVSR_REQUIRE(rc == 0, "gemm_tn_batched: unsupported dtype combination");
    AVSR_CHECK_LAUNCH("gemm_tn_batched");
    return 0;
}

// The key/value side of the attention backward (attention.py:59-104,131-193 under autograd) in ONE launch:
//   dV[b,h] = Pd[b,h]^T dO[b,h]      dK[b,h] = dS[b,h]^T Qu[b,h]      dpos += sum_{b,h} skew(dS[b,h])^T Qv[b,h]
// pd / ds: [B,H,Tq,lds] (avsr_attention_bwd_dq); dout / qu / qv: [B,Tq,H,64] views (row pitch ld*, batch stride sb*);
// dk / dv: [B,Tk,H,64] views; dpos (may be NULL together with qv): f32 [2Tq-1, H*64] with row pitch ldpos (a column block
// of an all-layer buffer), accumulated into (caller zeroes).
// The three contractions are independent and individually far too small for 256 CUs.
extern "C" int avsr_attention_bwd_kv(const void* pd, const void* ds, int lds, const void* dout, int ldo, int64_t sbo,
                                     const void* qu, const void* qv, int ldq, int64_t sbq, void* dk, int ldk, int64_t sbk,
                                     void* dv, int ldv, int64_t sbv, float* dpos, int ldpos, int dtype, int precise, int B,
                                     int H, int Tq, int Tk, int dk_dim, hipStream_t stream) {
    AVSR_REQUIRE(lds % 8 == 0 && ldo % 8 == 0 && ldq % 8 == 0, "attention_bwd_kv: pitches must be multiples of 8 elements");
    AVSR_REQUIRE((dpos == nullptr) == (qv == nullptr), "attention_bwd_kv: dpos and qv go together");
    AVSR_REQUIRE(dpos == nullptr || ldpos >= H * dk_dim, "attention_bwd_kv: dpos row pitch smaller than H * dk");
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return 0;
    // bf16: the k-major tile kernel with LDS transpose reads (attention_kv.hip); avsr_tune knob 10 = 1 selects the generic
    // batched TN path below for A/B runs
    if (dtype == 1 && !precise && dk_dim == 64 && avsr_tune_knobs[10] != 1 && lds % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 &&
        (dpos == nullptr || ldpos % 4 == 0)) {
        avsr_attention_bwd_kv_fast(pd, ds, lds, dout, ldo, sbo, qu, qv, ldq, sbq, dk, ldk, sbk, dv, ldv, sbv, dpos, ldpos, B, H,
                                   Tq, Tk, stream);
        AVSR_CHECK_LAUNCH("attention_bwd_kv");
        return 0;
    }
    avsr_gemm_impl::Params ps[3];
    int n = 0;
    auto base = [&](const void* A, const void* Bm, int ldb, int64_t sBb, void* C, int c_dtype, int ldc, int64_t sCb, int64_t sCh,
                    int M) {
        avsr_gemm_impl::Params p{};
        p.A = A; p.B = Bm; p.lda = lds; p.ldb = ldb;
        p.M = M; p.N = dk_dim; p.K = Tq;
        p.alpha = 1.f; p.gate_scale = 1.f;
        p.C = C; p.c_dtype = c_dtype; p.ldc = ldc;
        p.batch_h = H; p.nbatch = B * H;
        p.sAb = (long)H * Tq * lds; p.sAh = (long)Tq * lds;
        p.sBb = sBb; p.sBh = dk_dim;
        p.sCb = sCb; p.sCh = sCh;
        return p;
    };
    ps[n++] = base(pd, dout, ldo, sbo, dv, dtype, ldv, sbv, dk_dim, Tk);
    ps[n++] = base(ds, qu, ldq, sbq, dk, dtype, ldk, sbk, dk_dim, Tk);
    if (dpos) {
        avsr_gemm_impl::Params p = base(ds, qv, ldq, sbq, dpos, 0, ldpos, 0, dk_dim, 2 * Tq - 1);
        p.accumulate = 1;
        p.a_skew = 1; p.skew_off = Tq - 1; p.skew_lim = Tk;
        ps[n++] = p;
    }
    const int rc = avsr_gemm_impl::run_tn_multi(ps, n, dtype, dtype, precise, stream);
    AVSR_REQUIRE(rc == 0, "attention_bwd_kv: unsupported dtype combination");
    AVSR_CHECK_LAUNCH("attention_bwd_kv");
    return 0;
}
