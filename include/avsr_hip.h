/* avsr_hip.h -- C ABI of libavsr_hip.so: the MI355X (gfx950) kernels behind the
 * auto_avsr training hot path  E2E.forward / backward
 * (reference: espnet/nets/pytorch_backend/e2e_asr_conformer.py:63-87).
 *
 * The reference has no FFI of its own (it is 100 % Python on ATen); every entry
 * point below replaces the implicit ATen/cuDNN/cuBLAS op sequence of the cited
 * reference lines.  Conventions:
 *   - plain pointers + sizes, caller-owned device buffers, no allocation, no
 *     synchronisation; work is enqueued on `stream`;
 *   - return 0 on success, non-zero on error (avsr_last_error() has the text);
 *   - dtype codes: 0 = float32, 1 = bfloat16 (raw 16-bit), 2 = IEEE float16 (forward activations of the mixed mode only);
 *   - row-major tensors; "ld" arguments are leading dimensions in elements.
 */
#ifndef AVSR_HIP_H
#define AVSR_HIP_H
#include <stdint.h>

#if defined(__HIP__) || defined(AVSR_EMU)
typedef hipStream_t avsr_stream_t;
#else
typedef void* avsr_stream_t; /* a hipStream_t */
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library / status ------------------------------------------------------------------ */
int avsr_abi_version(void);
int avsr_is_emulator(void);
const char* avsr_last_error(void);

/* ---- LayerNorm (layer_norm.py:12-33, eps 1e-12) ------------------------------------------ */
/* y = (x-mean)*rstd*gamma+beta ; x f32 [rows,cols]; y dtype selectable; saves mean/rstd [rows] */
int avsr_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype,
                       float* mean, float* rstd, int rows, int cols, float eps,
                       avsr_stream_t stream);
/* the same with an f32 result AND its bf16 twin y2 in one pass ("hpf" numerical mode: f32 forward, bf16 copies for the backward) */
int avsr_layernorm_fwd2(const float* x, const float* gamma, const float* beta, float* y, void* y2, float* mean, float* rstd,
                        int rows, int cols, float eps, avsr_stream_t stream);
/* dx = LN'(dy) (+ dres if non-null); dgamma/dbeta are accumulated into.
 * gout (bf16 [rows][cols], may be NULL) = bf16(alpha * dropout(dx)) and gsum (f32 [cols], may be NULL; accumulated into)
 * += column sums of gout: the backward prologue of the Linear whose output gradient dx is -- the output projection of the
 * PREVIOUS pre-LN sub-layer (conformer_encoder.py:110-159: x + ff_scale * dropout(ff(LN(x))) etc.), fused here so that
 * no separate cast / bias-gradient pass re-reads dx.  Dropout stream = avsr_cast_transpose_colsum's. */
int avsr_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma,
                       const float* mean, const float* rstd, const float* dres, float* dx,
                       float* dgamma, float* dbeta, void* gout, float* gsum, float alpha, float drop_p, uint64_t seed,
                       const uint64_t* seed_dev, int rows, int cols, avsr_stream_t stream);

/* ---- MFMA GEMM family (Linear / pointwise conv and their gradients) ---------------------- */
/* C[M,N] = epi(sum_k A[m,k]*B[n,k]);  layout 0 "NT": A [M][K], B [N][K] (Linear forward, B = weight)
 *                                     layout 1 "NN": A [M][K], B [K][N] (data gradient, B = weight)
 *                                     layout 2 "TN": A [K][M], B [K][N] (weight gradient: A = dY, B = x)
 * precise=1: f32 operands split into hi+lo bf16 planes (3 MFMAs per product, ~1e-5 rel error).
 * epilogue order: +bias[n] -> act (0 none,1 relu,2 silu) -> gate (v = gate[m,n]>0 ? v*gate_scale : 0)
 *   -> dropout(drop_p, seed) -> *alpha -> +resid[m,n] -> store (c_dtype) or atomicAdd (accumulate=1, f32).
 * seed_dev / alpha_dev (may be NULL): device-resident scalars added to seed / multiplied into alpha, so that a
 * captured hipGraph replays with fresh dropout masks and live upstream gradients.
 * split_k>1 requires accumulate=1.  force_tile: 0 auto, 64 or 128. */
int avsr_gemm(int layout, const void* A, int a_dtype, int lda, const void* B, int b_dtype, int ldb,
              int M, int N, int K, int precise, const float* bias, int act, const void* gate,
              int gate_dtype, int ldg, float gate_scale, float drop_p, uint64_t seed, const uint64_t* seed_dev,
              float alpha, const float* alpha_dev, const float* resid, int ldr, void* C, int c_dtype, int ldc,
              int accumulate, int split_k, int force_tile, avsr_stream_t stream);

/* ---- fused multi-head attention (attention.py:59-104,131-193) ----------------------------- */
/* out[b,i,h,:] = softmax_j( scale*(qu_i.k_j + [pos!=NULL] qv_i.pos[j-i+Tq-1]) , mask ) @ v ; d_k = 64.
 * qu/qv/k/v/out are [B,T,H,64] views: element (b,t,h,d) at b*sb + t*ld + h*64 + d.  pos: [2Tq-1, H*64]
 * (ldp).  mask[b*mask_sb + i*mask_sq + j] != 0 -> attend (NULL: none); fully masked rows give zeros
 * (attention.py:71-77).  drop_p: dropout on the probabilities (attention.py:79).  lse: [B,H,Tq]. */
int avsr_attention_fwd(const void* qu, const void* qv, const void* k, const void* v, const void* pos,
                       int dtype, int precise, const uint8_t* mask, int64_t mask_sb, int64_t mask_sq,
                       void* out, float* lse, int B, int H, int Tq, int Tk, int dk, int ldq, int ldk,
                       int ldv, int ldp, int ldo, int64_t sbq, int64_t sbk, int64_t sbv, int64_t sbo,
                       float scale, float drop_p, uint64_t seed, const uint64_t* seed_dev, avsr_stream_t stream);
/* the precise (f32) forward + the bf16 twin of its output (same strides as out) in one pass: the "hpf" numerical mode */
int avsr_attention_fwd2(const void* qu, const void* qv, const void* k, const void* v, const void* pos, const uint8_t* mask,
                        int64_t mask_sb, int64_t mask_sq, void* out, void* out2, float* lse, int B, int H, int Tq, int Tk, int dk,
                        int ldq, int ldk, int ldv, int ldp, int ldo, int64_t sbq, int64_t sbk, int64_t sbv, int64_t sbo, float scale,
                        float drop_p, uint64_t seed, const uint64_t* seed_dev, avsr_stream_t stream);
/* backward, query side: recomputes P from lse; writes dqu (and dqv), plus pd = dropout(P) and
 * ds = scale*dS as [B,H,Tq,lds] tensors from which dK, dV and dpos follow as batched TN GEMMs. */
int avsr_attention_bwd_dq(const void* qu, const void* qv, const void* k, const void* v, const void* pos,
                          int dtype, int precise, const uint8_t* mask, int64_t mask_sb, int64_t mask_sq,
                          const void* out, const float* lse, const void* dout, void* dqu, void* dqv,
                          void* pd, void* ds, int lds, int B, int H, int Tq, int Tk, int dk, int ldq,
                          int ldk, int ldv, int ldp, int ldo, int64_t sbq, int64_t sbk, int64_t sbv,
                          int64_t sbo, float scale, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                          void* dq_sum /* may be NULL; else (relative-position form): dq_sum = dqu + dqv is written as a
                          [B,Tq,H,64] view (row pitch lddq, batch stride sbdq) INSTEAD of dqu / dqv, and the column sums of
                          the two parts are added to du / dv [H*64] (f32, zeroed by the caller): the gradients of q, pos_bias_u
                          and pos_bias_v of attention.py:171-177 without a separate pass */,
                          int lddq, int64_t sbdq, float* du, float* dv, avsr_stream_t stream);

/* key/value side of the attention backward in ONE launch (three independent batched TN contractions):
 * dV = Pd^T dO, dK = dS^T Qu, dpos += skew(dS)^T Qv (dpos / qv may both be NULL); pd/ds from avsr_attention_bwd_dq;
 * dout/qu/qv/dk/dv are [B,T,H,64] views (row pitch ld*, batch stride sb*); dpos f32 [2Tq-1, H*64] with row pitch ldpos, caller zeroes */
int avsr_attention_bwd_kv(const void* pd, const void* ds, int lds, const void* dout, int ldo, int64_t sbo,
                          const void* qu, const void* qv, int ldq, int64_t sbq, void* dk, int ldk, int64_t sbk, void* dv,
                          int ldv, int64_t sbv, float* dpos, int ldpos, int dtype, int precise, int B, int H, int Tq,
                          int Tk, int dk_dim, avsr_stream_t stream);
/* batched TN contraction over (b,h) for the attention backward (dV = Pd^T dO, dK = dS^T Qu, dpos = skew(dS)^T Qv):
 * C[b,h][M,N] (+)= sum_k A[b,h][k,m] * B[b,h][k,n]; operand (b,h) slices start at b*s?b + h*s?h elements.
 * a_skew: A[m][k] = src[k*lda + m + k - skew_off], valid iff that column lies in [0, skew_lim)  (inverse of
 * rel_shift, attention.py:131-151). */
int avsr_gemm_tn_batched(const void* A, int a_dtype, int lda, int64_t sAb, int64_t sAh, const void* B,
                         int b_dtype, int ldb, int64_t sBb, int64_t sBh, void* C, int c_dtype, int ldc,
                         int64_t sCb, int64_t sCh, int nb, int nh, int M, int N, int K, int precise,
                         int accumulate, int a_skew, int skew_off, int skew_lim, avsr_stream_t stream);

/* ---- elementwise glue (elementwise.hip) --------------------------------------------------- */
/* out = dropout(alpha*x + add[i % add_period])   (add may be NULL; embedding.py:78-87,179-184; the dropout
 * branches of conformer_encoder.py:114-157 on the backward side; ctc.py:54).  x_dtype 3 (round 5): x is an f32-sized activation
 * stored in the split8 layout (n % 8 == 0) -- the cast at the boundary between a split-plane and an f16 component */
int avsr_scale_dropout(const void* x, int x_dtype, void* out, int out_dtype, int64_t n, float alpha,
                       const float* alpha_dev, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                       const float* add, int64_t add_period, avsr_stream_t stream);
/* o1 = x + b1[col], o2 = x + b2[col]  (q + pos_bias_u / q + pos_bias_v, attention.py:176-178); o1,o2 dense */
int avsr_head_bias_fwd(const void* x, int dtype, int64_t ldx, const float* b1, const float* b2, void* o1,
                       void* o2, int64_t rows, int cols, avsr_stream_t stream);
/* f32 outputs + their bf16 twins (t1, t2: both or none) */
int avsr_head_bias_fwd2(const float* x, int64_t ldx, const float* b1, const float* b2, float* o1, float* o2, void* t1, void* t2,
                        int64_t rows, int cols, avsr_stream_t stream);
/* dq = d1 + d2 (row stride ldo; d2/dq may be NULL); db1 += colsum(d1); db2 += colsum(d2) (NULL skips) */
int avsr_head_bias_bwd(const void* d1, const void* d2, int dtype, void* dq, int64_t ldo, float* db1,
                       float* db2, int64_t rows, int cols, avsr_stream_t stream);
/* GLU over channel halves of a [rows, 2C] tensor (conformer_encoder.py:32) */
int avsr_glu_fwd(const void* a, void* g, int dtype, int64_t rows, int C, avsr_stream_t stream);
int avsr_glu_bwd(const void* a, const void* dg, void* da, int dtype, int64_t rows, int C,
                 avsr_stream_t stream);

/* ---- depthwise conv over time on (B,T,C) (conformer_encoder.py:25,33) ------------------------ */
/* y[b,t,c] = bias[c] + sum_k w[c,k] x[b,t+k-(K-1)/2,c]; flip=1 reverses taps (= data gradient, bias NULL).
 * The GLU in front of the convolution (conformer_encoder.py:32 `nn.functional.glu(x, dim=1)`) is folded in on request:
 *   glu_in = 1: x is the PRE-GLU tensor [B*T, 2C]; the convolution runs on x[:, :C] * sigmoid(x[:, C:]), formed while the
 *               time window is staged (the GLU output is never written);
 *   glu_a != NULL (flip = 1): the conv result r (= gradient w.r.t. the GLU output) goes through the GLU backward on the
 *               way out: y is [B*T, 2C] = (r * sigmoid(g), r * a * sigmoid(g) * (1 - sigmoid(g))) for glu_a = [a | g]. */
int avsr_dwconv_fwd(const void* x, int dtype, const float* w, const float* bias, void* y, int B, int T,
                    int C, int K, int flip, int glu_in, const void* glu_a, avsr_stream_t stream);
/* f32 forward (flip = 0, no GLU backward epilogue) + the bf16 twin y2 of its output */
int avsr_dwconv_fwd2(const float* x, const float* w, const float* bias, float* y, void* y2, int B, int T, int C, int K, int glu_in,
                     avsr_stream_t stream);
/* dw[c,k] += sum dy[b,t,c] x[b,t+k-pad,c]; db[c] += sum dy; glu_in = 1: x is the pre-GLU tensor [B*T, 2C] */
int avsr_dwconv_wgrad(const void* x, const void* dy, int dtype, float* dw, float* db, int B, int T, int C,
                      int K, int glu_in, avsr_stream_t stream);

/* ---- training-mode BatchNorm (+SiLU, + residual add) on channels-last [rows, C] ------------------- */
int64_t avsr_bn_workspace_floats(int C);
/* stats [3][C] (shift, sum(x-shift), sum((x-shift)^2)), overwritten; workspace: avsr_bn_workspace_floats(C);
 * *count_out = (float)rows when != NULL (lets [stats | count] travel as one all-gather payload) */
int avsr_bn_stats(const void* x, int dtype, float* stats, float* workspace, int64_t rows, int C,
                  float* count_out, avsr_stream_t stream);
/* merge `world` per-rank partials (rank w: stats + w*stats_stride as [3][C], count counts[w*counts_stride]; strides in
 * floats, 0 = dense [world][3][C] / [world]) -> mean, invstd; momentum update of the running stats (unbiased
 * variance) when running_mean != NULL; *num_batches_tracked += 1 and *n_total = sum of counts when != NULL */
int avsr_bn_finalize(const float* stats, const float* counts, int world, int C, int64_t stats_stride,
                     int64_t counts_stride, float eps, float momentum, float* mean, float* invstd,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked, float* n_total,
                     avsr_stream_t stream);
/* single-rank training statistics: avsr_bn_stats + avsr_bn_finalize(world = 1) fused (two launches, not three) */
int avsr_bn_stats_finalize(const void* x, int dtype, float* workspace, int64_t rows, int C, float eps,
                           float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, avsr_stream_t stream);
int avsr_bn_eval_params(const float* running_mean, const float* running_var, float eps, int C, float* mean,
                        float* invstd, avsr_stream_t stream);
/* Single-launch BatchNorm(+activation) of a SMALL [rows, C] activation on ONE rank (the ConvolutionModule's BatchNorm1d,
 * conformer_encoder.py:26,33 + Swish :28; rows = B*T <= avsr_bn_small_max_rows()): batch statistics, running-stat /
 * num_batches_tracked update (as avsr_bn_stats_finalize) and y = act(gamma*(x-mean)*invstd + beta) (as avsr_bn_act_fwd);
 * mean / invstd [C] are kept for the backward.  avsr_bn_small_bwd: dx, dgamma = sum dz*xhat, dbeta = sum dz of the same
 * (as avsr_bn_bwd_reduce + avsr_bn_bwd_apply with n = rows).  Cross-rank synchronised BatchNorm keeps the three-phase entry
 * points (the collective sits between the phases). */
int avsr_bn_small_max_rows(void);
int avsr_bn_small_fwd(const void* x, int dtype, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                      float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, int act, void* y,
                      float* mean, float* invstd, avsr_stream_t stream);
/* f32 in, f32 (y_dtype 0: the "hpf" numerical mode) or f16 (y_dtype 2: the mixed mode's convolution module, whose element-wise
 * chain stays f32 up to this point) out + the bf16 twin y2 of the output */
int avsr_bn_small_fwd2(const float* x, int64_t rows, int C, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, int64_t* num_batches_tracked, int act, void* y, int y_dtype,
                       void* y2, float* mean, float* invstd, avsr_stream_t stream);
int avsr_bn_small_bwd(const void* x, const void* dy, int dtype, int64_t rows, int C, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, int act, void* dx, float* dgamma, float* dbeta,
                      avsr_stream_t stream);
/* The element-wise middle of the Conformer ConvolutionModule as ONE launch each way on a single rank (round 6;
 * conformer_encoder.py:32-34 `glu` -> `depthwise_conv` -> `norm` -> `activation`): a [B*T, 2C] (dtype 0 f32 / 1 bf16) -> GLU ->
 * depthwise Conv1d(K odd <= 31, weights wdw [C][K], bias bdw or NULL) -> BatchNorm1d over all B*T <= avsr_convmod_fused_max_rows()
 * frames (running statistics / batch counter updated, may be NULL) -> SiLU -> s [B*T, C] (s_dtype 0 f32 / 1 bf16 / 2 f16) + its
 * bf16 twin s2 (may be NULL).  The depthwise output the backward needs leaves as c_out (dtype of a) and / or c2 (bf16), either may
 * be NULL; mean / invstd [C].  Same arithmetic as avsr_dwconv_fwd(glu_in) + avsr_bn_small_fwd.
 * avsr_convmod_dwbn_bwd: ds -> BatchNorm + SiLU backward -> depthwise weight / bias gradient (ADDED to dwdw [C][K] / dbdw [C]) and
 * data gradient -> GLU backward -> da [B*T, 2C]; dgamma / dbeta [C] overwritten (avsr_bn_small_bwd + avsr_dwconv_wgrad +
 * avsr_dwconv_fwd(flip, glu_a) in one launch; a / c / ds / da share `dtype`). */
int avsr_convmod_fused_max_rows(void);
int avsr_convmod_dwbn_fwd(const void* a, int dtype, const float* wdw, const float* bdw, int B, int T, int C, int K,
                          const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                          float* running_var, int64_t* num_batches_tracked, void* c_out, void* c2, void* s, int s_dtype, void* s2,
                          float* mean, float* invstd, avsr_stream_t stream);
int avsr_convmod_dwbn_bwd(const void* a, const void* c, const void* ds, int dtype, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, const float* wdw, int B, int T, int C, int K, void* da,
                          float* dwdw, float* dbdw, float* dgamma, float* dbeta, avsr_stream_t stream);
/* y = act(gamma*(x-mean)*invstd + beta (+ add)); act 0 none, 1 SiLU */
int avsr_bn_act_fwd(const void* x, const void* add, int dtype, const float* mean, const float* invstd,
                    const float* gamma, const float* beta, void* y, int64_t rows, int C, int act,
                    avsr_stream_t stream);
/* f32 in / f32 out + the bf16 twin y2 of the output.  layout (round 5): bit 0 -- y is written in the "split8" layout (every group
 * of 8 consecutive values as its 8 hi bf16 + 8 lo bf16: the bytes of the f32 tensor, and what avsr_conv2d_f32s* consumes as a
 * pre-split A operand -- tile codes 23 - 26 -- without a conversion pass); bit 1 -- `add` is stored in that layout */
int avsr_bn_act_fwd2(const float* x, const void* add, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, void* y, void* y2, int64_t rows, int C, int act, int layout, avsr_stream_t stream);
/* maxpool(act(bn(x))) in one pass: y [N][OH][OW][C] + argmax idx (uint8, kh*K+kw of the first maximum) from
 * x [N][H][W][C]; replaces BatchNorm3d + SiLU + MaxPool3d((1,3,3),(1,2,2),(0,1,1)) of the video stem
 * (frontend/resnet.py:212-218) without materialising the full-resolution activation */
int avsr_bn_act_pool_fwd(const void* x, int dtype, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, void* y, uint8_t* idx,
                         void* xsel /* may be NULL; 3x3 / stride 2 / pad 1 only: [N][OH][OW][C], the RAW x at the arg-max --
                                       with it the backward reduce pass is avsr_bn_bwd_reduce(xsel, dpool) on the pooled
                                       tensors alone (sum over pooled outputs == sum over pixels) */,
                         int64_t N, int H, int W, int C, int K, int S, int P, int act, avsr_stream_t stream);
/* the 3x3 / stride 2 / pad 1 case on an f32 input with the bf16 twin y2 of the pooled f32 output and the arg-max inputs xsel2 in
 * bf16 written in the same pass (either may be NULL): the video stem of the hpf / mixed modes */
int avsr_bn_act_pool3_fwd2(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, void* y,
                           void* y2, uint8_t* idx, void* xsel2, int64_t N, int H, int W, int C, int act, int y_split8,
                           avsr_stream_t stream);
/* backward of avsr_bn_act_pool_fwd in the two BatchNorm backward passes, the activation gradient gathered from the pooled
 * gradient dpool [N][OH][OW][C] through idx (no full-resolution gradient tensor): sums [2][C] = (sum dz, sum dz*xhat),
 * then dx [N][H][W][C] from the (all-reduced) sums; workspace / inv_n / n_dev as avsr_bn_bwd_reduce / _apply */
int avsr_bn_pool_bwd_reduce(const void* x, const void* dpool, const uint8_t* idx, int dtype, const float* mean,
                            const float* invstd, const float* gamma, const float* beta, float* sums, float* workspace,
                            int64_t N, int H, int W, int C, int K, int S, int P, int act, avsr_stream_t stream);
int avsr_bn_pool_bwd_apply(const void* x, const void* dpool, const uint8_t* idx, int dtype, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, const float* sums, float inv_n,
                           const float* n_dev, void* dx, int64_t N, int H, int W, int C, int K, int S, int P, int act,
                           avsr_stream_t stream);
/* sums [2][C] = (sum dz, sum dz*xhat), dz = dy*act'(z); workspace as above */
int avsr_bn_bwd_reduce(const void* x, const void* dy, const void* add, int dtype, const float* mean,
                       const float* invstd, const float* gamma, const float* beta, float* sums, float* workspace,
                       int64_t rows, int C, int act, avsr_stream_t stream);
/* dx = gamma*invstd*(dz - sums0*inv_n - xhat*sums1*inv_n); dadd = dz (NULL skips); n_dev != NULL overrides
 * inv_n with 1 / *n_dev (device-resident global row count under cross-rank statistics) */
int avsr_bn_bwd_apply(const void* x, const void* dy, const void* add, int dtype, const float* mean,
                      const float* invstd, const float* gamma, const float* beta, const float* sums,
                      float inv_n, const float* n_dev, void* dx, void* dadd, int64_t rows, int C, int act,
                      avsr_stream_t stream);

/* ---- loss heads and decoder embedding (loss.hip) ------------------------------------------- */
int avsr_row_lse(const void* x, int dtype, int64_t ld, float* lse, int64_t rows, int V, avsr_stream_t stream);
int64_t avsr_ctc_workspace_bytes(int B, int T, int Lmax);
/* CTC (ctc.py:32-38,54-63; blank 0): logits [B,T,V] rows of pitch ld; labels int64 [B,Lmax] padded with
 * ignore_id; in_lens int64 [B].  nll[b] = -log p (inf if infeasible); grad (same dtype, pitch ldg, may be
 * NULL) = d nll[b]/d logits, zero for t >= in_lens[b] and for infeasible targets (zero_infinity); the pad columns [V, ldg) of
 * every row are written as zeros (round 6: the buffer need not be initialised). */
int avsr_ctc_loss(const void* logits, int dtype, int64_t ld, const int64_t* labels, int Lmax, int ignore_id,
                  const int64_t* in_lens, float* nll, void* grad, int64_t ldg, void* workspace, int B, int T,
                  int V, avsr_stream_t stream);
/* label-smoothing KL (label_smoothing_loss.py:41-63) per row + argmax hit (nets_utils.py:272-292);
 * grad = softmax - smoothed target (zero rows for ignored targets; pad columns [V, ldg) written as zeros), may be NULL */
int avsr_ce_smooth(const void* logits, int dtype, int64_t ld, const int64_t* target, int ignore_id, int V,
                   float smoothing, float* row_loss, float* row_hit, void* grad, int64_t ldg, int64_t rows,
                   avsr_stream_t stream);
/* out[0] = scale * sum(a[0..n)); finite_only skips +-inf entries (zero_infinity of ctc.py:26-28) */
int avsr_sum_scale(const float* a, int n, float scale, float* out, int finite_only, avsr_stream_t stream);
/* decoder targets of the attention branch in one launch (e2e_asr_conformer.py:138-139: add_sos_eos, add_sos_eos.py:12-31, and
 * target_mask, mask.py:11-37) with the static width L + 1: ys_pad [B, L] padded with ignore_id anywhere -> ys_in [B, L+1]
 * (<sos> labels <eos>...), ys_out [B, L+1] (labels <eos> <ignore_id>...), mask [B, L+1, L+1] bytes (may be NULL),
 * n_tokens[0] = #(ys_out != ignore_id) (may be NULL) */
int avsr_prepare_targets(const int64_t* ys_pad, int B, int L, int64_t sos, int64_t eos, int64_t ignore_id, int64_t* ys_in,
                         int64_t* ys_out, uint8_t* mask, int64_t* n_tokens, avsr_stream_t stream);
/* out[r,:] = dropout(table[ids[r],:]*scale + pe[r % L,:])  (transformer_decoder.py:186-189, embedding.py:78-87) */
int avsr_embed_fwd(const int64_t* ids, const float* table, const float* pe, float* out, int64_t rows, int L,
                   int D, float scale, float drop_p, uint64_t seed, const uint64_t* seed_dev, avsr_stream_t stream);
int avsr_embed_bwd(const int64_t* ids, const float* dout, float* dtable, int64_t rows, int D, float scale,
                   float drop_p, uint64_t seed, const uint64_t* seed_dev, avsr_stream_t stream);

/* out[r,:V] = x[r,:V] - logsumexp(x[r,:V]); f32, same pitch ld for x and out; lse_ws: [rows] scratch */
int avsr_log_softmax(const float* x, int64_t ld, float* lse_ws, float* out, int64_t rows, int V,
                     avsr_stream_t stream);

/* ---- implicit-GEMM convolutions on channels-last activations (gemm_conv.hip) -------------------- */
/* torch weight [Cout][Cin][taps] (f32) -> out[a][tap][b] with row pitch ld_out, a/b = co/ci (to_dgrad=0: forward and
 * weight-gradient layout) or ci/co (to_dgrad=1: data-gradient layout), cast to out_dtype (0 f32, 1 bf16, 2 = the split8 layout of
 * avsr_split_pack over the dense [a][tap][b] order; ld_out must then be taps * b) */
int avsr_conv_weight_permute(const float* w, void* out, int out_dtype, int Cout, int Cin, int taps, int to_dgrad,
                             int64_t ld_out, avsr_stream_t stream);
/* dw[Cout][Cin][taps] = dwp[Cout][taps][Cin] */
/* all conv weights of a model in one launch: table of 48-byte entries {const float* w, bf16* out, int Cout, Cin, taps,
 * to_dgrad, blk0, 0, 0, 0} in device memory, blk0 = running sum of avsr_weight_permute_blocks(Cout, Cin, to_dgrad) (one block =
 * one 8 x 64 / 64 x 8 (co, ci) tile x all taps, transposed through LDS); max_taps = the largest taps of the table (<= 64) */
int64_t avsr_weight_permute_blocks(int Cout, int Cin, int to_dgrad);
int avsr_multi_weight_permute(const void* table, int n, int total_blocks, int max_taps, avsr_stream_t stream);
int avsr_conv_weight_unpermute(const float* dwp, float* dw, int Cout, int Cin, int taps, avsr_stream_t stream);
/* y[N,OH,OW,Cout] = conv(x[N,H,W,Cin], wp[Cout][KH][KW][Cin])   (frontend/resnet.py:10-17,20-35; H = 1 for 1-D) */
int avsr_conv2d_fwd(const void* x, int dtype, const void* wp, int w_dtype, void* y, int N, int H, int W, int Cin,
                    int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int precise, avsr_stream_t stream);
/* dx[N,H,W,Cin] = conv^T(dy[N,OH,OW,Cout], wpd[Cin][KH][KW][Cout]) (+ resid, same dtype/shape, may be NULL) */
int avsr_conv2d_dgrad(const void* dy, int dtype, const void* wpd, int w_dtype, const void* resid, void* dx, int N,
                      int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int precise,
                      avsr_stream_t stream);
/* dwp[Cout][KH][KW][Cin] (f32, caller zeroes) += dy^T im2col(x) */
int avsr_conv2d_wgrad(const void* dy, const void* x, int dtype, float* dwp, int N, int H, int W, int Cin, int Cout,
                      int KH, int KW, int stride, int pad_h, int pad_w, int precise, avsr_stream_t stream);
/* single-input-channel stem with temporal taps (resnet.py:204-211 Conv3d(1,64,(5,7,7),s(1,2,2)); resnet1d.py:124-131
 * Conv1d(1,64,80,s4) with KT=KH=1): x f32 [B,T,H,W]; wp [Cout][ldw] */
int avsr_conv_stem_fwd(const float* x, const void* wp, int w_dtype, int ldw, void* y, int y_dtype, int B, int T, int H,
                       int W, int Cout, int KT, int KH, int KW, int stride, int pad_t, int pad_h, int pad_w,
                       int precise, avsr_stream_t stream);
/* dw[Cout][KT*KH*KW] (f32, caller zeroes) += dy^T im2col(x) */
int avsr_conv_stem_wgrad(const void* dy, int dy_dtype, const float* x, float* dw, int B, int T, int H, int W, int Cout,
                         int KT, int KH, int KW, int stride, int pad_t, int pad_h, int pad_w, int precise,
                         avsr_stream_t stream);

/* ---- pooling on channels-last tensors (pool.hip) ------------------------------------------------- */
/* idx (may be NULL): uint8 [N,OH,OW,C], window position kh*K+kw of the first maximum (what the backward routes to) */
int avsr_maxpool2d_fwd(const void* x, void* y, uint8_t* idx, int dtype, int64_t N, int H, int W, int C, int K, int S, int P,
                       avsr_stream_t stream);
int avsr_maxpool2d_bwd(const uint8_t* idx, const void* dy, void* dx, int dtype, int64_t N, int H, int W, int C, int K, int S,
                       int P, avsr_stream_t stream);
/* y[g,:] = mean of rows g*win .. g*win+win-1 of x [groups*win, C]; y f32 */
int avsr_avgpool_fwd(const void* x, int dtype, float* y, int64_t groups, int win, int C, avsr_stream_t stream);
int avsr_avgpool_bwd(const float* dy, void* dx, int dtype, int64_t groups, int win, int C, avsr_stream_t stream);

/* ---- tuned bf16 NT GEMM (gemm_fast.hip): LDS-DMA operand ring, swizzled LDS, counted vmcnt ------------------ */
/* C[M,N] = epi(A[M,K] . B[N,K]^T), A and B bf16 k-contiguous, K % 64 == 0; epilogue as avsr_gemm (resid may be
 * f32 or bf16); tile: 0 auto, 1 = 64x64, 2 = 128x64, 3 = 128x128 (3-stage ring, 4 waves), 4 = 128x128 with a 2-stage
 * ring, 5 = 256x128 with 8 waves, 6 = 256x128 with 4 waves, 7 = 128x64 with a 2-stage ring, 8 = 256x64 with 8 waves */
int avsr_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias, int act,
                      const void* gate, int gate_dtype, int ldg, float gate_scale, float drop_p, uint64_t seed,
                      const uint64_t* seed_dev, float alpha, const float* alpha_dev, const void* resid, int resid_dtype,
                      int ldr, void* C, int c_dtype, int ldc, int accumulate, int split_k, int tile,
                      float* colsum /* may be NULL: colsum[n] += sum_m C[m,n] (f32, before rounding; accumulate = 0) */,
                      avsr_stream_t stream);
/* ---- tuned NT GEMM of the precise mode (gemm_split.hip): f32 operands, split hi + lo bf16 planes formed in registers --------- */
/* C[M,N] = epi(A[M,K] . B[N,K]^T), A and B f32 k-contiguous (lda, ldb % 4 == 0), K % 64 == 0; three bf16 MFMAs per product
 * (the arithmetic of avsr_gemm with precise = 1) on the LDS-DMA operand ring of avsr_gemm_bf16_nt; epilogue and arguments as
 * avsr_gemm_bf16_nt; tile: 0 auto, 1 = 64x64 / 3 stages, 2 = 64x64 / 2 stages, 3 = 128x64 / 2 stages, 4 = 128x128 / 2 stages,
 * 5 / 6 = 3 / 4 with 3 stages, 7 = 128x64 / 2 stages with a 4 x 1 wave grid, 11 .. 16 = 1 .. 6 with the staged A tile converted to
 * split8 in place once per stage (the default) instead of on every fragment read.
 * Forward contractions of the precise / hpf modes: positionwise_feed_forward.py:24-30, attention.py:31-34,123,
 * conformer_encoder.py:24,27, e2e_asr_conformer.py:31, ctc.py:21, transformer_decoder.py:225. */
int avsr_gemm_f32s_nt(const float* A, int lda, const float* B, int ldb, int M, int N, int K, const float* bias, int act,
                      const void* gate, int gate_dtype, int ldg, float gate_scale, float drop_p, uint64_t seed,
                      const uint64_t* seed_dev, float alpha, const float* alpha_dev, const void* resid, int resid_dtype,
                      int ldr, void* C, int c_dtype, int ldc, int accumulate, int split_k, int tile, float* colsum,
                      int b_split /* 1: B is in the split8 layout of avsr_split_pack (same pitch) */,
                      void* c2 /* may be NULL: bf16 twin of an f32, non-accumulating C (row pitch ldc2) */, int ldc2,
                      avsr_stream_t stream);
/* split8 layout: every group of 8 consecutive f32 of a buffer replaced, in place of its 32 bytes, by its 8 hi bf16 followed
 * by its 8 lo bf16 (hi = bf16(x), lo = bf16(x - hi)); n % 8 == 0.  avsr_multi_split_pack: many tensors in one launch, table
 * of 32-byte entries {const float* src, void* dst, int64 n / 8, int64 blk0}, blk0 = running sum of ceil(n / 2048). */
int avsr_split_pack(const float* src, void* dst, int64_t n, avsr_stream_t stream);
int avsr_multi_split_pack(const void* table, int n, int total_blocks, avsr_stream_t stream);
/* f32 convolution forward on the same kernel (implicit GEMM, channels-last, Cin % 64 == 0; frontend/resnet.py:10-35):
 * x[N,H,W,Cin] * wp[Cout][KH][KW][Cin] -> y[N,OH,OW,Cout], all f32; zero_page: >= 16 zero bytes of device memory.
 * tile: 0 = automatic (128 x 128 for Cout >= 128, else 128 x 64; the staged A tile converted to split8 in LDS once per stage);
 * 23 / 24 (25 / 26 with a 3-stage ring) = 128 x 64 / 128 x 128 on an x that ARRIVES in the split8 layout (w_split = 1 required:
 * avsr_bn_act_fwd2 layout bit 0 / avsr_bn_act_pool3_fwd2 y_split8 write it) -- no conversion pass, one barrier per k-tile */
int avsr_conv2d_f32s(const float* x, const float* wp, float* y, const void* zero_page, int N, int H, int W, int Cin, int Cout,
                     int KH, int KW, int stride, int pad_h, int pad_w, int tile, int w_split /* 1: wp in the split8 layout */,
                     void* y2 /* may be NULL: bf16 twin of y */, avsr_stream_t stream);
/* the same convolution leaving the BatchNorm statistics of its output behind (frontend/resnet.py:82-98: every trunk convolution
 * feeds a BatchNorm in batch-statistics mode): row t of stats_part [stats_tiles >= ceil(rows / 128)][2][Cout] (need not be
 * initialised) = column sums / sums of squares of the stored values of output rows [128 t, 128 t + 128), from the epilogue -- no
 * statistics pass over y; finish with avsr_bn_finalize_parts (one rank) or avsr_bn_stats_parts + avsr_bn_finalize (cross-rank);
 * ws: 512 * C floats of scratch, zeros: >= C zero floats */
int avsr_conv2d_f32s_stats(const float* x, const float* wp, float* y, const void* zero_page, int N, int H, int W, int Cin,
                           int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int tile, int w_split, void* y2,
                           float* stats_part, int stats_tiles, avsr_stream_t stream);
int avsr_bn_finalize_parts(const float* part, int ntiles, int C, const float* zeros, float* ws, int64_t rows, float eps, float momentum,
                           float* mean, float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           avsr_stream_t stream);
int avsr_bn_stats_parts(const float* part, int ntiles, int C, const float* zeros, float* ws, float* stats, float* count_out, int64_t rows,
                        avsr_stream_t stream);
/* ---- collectives of the data-parallel step straight on RCCL (csrc/comm.hip; RCCL is bound with dlopen at the first call).
 * Replaces, for graph-captured steps, the torch.distributed calls of train.py:30-42 (DDP gradient all-reduce, SyncBatchNorm
 * statistics) and lightning.py:88-90 (batch-size all-gather): every call is ONE stream operation on `stream`, nothing else.
 * Up to four communicators per process (slot 0..3; RCCL serialises the operations of one communicator across streams -- the
 * gradient buckets and the BatchNorm collectives use one each).  avsr_comm_unique_id: rank 0 fills 128 bytes, the host side
 * distributes them; avsr_comm_init: collective over all ranks (device = the calling thread's current HIP device). */
int avsr_comm_unique_id(void* out128);
int avsr_comm_init(int slot, const void* id128, int nranks, int rank);
int avsr_comm_destroy(int slot);
int64_t avsr_comm_size(int slot); /* 0 without a communicator */
int avsr_comm_all_reduce_f32(int slot, void* buf, int64_t count, avsr_stream_t stream);
/* the same for dtype 0 = f32 / 1 = bf16 (narrow wire format of the gradient buckets: half the bytes per xGMI link) */
int avsr_comm_all_reduce(int slot, void* buf, int64_t count, int dtype, avsr_stream_t stream);                          /* in place, sum */
int avsr_comm_all_gather_f32(int slot, const void* send, void* recv, int64_t count_per_rank, avsr_stream_t stream); /* recv: nranks x count */
/* recv[0 .. count_per_rank) = sum over ranks of send[rank * count_per_rank ..) (ncclReduceScatter; dtype 0 = f32, 1 = bf16; recv may
 * alias its own slice of send): the gradient exchange of the sharded optimizer -- train.py:30-42 (DDP averaging) at half the wire bytes
 * inside the backward pass, the other half being the all-gather of the updated weights */
int avsr_comm_reduce_scatter(int slot, const void* send, void* recv, int64_t count_per_rank, int dtype, avsr_stream_t stream);
/* Tuning knobs of the tuned kernels (process-wide; meant for benchmarks, defaults are the measured best):
 * knob 0 = tile code forced on avsr_conv2d_bf16 (0 = auto), 1 = XCD-aware tile order (0 = automatic: on for the 64x64 GEMM tile, whose
 * operands are not cache-resident in the training step; 1 = always on; 2 = always off),
 * 2 = ablation mode of the 128x128 forward convolution kernel (1 = no MFMA, 2 = no operand loads; wrong results),
 * 3 = avsr_conv3x3_wgrad_bf16 output stage (0 = as the workspace argument says, 1 = always atomics, 2 = none),
 * 4 = 1 disables the XCD-aware work order of avsr_conv3x3_wgrad_bf16, 5 = ablation bits of avsr_conv3x3_wgrad_bf16 (4 = no staging
 * after the first tile, 16 = no tiles),
 * 6 = persistent-block count of the video-stem weight gradient (0 = default 512), 7 = per-block rotation of the k order in
 * avsr_gemm_bf16_nt (0 = off; measured neutral), 8 / 9 = 1 selects the generic attention forward / backward-dq kernel for
 * bf16 inputs instead of the transposed-formulation kernels, 10 = 1 selects the generic batched TN path of
 * avsr_attention_bwd_kv instead of the k-major tile kernel, 11 = bit mask of that kernel's contractions to skip (fault isolation),
 * 12 = 1 keeps 64 -> 64 channel 3x3 / stride-1 convolutions on the tiled kernel instead of conv3x3_c64.hip, 13 = ablation bits
 * of conv3x3_c64.hip (1 = no MFMA loop, 2 = no staging, 4 = no copy-out; wrong results), 14 = tile code forced on the
 * avsr_gemm_bf16_nt problems whose 64x64 grid has 257..512 tiles (0 = auto), 15 = block-count target of avsr_conv3x3_wgrad_bf16
 * (0 = one resident set: 512 blocks for the default variant, 256 for the others), 16 = variant of avsr_conv3x3_wgrad_bf16 (0 / 1 =
 * four thin waves, two blocks per CU; 2 = four fat waves in two k groups, one block per CU; 3 = eight thin waves in two k groups --
 * both measured slower, kept for A/B runs), 17 / 18 = tile code forced on the two-plane f16 GEMM / convolution (avsr_gemm_h16_nt with
 * B_lo, avsr_conv2d_h16 with wp_lo), 21 / 22 = tile code forced on avsr_conv2d_f32s(_stats) calls whose activation arrives in the
 * split8 layout (Cout < 128 / Cout >= 128: 23 .. 30, see gemm_split.hip), 20 = the patch-staged 3x3 kernel of conv_patch.hip (1 = off:
 * keep the tiled kernel; 2 = on whatever the grid size), 23 = deterministic mode (see auto_avsr_amd.functional.set_deterministic),
 * 24 = block-count target of avsr_conv2d_wgrad_bf16's k split (0 = by tile count: 512 / 1024 / 2048), 25 = rows per wave of
 * avsr_layernorm_bwd's dx blocks (0 = 1, or 2 with a column-sum output on > 1024 rows; tools/microbench_small.py), 26 = k split of
 * avsr_gemm_h16_nt with f32 atomics onto a ZEROED f32 C (probe: tools/microbench_splitk.py; slower at M = 1600).  Knobs 0..31 exist. */
int avsr_tune(int knob, int value);
/* bf16 implicit-GEMM convolution on the tuned LDS-DMA kernel: dgrad = 0 forward, 1 data gradient (see
 * avsr_conv2d_fwd / avsr_conv2d_dgrad for the tensor conventions); gathered channel count % 64 == 0; stride 1 or 2
 * (a strided data gradient runs as stride^2 dense sub-problems, one per residue class of the input pixel);
 * zero_page: >= 16 zero bytes of device memory (source of the padding taps) */
int avsr_conv2d_bf16(int dgrad, const void* src, const void* wp, const void* resid, void* out, const void* zero_page,
                     int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_h, int pad_w,
                     avsr_stream_t stream);
/* dst[c][r] = bf16(src[r][c]); dst row pitch ld_dst >= R, columns [R, ld_dst) zero-filled */
int avsr_transpose_cast(const void* src, int src_dtype, int64_t ld_src, void* dst, int64_t ld_dst, int R, int C,
                        avsr_stream_t stream);

/* backward prologue of a Linear layer in one pass: v = alpha*dropout(src[R][C]); dst = bf16(v) (NULL skips);
 * dstT = bf16(v)^T with row pitch ld_dstT >= R, zero tail (NULL skips); colsum[C] += column sums (NULL skips) */
int avsr_cast_transpose_colsum(const void* src, int src_dtype, int64_t ld_src, void* dst, void* dstT, int64_t ld_dstT,
                               float* colsum, int R, int C, float alpha, const float* alpha_dev, float drop_p,
                               uint64_t seed, const uint64_t* seed_dev, avsr_stream_t stream);

/* one launch: for each of n table entries (64 bytes: {const float* src; bf16* dst; bf16* dstT; int R, C, ldT, blk0,
 * tiles_c, 0, 0, 0}) write the bf16 copy dst[R][C] and/or the transposed copy dstT[C][ldT] (zero tail) of src */
int avsr_multi_cast_transpose(const void* table, int n, int total_blocks, avsr_stream_t stream);

/* dedicated bf16 kernels for the Conv3d(1,64,(5,7,7),s(1,2,2),p(2,3,3)) visual stem (resnet.py:204-211): the 35
 * (kt,kh) input rows of an output row are staged once in LDS.  workspace: avsr_stem357_workspace_bytes() bytes */
int64_t avsr_stem357_workspace_bytes(void);
int avsr_stem357_fwd(const float* x, const float* w, void* y, void* workspace, int B, int T, int H, int W,
                     avsr_stream_t stream);
int avsr_stem357_wgrad(const void* dy, const float* x, float* dw, void* workspace, int B, int T, int H, int W,
                       avsr_stream_t stream);
/* precise / hpf modes: the same convolution with an f32 result from split hi + lo bf16 planes (three MFMAs per product) */
int avsr_stem357_fwd_f32s(const float* x, const float* w, float* y, void* y2 /* may be NULL: bf16 twin of y */, void* workspace,
                          int B, int T, int H, int W, avsr_stream_t stream);
/* ... leaving the BatchNorm statistics of its output behind (frontend/resnet.py:203-219: Conv3d -> BatchNorm3d in batch-statistics
 * mode): row j of stats_part [stats_rows >= avsr_stem357_stat_rows(B, T, H)][2][64] = per-channel sums / sums of squares of the
 * output rows block j wrote; finish with avsr_bn_finalize_parts / avsr_bn_stats_parts */
int64_t avsr_stem357_stat_rows(int B, int T, int H);
int avsr_stem357_fwd_f32s_stats(const float* x, const float* w, float* y, void* y2, void* workspace, int B, int T, int H, int W,
                                float* stats_part, int stats_rows, avsr_stream_t stream);

/* bf16 weight-gradient contraction without transposed copies (gemm_tn_fast.hip: LDS-DMA k-major tiles +
 * ds_read_b64_tr_b16): C[M][N] (f32, ldc) (+)= sum_k A[k][m] B[k][n]; A [K][lda], B [K][ldb] bf16.
 * colsum_a (f32 [M], may be NULL; accumulated into) += sum_k A[k][m]: with A = dY this is the bias gradient of the Linear
 * whose weight gradient the call computes (torch: grad_bias = dY.sum(0)), taken from the A tiles the kernel stages anyway */
int avsr_gemm_bf16_tn(const void* A, int lda, const void* B, int ldb, int M, int N, int K, float* C, int ldc,
                      int accumulate, int split_k, const void* zero_page, float* colsum_a, avsr_stream_t stream);
/* Paired launch of the two GEMMs of a Linear backward pass (gemm_pair.hip): between begin and end, the first
 * avsr_gemm_bf16_nt call (split_k 1, non-accumulating, 64x64 / 128x64 tile shapes) and the first avsr_gemm_bf16_tn call
 * of the calling thread are recorded and then launched together by avsr_gemm_pair_end as ONE grid (NT tiles first, TN
 * tiles after) -- the two problems must be independent.  Calls the pair cannot hold launch immediately, as usual. */
int avsr_gemm_pair_begin(void);
int avsr_gemm_pair_end(void);
/* convolution weight gradient on the same kernel (B = im2col gather); dwp [Cout][KH][KW][Cin] f32, caller zeroes */
int avsr_conv2d_wgrad_bf16(const void* dy, const void* x, float* dwp, const void* zero_page, int N, int H, int W,
                           int Cin, int Cout, int KH, int KW, int stride, int pad_h, int pad_w,
                           avsr_stream_t stream);
/* weight gradient of a 3x3 / pad 1 convolution, stride 1 or 2 (conv_wgrad.hip; reference resnet.py:10-35 under
 * autograd): one zero-padded x patch per pixel tile serves all nine taps as shifted LDS views; Cin, Cout % 64 == 0;
 * dwp [Cout][3][3][Cin] f32.  With a workspace of avsr_conv3x3_wgrad_workspace_bytes() dwp is overwritten with the
 * ordered (deterministic) sum of per-block partial gradients; with workspace = NULL the blocks atomicAdd into dwp,
 * which the caller must have zeroed. */
int64_t avsr_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int stride);
int avsr_conv3x3_wgrad_bf16(const void* dy, const void* x, float* dwp, const void* zero_page, void* workspace,
                            int64_t workspace_bytes, int N, int H, int W, int Cin, int Cout, int stride,
                            int torch_layout /* 1: write [Cout][Cin][3][3] (workspace mode only) */,
                            avsr_stream_t stream);
/* ---- beam-search support (ctc_prefix.hip): CTC prefix scores of every (running hypothesis, candidate token) pair in
 * one launch -- the python loop over frames of espnet/nets/ctc_prefix_score.py:71-187 (CTCPrefixScoreTH.__call__).
 * logp [T][ldv] f32 log-softmax of one utterance; r_prev [T][2][NH]; last [NH]; cand [NH][S]; out_len = len(prefix)-1;
 * r_new [T][2][NH][S]; psi [NH][S]; psi_eos [NH] */
int avsr_ctc_prefix_score(const float* logp, int T, int V, int ldv, const float* r_prev, const int64_t* last,
                          const int64_t* cand, int NH, int S, int out_len, int blank, float* r_new, float* psi,
                          float* psi_eos, avsr_stream_t stream);

/* ---- beam search, one decoding step per host call (decode.hip) ------------------------------------------------------
 * One iteration of BatchBeamSearch.search (espnet/nets/batch_beam_search.py:208-349 over beam_search.py:330-406) with the
 * reference's scorer wiring (lightning.py:126-158): TransformerDecoder.batch_score (decoder/transformer_decoder.py:226-258,
 * 301-334), CTCPrefixScorer.batch_score_partial (scorers/ctc.py:101-126), LengthBonus, pre-beam on the decoder scores, top-k
 * over beam x vocabulary, state selection -- issued from C++ (~75 launches, one device-to-host copy, one stream sync) on
 * [position][slot] K / V caches with per-hypothesis ancestry instead of re-projecting every previous position.
 * cfg: D, H, FF, V, n_layers, beam, S (pre-beam size), sos, eos, blank, has_length_bonus, rows of the position table.
 * fcfg: w_decoder, w_ctc, w_length_bonus, embedding scale, LayerNorm eps.
 * w: embed [V][D], position table [rows][D], per layer {norm1 g, b, W_qkv [3D][D], b_qkv, W_o, b_o, norm2 g, b, src W_q, b_q,
 * src W_kv [2D][D], b_kv, src W_o, b_o, norm3 g, b, W_1 [FF][D], b_1, W_2 [D][FF], b_2}, after_norm g, b, output W [V][D], b;
 * all f32 device pointers that outlive the session.  Returns an opaque handle, 0 on an unsupported configuration. */
int64_t avsr_beam_create(const int32_t* cfg, const float* fcfg, const void* const* w, int n_w);
int avsr_beam_destroy(int64_t handle);
int64_t avsr_beam_workspace_bytes(int64_t handle, int T, int Lmax);
/* the linear layer of a decoding step on its own: C = act(LN?(A) W^T + bias) + resid for M <= 128 rows (transformer_decoder.py:84-126
 * on one position per hypothesis); st_in [M][st_in_nt][2] per-row (sum, sum of squares) partials of A when ln_g != NULL;
 * st_out [M][ceil(N/16)][2] (may be NULL) the same for the rows of C, partial count per row through st_out_nt; partial: scratch of
 * 8 * M * N floats, used when K > 768 */
int avsr_decode_linear(const float* A, int lda, const float* W, int M, int N, int K, const float* bias, const float* ln_g,
                       const float* ln_b, float eps, const float* st_in, int st_in_nt, int act, const float* resid, int ldr,
                       float* C, int ldc, float* st_out, int* st_out_nt, float* partial, avsr_stream_t stream);
/* new utterance: memory [T][D] f32 encoder output, ctc_logp [T][ld_ctc] f32 log-softmax of the CTC head, r_init [T][2] CTC
 * state of the empty prefix (ctc_prefix_score.py:60-66), workspace of avsr_beam_workspace_bytes(handle, T, Lmax) */
int avsr_beam_begin(int64_t handle, const float* memory, int T, const float* ctc_logp, int ld_ctc, const float* r_init,
                    void* workspace, int64_t workspace_bytes, int Lmax, avsr_stream_t stream);
/* one step for all running hypotheses; host_out [K][8] f32 = {token, parent, total, decoder sum, ctc sum, length sum, 0, 0},
 * valid on return (synchronises the stream); K through n_out */
int avsr_beam_step(int64_t handle, float* host_out, int* n_out, avsr_stream_t stream);
/* drop the hypotheses not listed (ended ones, batch_beam_search.py:178-206); keep: ascending indices into the current beam */
int avsr_beam_keep(int64_t handle, const int32_t* keep, int n_keep, avsr_stream_t stream);
/* token sequences of the current beam into host_yseq [n][*ldy_out] (first *L_out entries of a row valid); synchronises */
int avsr_beam_fetch_yseq(int64_t handle, int64_t* host_yseq, int* ldy_out, int* L_out, avsr_stream_t stream);

/* ---- optimizer step (optim.hip): global-norm clip + AdamW + warm-up cosine schedule, all parameters in 3 launches ---
 * Replaces torch.nn.utils.clip_grad_norm_(params, max_grad_norm) + torch.optim.AdamW(...).step() +
 * WarmupCosineScheduler.step() (reference lightning.py:48-52, train.py:41, cosine.py:6-25).
 * table: n entries of 48 bytes {float* p, const float* g, float* m, float* v, int64 numel, int blk0, int 0} in device
 * memory, blk0 = running sum of ceil(numel / 4096), total_blocks = the final sum; partial: total_blocks floats of
 * scratch; state: 4 device floats {step, lr, grad_norm, clip_coef} -- step starts at 0 and is incremented by the call,
 * lr = base_lr * warm-up-cosine(step) (total_steps <= 0: constant).  max_grad_norm <= 0 disables clipping. */
int avsr_adamw_step(const void* table, int n, int total_blocks, float* partial, float* state, float base_lr, float beta1,
                    float beta2, float eps, float weight_decay, float max_grad_norm, int64_t warmup_steps,
                    int64_t total_steps, avsr_stream_t stream);
/* The same step in two halves for an optimizer sharded over the data-parallel ranks (every rank updates 1 / N of every flat bucket;
 * lightning.py:48-52, train.py:37,41): avsr_multi_sumsq leaves the sum of squares of THIS rank's gradient slices in sumsq[0]; the
 * caller all-reduces that float; avsr_adamw_apply derives clip coefficient / step / learning rate from the global sum and runs
 * AdamW on the table's slices.  Same table format and state as avsr_adamw_step. */
int avsr_multi_sumsq(const void* table, int n, int total_blocks, float* partial, float* sumsq, avsr_stream_t stream);
int avsr_adamw_apply(const void* table, int n, int total_blocks, const float* sumsq, float* state, float base_lr, float beta1,
                     float beta2, float eps, float weight_decay, float max_grad_norm, int64_t warmup_steps, int64_t total_steps,
                     avsr_stream_t stream);
/* dst_i = scale * src_i for n f32 tensors in ONE launch: table entries as above with p = dst, g = src (m, v unused).  The
 * data-parallel gradient exchange gathers a bucket's gradients into its flat RCCL all-reduce buffer with it (train.py:37
 * DDPStrategy: gradient averaging = scale 1 / world). */
int avsr_multi_copy_scale(const void* table, int n, int total_blocks, float scale, avsr_stream_t stream);
/* The same step with the bf16 operand copies of 2-D weights (see avsr_multi_cast_transpose) rewritten in the update pass,
 * so that no separate re-cast launch re-reads the f32 weights before the next forward pass:
 *   table / n / total_blocks          every parameter (48-byte entries as above) -- gradient norm only
 *   lin_table / lin_n / lin_blocks    parameters updated by the linear kernel (same format, own blk0 numbering)
 *   tile_table / tile_n / tile_blocks 80-byte entries {float* p, const float* g, float* m, float* v, bf16* dst,
 *       bf16* dstT (either may be 0), int R, C, ldT, blk0, tiles_c, limT, f16* dst16 (may be 0: IEEE-half [R][C] copy, the
 *       forward operand of the mixed mode's f16 components)}: weight [R][C] updated in 64x64 tiles,
 *       dst = bf16 [R][C], dstT = bf16 [C][ldT] (rows [R, limT) zero; limT 0 = ldT), blk0 = running sum of
 *       ceil(max(R, limT ? limT : ldT)/64) * ceil(C/64)
 * Every parameter must appear in exactly one of lin_table / tile_table. */
int avsr_adamw_cast_step(const void* table, int n, int total_blocks, const void* lin_table, int lin_n, int lin_blocks,
                         const void* tile_table, int tile_n, int tile_blocks, float* partial, float* state,
                         float base_lr, float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                         int64_t warmup_steps, int64_t total_steps, avsr_stream_t stream);

/* ---- input pipeline on the device (augment.hip): the per-sample transforms of the reference's DataLoader workers and its
 * padding collation, one launch per batch.
 * Replaces VideoTransform (datamodule/transforms.py:89-110: x/255, Random/CenterCrop(88), Grayscale,
 * AdaptiveTimeMask(10,25), Normalize(0.421,0.165)), AudioTransform (:113-136: AdaptiveTimeMask(6400,16000), AddNoise
 * (:67-88, torchaudio.functional.add_noise), layer_norm eps 1e-8) and pad / collate_pad (datamodule/data_module.py:10-41).
 * Random decisions (crop origin, masking intervals, noise offset, SNR) are drawn on the host and passed in.
 * avsr_video_transform: src_ptr[b] = device address of clip b, uint8 [lens[b]][H][W][3] (decoder layout);
 *   out [B][Tmax][1][crop][crop] (out_dtype 0 f32 / 1 bf16), frames >= lens[b] zero; iv int32 [B][max_iv][2] frame
 *   intervals [start, end) to mask, niv[b] used, NULL = none.  f32 results are bit-identical to the torch CPU ops.
 * avsr_audio_transform: wav_ptr[b] = device address of utterance b, f32 [lens[b]]; out f32 [B][Lmax][1], zero tail; iv in
 *   samples; noise (NULL = none) f32 recording, utterance b uses noise[noise_start[b] + i] at snr_db[b] (start < 0: clean);
 *   workspace: avsr_audio_transform_workspace_bytes(B, Lmax) bytes of device scratch. */
int avsr_video_transform(const int64_t* src_ptr, const int32_t* lens, const int32_t* crop_y,
                         const int32_t* crop_x, const int32_t* iv, const int32_t* niv, int max_iv, void* out,
                         int out_dtype, int B, int Tmax, int H, int W, int crop, float mean, float std,
                         avsr_stream_t stream);
int64_t avsr_audio_transform_workspace_bytes(int B, int64_t Lmax);
int avsr_audio_transform(const int64_t* wav_ptr, const int32_t* lens, const int32_t* iv,
                         const int32_t* niv, int max_iv, const float* noise, const int64_t* noise_start,
                         const float* snr_db, float eps, float* out, int B, int64_t Lmax, void* workspace,
                         avsr_stream_t stream);

/* ---- "mixed" numerical mode: f16 forward operands of the Conformer encoder -------------------------------------------------
 * The north star bounds logits / CTC log-probabilities at 1e-3 relative to the fp32 reference; with bf16 operands (8
 * significant bits) the 12-layer encoder alone generates 3-6e-3, with IEEE half (11 bits, the same 2 bytes and the same MFMA
 * rate: v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16) 4.6e-4 (tools/precision_study.py).  These entry points are the
 * forward kernels of the encoder sub-layers on f16 activations; each can write the bf16 twin of its result in the same pass,
 * which is what the (unchanged, bf16) backward pass reads.  Replaced reference lines as for the bf16 entry points:
 * layer_norm.py:12-33; positionwise_feed_forward.py:24-30, attention.py:31-34,123, conformer_encoder.py:24,27 (Linear /
 * pointwise conv forward); attention.py:59-88,153-193 (fused attention forward); attention.py:176-178 (q + pos_bias_u / v);
 * conformer_encoder.py:25,32,33 (GLU + depthwise conv); conformer_encoder.py:26,33-34 (BatchNorm1d + Swish). */
int avsr_layernorm_fwd_h16(const float* x, const float* gamma, const float* beta, void* y, void* y2, float* mean, float* rstd,
                           int rows, int cols, float eps, avsr_stream_t stream);
/* C[M,N] = epi(A[M,K] . B[N,K]^T), A and B f16 k-contiguous (lda, ldb % 8 == 0), K % 64 == 0; epilogue +bias -> act -> dropout
 * -> *alpha -> +resid; C f32 / bf16 / f16 (c_dtype 0 / 1 / 2); c2 (may be NULL): bf16 twin of an f32 / f16 C, row pitch ldc2;
 * tile: 0 auto, 1 = 64x64, 4 = 128x128, 7 = 128x64.
 * B_lo (round 5; may be NULL): second plane of the WEIGHT, f16((w - B) * 2^11), same pitch ldb -- every product is formed with
 * both planes (two MFMAs), so the weight enters with ~22 significant bits and only the activation's f16 rounding is left
 * (decoder-logit error of the mixed mode 0.65x of the one-plane figure).  The forward copies this build keeps are
 * row-interleaved [N][2][K] (hi row, lo row): B = copy, B_lo = copy + K, ldb = 2 K. */
int avsr_gemm_h16_nt(const void* A, int lda, const void* B, const void* B_lo, int ldb, int M, int N, int K, const float* bias, int act,
                     float drop_p, uint64_t seed, const uint64_t* seed_dev, float alpha, const void* resid, int resid_dtype,
                     int ldr, void* C, int c_dtype, int ldc, int tile, void* c2, int ldc2, avsr_stream_t stream);
/* f16 forward convolution (resnet.py:10-35 in the mixed mode): x [N,H,W,Cin] f16 * wp [Cout][KH][KW][Cin] f16 -> y f16 (+ bf16 twin y2,
 * may be NULL); Cin % 64 == 0, stride 1 or 2; wp from avsr_conv_weight_permute (out_dtype 3; dense, ldw = 0) or from
 * avsr_multi_weight_permute (entry field pad0 = 2: the two-plane image [Cout][2][KH][KW][Cin] -> wp = image, wp_lo = image +
 * KH KW Cin, ldw = 2 KH KW Cin); wp_lo (may be NULL): scaled lo plane of the filter as in avsr_gemm_h16_nt */
int avsr_conv2d_h16(const void* x, const void* wp, const void* wp_lo, int ldw, void* y, void* y2, const void* zero_page, int N, int H, int W, int Cin,
                    int Cout, int KH, int KW, int stride, int pad_h, int pad_w, avsr_stream_t stream);
int avsr_head_bias_fwd_h16(const void* x, int64_t ldx, const float* b1, const float* b2, void* o1, void* o2, void* t1, void* t2,
                           int64_t rows, int cols, avsr_stream_t stream);
/* arguments as avsr_attention_fwd2; q / k / v / pos / out are f16, out2 (may be NULL) the bf16 twin of out */
int avsr_attention_fwd_h16(const void* qu, const void* qv, const void* k, const void* v, const void* pos, const uint8_t* mask,
                           int64_t mask_sb, int64_t mask_sq, void* out, void* out2, float* lse, int B, int H, int Tq, int Tk,
                           int dk, int ldq, int ldk, int ldv, int ldp, int ldo, int64_t sbq, int64_t sbk, int64_t sbv,
                           int64_t sbo, float scale, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                           avsr_stream_t stream);
int avsr_dwconv_fwd_h16(const void* x, const float* w, const float* bias, void* y, void* y2, int B, int T, int C, int K,
                        int glu_in, avsr_stream_t stream);
int avsr_bn_act_fwd_h16(const void* x, const void* add, const float* mean, const float* invstd, const float* gamma,
                        const float* beta, void* y, void* y2, int64_t rows, int C, int act, avsr_stream_t stream);
int avsr_bn_small_fwd_h16(const void* x, int64_t rows, int C, const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, int act, void* y, void* y2,
                          float* mean, float* invstd, avsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AVSR_HIP_H */
