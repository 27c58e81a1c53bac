// loss.hip -- the joint CTC / attention loss heads and the decoder input embedding.
//
//   CTC          ctc.py:32-38,54-63: log_softmax over V -> torch.nn.CTCLoss(blank=0, reduction="sum",
//                zero_infinity=True) -> / B.  Here: row log-sum-exp (wave per row), gather of the 2L+1
//                extended-label log-probs, alpha and beta recursions (one 64-lane wave per utterance and
//                direction, states in registers, neighbours by wave shuffle -- no barriers in the T loop),
//                then the dense gradient  softmax - occupancy  written once.
//   label-smoothing CE + token accuracy
//                label_smoothing_loss.py:41-63 (KLDiv against conf/eps-smoothed one-hot, sum / B) and
//                nets_utils.py:272-292 (argmax accuracy); one block per target row, one pass for the
//                statistics and one for the gradient; the smoothed target is never materialised.
//   embedding    transformer_decoder.py:186-189 + embedding.py:78-87: table[id]*sqrt(d) + pe[pos], dropout.
//
// Gradients of the heads are produced in the forward pass, unscaled; the upstream scalar is applied by the
// consuming GEMMs through their device-side alpha (avsr_gemm alpha_dev).
#include "prims.h"
#include "avsr_hip.h"

namespace {

constexpr float LOG_ZERO = -1e30f;
constexpr int CTC_MAXSPL = 8;  // states per lane -> S <= 512, L <= 255
constexpr int CTC_PF = 6;      // frames of emissions kept in flight ahead of the alpha / beta recursion

AVSR_DEV float log_add(float a, float b) {
    const float m = fmaxf(a, b);
    if (m <= LOG_ZERO) return LOG_ZERO;
    return m + logf(avsr_exp(a - m) + avsr_exp(b - m));
}
// log(e^a + e^b + e^c) in one go for the alpha / beta recursions (one wave walks T dependent frames: the instruction count of a
// step IS the kernel time).  The argument of the logarithm lies in [1, 3]: the hardware log2 (v_log_f32, 1 ulp) is exact
// enough there -- absolute error ~1e-7 per frame on log-likelihoods of order 1e2.
AVSR_DEV float log_add3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m <= LOG_ZERO) return LOG_ZERO;
    const float sum = avsr_exp(a - m) + avsr_exp(b - m) + avsr_exp(c - m);
#ifdef AVSR_EMU
    return m + logf(sum);
#else
    return m + __logf(sum);
#endif
}

// ---- row-wise log-sum-exp over the first V columns: one wave per row
template <class T>
__global__ __launch_bounds__(256) void row_lse_kernel(const T* __restrict__ x, long ld, float* __restrict__ lse,
                                                      long rows, int V) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * ld;
    float m = LOG_ZERO, s = 0.f;
    const int nvec = V >> 3;
    for (int c = lane; c < nvec; c += 64) {
        float v[8];
        load8(xr + c * 8, v);
        float mx = v[0];
#pragma unroll
        for (int e = 1; e < 8; e++) mx = fmaxf(mx, v[e]);
        const float mn = fmaxf(m, mx);
        float acc = s * avsr_exp(m - mn);
#pragma unroll
        for (int e = 0; e < 8; e++) acc += avsr_exp(v[e] - mn);
        s = acc;
        m = mn;
    }
    for (int c = nvec * 8 + lane; c < V; c += 64) {
        const float v = Elem<T>::ld(xr + c);
        const float mn = fmaxf(m, v);
        s = s * avsr_exp(m - mn) + avsr_exp(v - mn);
        m = mn;
    }
    const float gm = wave_max(m);
    s = wave_sum(s * avsr_exp(m - gm));
    if (lane == 0) lse[row] = gm + logf(s);
}

// out[r, v] = x[r, v] - lse[r]
__global__ __launch_bounds__(256) void sub_row_scalar_kernel(const float* __restrict__ x, long ld,
                                                             const float* __restrict__ lse, float* __restrict__ out,
                                                             long rows, int V) {
    const long r = blockIdx.x;
    const float l = lse[r];
    for (int v = threadIdx.x; v < V; v += 256) out[r * ld + v] = x[r * ld + v] - l;
}

// ---- per utterance: strip ignore_id from the padded label row, build the blank-interleaved sequence
__global__ void ctc_prepare_kernel(const int64_t* __restrict__ labels, int Lmax, int ignore_id, int* __restrict__ ext,
                                   int Smax, int* __restrict__ lens /* [B]: L_b */) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    int L = 0;
    int* e = ext + (long)b * Smax;
    e[0] = 0;
    for (int i = 0; i < Lmax; i++) {
        const int64_t y = labels[(long)b * Lmax + i];
        if (y == ignore_id) continue;
        e[2 * L + 1] = (int)y;
        e[2 * L + 2] = 0;
        L++;
    }
    lens[b] = L;
}

// ---- lpg[b,t,s] = logit[b,t,ext[b,s]] - lse[b,t]
template <class T>
__global__ __launch_bounds__(256) void ctc_gather_kernel(const T* __restrict__ logits, long ld,
                                                         const float* __restrict__ lse, const int* __restrict__ ext,
                                                         const int* __restrict__ lens, float* __restrict__ lpg, int Tlen,
                                                         int Smax) {
    const long bt = blockIdx.x;
    const int b = (int)(bt / Tlen);
    const int S = 2 * lens[b] + 1;
    const float l = lse[bt];
    for (int s = threadIdx.x; s < S; s += 256)
        lpg[bt * Smax + s] = Elem<T>::ld(logits + bt * ld + ext[(long)b * Smax + s]) - l;
}

// ---- alpha (blockIdx.y == 0) / beta (== 1) recursions: one wave per utterance, lane owns SPL contiguous states.
// The recursion is a chain of T_b dependent steps executed by ONE wave: its instruction count per step IS the kernel time.
// SPL is a template parameter (from the batch's widest label row), so the per-state loops are exact and branch-free --
// states beyond an utterance's S = 2L + 1 are carried as LOG_ZERO -- and the emissions of the next CTC_PF frames, which do
// not depend on the recursion, are kept in flight in registers (a load inside the step would put a memory round trip on
// every link of the chain).
template <int SPL>
__global__ __launch_bounds__(64) void ctc_alphabeta_kernel(const float* __restrict__ lpg, const int* __restrict__ ext,
                                                           const int* __restrict__ lens,
                                                           const int64_t* __restrict__ in_lens, float* __restrict__ alpha,
                                                           float* __restrict__ beta, float* __restrict__ nll, int Tlen,
                                                           int Smax) {
    const int b = blockIdx.x, dir = blockIdx.y, lane = threadIdx.x;
    const int L = lens[b], S = 2 * L + 1;
    int Tb = (int)in_lens[b];
    if (Tb > Tlen) Tb = Tlen;
    const int* e = ext + (long)b * Smax;
    const float* lp = lpg + (long)b * Tlen * Smax;
    float* out = (dir == 0 ? alpha : beta) + (long)b * Tlen * Smax;
    const int s0 = lane * SPL;
    // can state s take the skip transition from s-2 (alpha) / to s+2 (beta)?
    bool skip[SPL], valid[SPL];
    float cur[SPL];
#pragma unroll
    for (int i = 0; i < SPL; i++) {
        const int s = s0 + i;
        valid[i] = s < S;
        skip[i] = false;
        cur[i] = LOG_ZERO;
        if (valid[i]) {
            if (dir == 0) skip[i] = (s >= 2) && (e[s] != 0) && (e[s] != e[s - 2]);
            else skip[i] = (s + 2 < S) && (e[s] != 0) && (e[s] != e[s + 2]);
        }
    }
    if (Tb <= 0) {
        if (dir == 0 && lane == 0) nll[b] = (L == 0) ? 0.f : INFINITY;
        return;
    }
    // t = first step
    const int tfirst = dir == 0 ? 0 : Tb - 1;
#pragma unroll
    for (int i = 0; i < SPL; i++) {
        const int s = s0 + i;
        if (valid[i]) {
            const bool start = dir == 0 ? (s < 2) : (s >= S - 2);
            cur[i] = start ? lp[(long)tfirst * Smax + s] : LOG_ZERO;
            out[(long)tfirst * Smax + s] = cur[i];
        }
    }
    float pre[CTC_PF][SPL];
    auto fetch = [&](int step, float (&dst)[SPL]) {
        const int t = dir == 0 ? step : Tb - 1 - step;
#pragma unroll
        for (int i = 0; i < SPL; i++) dst[i] = (step < Tb && valid[i]) ? lp[(long)t * Smax + s0 + i] : 0.f;
    };
#pragma unroll
    for (int j = 0; j < CTC_PF; j++) fetch(1 + j, pre[j]);
    for (int step0 = 1; step0 < Tb; step0 += CTC_PF) {
#pragma unroll
        for (int j = 0; j < CTC_PF; j++) {
            const int step = step0 + j;
            if (step >= Tb) break;
            const int t = dir == 0 ? step : Tb - 1 - step;
            // values owned by the neighbouring lane: n1 = the state one step away, n2 = two steps away
            float n1, n2;
            if (dir == 0) {
                n1 = wave_up1(cur[SPL - 1], LOG_ZERO);
                n2 = SPL >= 2 ? wave_up1(cur[SPL >= 2 ? SPL - 2 : 0], LOG_ZERO) : wave_up1(n1, LOG_ZERO);
            } else {
                n1 = wave_down1(cur[0], LOG_ZERO);
                n2 = SPL >= 2 ? wave_down1(cur[SPL >= 2 ? 1 : 0], LOG_ZERO) : wave_down1(n1, LOG_ZERO);
            }
            float nxt[SPL];
#pragma unroll
            for (int i = 0; i < SPL; i++) {
                float a1, a2;
                if (dir == 0) {
                    a1 = i >= 1 ? cur[i >= 1 ? i - 1 : 0] : n1;
                    a2 = i >= 2 ? cur[i >= 2 ? i - 2 : 0] : (i == 1 ? n1 : n2);
                } else {
                    a1 = (i + 1 < SPL) ? cur[(i + 1 < SPL) ? i + 1 : 0] : n1;
                    a2 = (i + 2 < SPL) ? cur[(i + 2 < SPL) ? i + 2 : 0] : ((i + 1 < SPL) ? n1 : n2);
                }
                float v = log_add3(cur[i], a1, skip[i] ? a2 : LOG_ZERO) + pre[j][i];
                v = (valid[i] && v > LOG_ZERO) ? v : LOG_ZERO;
                nxt[i] = v;
                if (valid[i]) out[(long)t * Smax + s0 + i] = v;
            }
#pragma unroll
            for (int i = 0; i < SPL; i++) cur[i] = nxt[i];
            fetch(step + CTC_PF, pre[j]);
        }
    }
    if (dir == 0) {
        // log P = logaddexp(alpha_{Tb-1}(S-1), alpha_{Tb-1}(S-2)); gather from the owning lanes
        float mine = LOG_ZERO;
#pragma unroll
        for (int i = 0; i < SPL; i++) {
            const int s = s0 + i;
            if (valid[i] && (s == S - 1 || s == S - 2)) mine = log_add(mine, cur[i]);
        }
        // combine across lanes (at most two lanes hold a contribution)
        float tot = mine;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) tot = log_add(tot, __shfl_xor(tot, m));
        if (lane == 0) nll[b] = tot <= LOG_ZERO * 0.5f ? INFINITY : -tot;
    }
}

// ---- dense gradient: g[b,t,v] = softmax[b,t,v] - occupancy[b,t,v]   (zero for t >= T_b or infeasible targets)
template <class T>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const T* __restrict__ logits, long ld,
                                                       const float* __restrict__ lse, const float* __restrict__ lpg,
                                                       const float* __restrict__ alpha, const float* __restrict__ beta,
                                                       const float* __restrict__ nll, const int* __restrict__ ext,
                                                       const int* __restrict__ lens, const int64_t* __restrict__ in_lens,
                                                       T* __restrict__ grad, long ldg, int Tlen, int V, int Smax) {
    AVSR_DYN_SMEM(smem);
    float* occ = reinterpret_cast<float*>(smem);  // [V]
    const long bt = blockIdx.x;
    const int b = (int)(bt / Tlen), t = (int)(bt % Tlen);
    T* g = grad + bt * ldg;
    const float nl = nll[b];
    const bool live = t < (int)in_lens[b] && nl < INFINITY;  // zero_infinity=True
    // (round 6: columns [V, ldg) -- the pitch padding the data-gradient GEMM contracts over -- are written here as zeros: no
    // fill launch over the whole buffer)
    if (!live) {
        for (int v = threadIdx.x; v < (int)ldg; v += 256) Elem<T>::st(g + v, 0.f);
        return;
    }
    // occupancy per extended state into LDS, then summed per label in a FIXED order (round 6: LDS float atomics from the states
    // that share a label -- every blank, repeated characters -- committed in a run-dependent order): thread 0 adds the blank
    // states (even s), thread 64 the label states (odd s), each walking its states in sequence; S <= 2 L + 1 is ~130
    float* val = occ + V;  // [Smax]
    for (int v = threadIdx.x; v < V; v += 256) occ[v] = 0.f;
    const int S = 2 * lens[b] + 1;
    for (int s = threadIdx.x; s < S; s += 256) {
        const long o = bt * Smax + s;
        const float lo = alpha[o] + beta[o] - lpg[o] + nl;  // log occupancy of state s at time t
        val[s] = lo > -80.f ? avsr_exp(lo) : 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0 || threadIdx.x == 64) {
        const int* e = ext + (long)b * Smax;
        for (int s = threadIdx.x ? 1 : 0; s < S; s += 2) occ[e[s]] += val[s];
    }
    __syncthreads();
    const float l = lse[bt];
    const T* x = logits + bt * ld;
    for (int v = threadIdx.x; v < (int)ldg; v += 256) Elem<T>::st(g + v, v < V ? avsr_exp(Elem<T>::ld(x + v) - l) - occ[v] : 0.f);
}

// ---- label-smoothing CE: one block per row
template <class T>
__global__ __launch_bounds__(256) void ce_smooth_kernel(const T* __restrict__ logits, long ld,
                                                        const int64_t* __restrict__ target, int ignore_id, int V,
                                                        float smoothing, float* __restrict__ row_loss,
                                                        float* __restrict__ row_hit, T* __restrict__ grad, long ldg) {
    __shared__ float red[4][4];
    __shared__ int redi[4];
    const long r = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* x = logits + r * ld;
    const int64_t y = target[r];
    float m = LOG_ZERO, s = 0.f, sum = 0.f;
    int am = 0;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float z = Elem<T>::ld(x + v);
        sum += z;
        if (z > m) {  // strict: keeps the first maximum within this thread's (ascending) stride
            s = s * avsr_exp(m - z) + 1.f;
            m = z;
            am = v;
        } else {
            s += avsr_exp(z - m);
        }
    }
    // wave then block reduction of (max, argmax-with-smallest-index, scaled sum, plain sum)
    float gm = wave_max(m);
    s = wave_sum(s * avsr_exp(m - gm));
    sum = wave_sum(sum);
    int cand = (m == gm) ? am : 0x7fffffff;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) cand = min(cand, __shfl_xor(cand, k));
    if (lane == 0) {
        red[wave][0] = gm;
        red[wave][1] = s;
        red[wave][2] = sum;
        redi[wave] = cand;
    }
    __syncthreads();
    float bm = red[0][0];
    for (int w = 1; w < 4; w++) bm = fmaxf(bm, red[w][0]);
    float bs = 0.f, bsum = 0.f;
    int bam = 0x7fffffff;
    for (int w = 0; w < 4; w++) {
        bs += red[w][1] * avsr_exp(red[w][0] - bm);
        bsum += red[w][2];
        if (red[w][0] == bm) bam = min(bam, redi[w]);
    }
    const float lse = bm + logf(bs);
    const bool ignored = (y == ignore_id);
    const float conf = 1.f - smoothing, eps = smoothing / (float)(V - 1);
    if (threadIdx.x == 0) {
        if (ignored) {
            row_loss[r] = 0.f;
            row_hit[r] = 0.f;
        } else {
            const float lpy = Elem<T>::ld(x + y) - lse;
            const float sum_lp = bsum - (float)V * lse;
            float c = 0.f;  // sum td*log(td) with 0*log0 = 0
            if (eps > 0.f) c += (float)(V - 1) * eps * logf(eps);
            if (conf > 0.f) c += conf * logf(conf);
            row_loss[r] = c - eps * (sum_lp - lpy) - conf * lpy;
            row_hit[r] = (bam == (int)y) ? 1.f : 0.f;
        }
    }
    if (grad) {
        T* g = grad + r * ldg;
        for (int v = threadIdx.x; v < (int)ldg; v += 256) {  // (columns [V, ldg): zeros, as in ctc_grad_kernel)
            float gv = 0.f;
            if (!ignored && v < V) gv = avsr_exp(Elem<T>::ld(x + v) - lse) - (v == (int)y ? conf : eps);
            Elem<T>::st(g + v, gv);
        }
    }
}

// out[0] = sum(a[0..n)) * scale ; generic small reduction (one block)
__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ a, int n, float scale,
                                                        float* __restrict__ out, int finite_only) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = a[i];
        if (!finite_only || (v < INFINITY && v > -INFINITY)) s += v;  // zero_infinity=True (ctc.py:26-28)
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) * scale;
}

// ---- attention-branch targets (add_sos_eos.py:12-31 + mask.py:11-37) for a [B, L] label matrix padded with ignore_id, with
// the static width L + 1: labels compacted to the left, ys_in = <sos> y <eos ...>, ys_out = y <eos> <ignore ...>,
// mask[b,i,j] = (ys_in[b,j] != ignore_id) & (j <= i), n_tokens = #(ys_out != ignore_id).  One block per utterance; the
// torch formulation of the same (sort / gather / where / tril ...) is ~30 launches of a few hundred bytes each.
__global__ __launch_bounds__(256) void prepare_targets_kernel(const int64_t* __restrict__ ys_pad, int B, int L, int64_t sos,
                                                              int64_t eos, int64_t ignore_id, int64_t* __restrict__ ys_in,
                                                              int64_t* __restrict__ ys_out, uint8_t* __restrict__ mask,
                                                              int64_t* __restrict__ n_tokens) {
    AVSR_DYN_SMEM(smem);
    int64_t* comp = reinterpret_cast<int64_t*>(smem);  // [L] compacted labels | [L] the raw row | [L + 1] "ys_in != ignore" bytes
    __shared__ int n_keep;
    const int b = blockIdx.x, W = L + 1;
    const int64_t* y = ys_pad + (long)b * L;
    uint8_t* ok = reinterpret_cast<uint8_t*>(comp + 2 * L);
    int64_t* raw = comp + L;  // the row, staged with coalesced loads (a serial walk over global memory is a chain of round trips)
    for (int j = threadIdx.x; j < L; j += 256) raw[j] = y[j];
    __syncthreads();
    if (threadIdx.x == 0) {  // L is a few hundred at most: a serial stable compaction out of LDS is a microsecond
        int n = 0;
        for (int j = 0; j < L; j++)
            if (raw[j] != ignore_id) comp[n++] = raw[j];
        n_keep = n;
    }
    __syncthreads();
    const int n = n_keep;
    for (int j = threadIdx.x; j < W; j += 256) {
        const int64_t vin = j == 0 ? sos : (j <= n ? comp[j - 1] : eos);
        const int64_t vout = j < n ? comp[j] : (j == n ? eos : ignore_id);
        ys_in[(long)b * W + j] = vin;
        ys_out[(long)b * W + j] = vout;
        ok[j] = vin != ignore_id;
    }
    __syncthreads();
    if (mask)
        for (int id = threadIdx.x; id < W * W; id += 256) {
            const int i = id / W, j = id - i * W;
            mask[(long)b * W * W + id] = (uint8_t)(ok[j] && j <= i);
        }
    if (b == 0 && n_tokens) {  // block 0 also counts the scored tokens of the whole batch: labels + one <eos> per utterance
        __shared__ int red[4];
        int cnt = 0;
        for (long i = threadIdx.x; i < (long)B * L; i += 256) cnt += ys_pad[i] != ignore_id;
        if (threadIdx.x == 0 && eos != ignore_id) cnt += B;  // one <eos> per utterance
        cnt = (int)wave_sum((float)cnt);  // exact: counts are far below 2^24
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) n_tokens[0] = (int64_t)(red[0] + red[1] + red[2] + red[3]);
    }
}

// ---- decoder input embedding: out[r,:] = table[id[r],:]*scale + pe[r % L,:], inverted dropout
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                        const float* __restrict__ pe, float* __restrict__ out, long rows,
                                                        int L, int D, float scale, float p, uint64_t seed0,
                                                        const uint64_t* seed_dev) {
    const uint64_t seed = seed0 + (seed_dev ? *seed_dev : 0ull);
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const int dv = D >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * dv; i += (long)gridDim.x * 256) {
        const long r = i / dv;
        const int c = (int)(i % dv) * 8;
        float e[8], q[8], o[8];
        load8(table + ids[r] * D + c, e);
        load8(pe + (r % L) * D + c, q);
#pragma unroll
        for (int k = 0; k < 8; k++)
            o[k] = (e[k] * scale + q[k]) * dropout_scale(seed, (uint64_t)(r * D + c + k), p, inv_keep);
        store8(out + r * D + c, o);
    }
}
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                                        float* __restrict__ dtable, long rows, int D, float scale,
                                                        float p, uint64_t seed0, const uint64_t* seed_dev) {
    const uint64_t seed = seed0 + (seed_dev ? *seed_dev : 0ull);
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    // rows that carry the same token add into the same table row: the FIRST such row sums all of them in row order and is the only
    // writer (round 6: was one float atomic per element, committed in a run-dependent order); rows is a few hundred
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * D; i += (long)gridDim.x * 256) {
        const long r = i / D;
        const int c = (int)(i % D);
        const int64_t id = ids[r];
        bool first = true;
        for (long q = 0; q < r; q++)
            if (ids[q] == id) {
                first = false;
                break;
            }
        if (!first) continue;
        float g = 0.f;
        for (long q = r; q < rows; q++)
            if (ids[q] == id) g += dout[q * D + c] * scale * dropout_scale(seed, (uint64_t)(q * D + c), p, inv_keep);
        dtable[id * D + c] += g;
    }
}

}  // namespace

extern "C" int avsr_row_lse(const void* x, int dtype, int64_t ld, float* lse, int64_t rows, int V, hipStream_t stream) {
    AVSR_REQUIRE(ld % 8 == 0, "row_lse: ld must be a multiple of 8");
    if (rows <= 0) return 0;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == 0) AVSR_LAUNCH((row_lse_kernel<float>), grid, block, 0, stream, (const float*)x, (long)ld, lse, (long)rows, V);
    else AVSR_LAUNCH((row_lse_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)x, (long)ld, lse, (long)rows, V);
    AVSR_CHECK_LAUNCH("row_lse");
    return 0;
}

// Workspace layout (f32 words): lse[B*T] | lpg[B*T*Smax] | alpha[B*T*Smax] | beta[B*T*Smax] ; ints: ext[B*Smax] | lens[B]
extern "C" int64_t avsr_ctc_workspace_bytes(int B, int T, int Lmax) {
    const int64_t Smax = 2 * (int64_t)Lmax + 1;
    return ((int64_t)B * T * (1 + 3 * Smax)) * 4 + ((int64_t)B * Smax + B) * 4 + 64;
}

// nll[b] = -log p(labels_b | logits_b) (+inf kept; the caller's sum applies zero_infinity), grad = d nll / d logits
extern "C" int avsr_ctc_loss(const void* logits, int dtype, int64_t ld, const int64_t* labels, int Lmax, int ignore_id,
                             const int64_t* in_lens, float* nll, void* grad, int64_t ldg, void* workspace, int B, int T,
                             int V, hipStream_t stream) {
    AVSR_REQUIRE(Lmax <= 255, "ctc: at most 255 labels per utterance");
    AVSR_REQUIRE(ld % 8 == 0 && ldg % 8 == 0, "ctc: ld must be a multiple of 8");
    if (B <= 0 || T <= 0) return 0;
    const int Smax = 2 * Lmax + 1;
    float* lse = reinterpret_cast<float*>(workspace);
    float* lpg = lse + (long)B * T;
    float* alpha = lpg + (long)B * T * Smax;
    float* beta = alpha + (long)B * T * Smax;
    int* ext = reinterpret_cast<int*>(beta + (long)B * T * Smax);
    int* lens = ext + (long)B * Smax;
    int rc = avsr_row_lse(logits, dtype, ld, lse, (int64_t)B * T, V, stream);
    if (rc) return rc;
    AVSR_LAUNCH(ctc_prepare_kernel, dim3(B), dim3(64), 0, stream, labels, Lmax, ignore_id, ext, Smax, lens);
    if (dtype == 0)
        AVSR_LAUNCH((ctc_gather_kernel<float>), dim3(B * T), dim3(256), 0, stream, (const float*)logits, (long)ld, lse, ext, lens, lpg, T, Smax);
    else
        AVSR_LAUNCH((ctc_gather_kernel<bf16_t>), dim3(B * T), dim3(256), 0, stream, (const bf16_t*)logits, (long)ld, lse, ext, lens, lpg, T, Smax);
    {
        const int spl = (Smax + 63) / 64;  // states per lane for the widest row of the batch (Lmax <= 255 -> at most 8)
#define AVSR_CTC_AB(N) AVSR_LAUNCH(ctc_alphabeta_kernel<N>, dim3(B, 2), dim3(64), 0, stream, lpg, ext, lens, in_lens, alpha, beta, nll, T, Smax)
        if (spl <= 1) AVSR_CTC_AB(1);
        else if (spl == 2) AVSR_CTC_AB(2);
        else if (spl == 3) AVSR_CTC_AB(3);
        else if (spl == 4) AVSR_CTC_AB(4);
        else if (spl <= 6) AVSR_CTC_AB(6);
        else AVSR_CTC_AB(8);
#undef AVSR_CTC_AB
    }
    if (grad) {
        const size_t sm = ((size_t)V + Smax) * sizeof(float);  // label occupancies + per-state values
        if (dtype == 0)
            AVSR_LAUNCH((ctc_grad_kernel<float>), dim3(B * T), dim3(256), sm, stream, (const float*)logits, (long)ld, lse, lpg, alpha, beta,
                        nll, ext, lens, in_lens, (float*)grad, (long)ldg, T, V, Smax);
        else
            AVSR_LAUNCH((ctc_grad_kernel<bf16_t>), dim3(B * T), dim3(256), sm, stream, (const bf16_t*)logits, (long)ld, lse, lpg, alpha, beta,
                        nll, ext, lens, in_lens, (bf16_t*)grad, (long)ldg, T, V, Smax);
    }
    AVSR_CHECK_LAUNCH("ctc_loss");
    return 0;
}

extern "C" int avsr_ce_smooth(const void* logits, int dtype, int64_t ld, const int64_t* target, int ignore_id, int V,
                              float smoothing, float* row_loss, float* row_hit, void* grad, int64_t ldg, int64_t rows,
                              hipStream_t stream) {
    if (rows <= 0) return 0;
    if (dtype == 0)
        AVSR_LAUNCH((ce_smooth_kernel<float>), dim3((unsigned)rows), dim3(256), 0, stream, (const float*)logits, (long)ld, target, ignore_id, V,
                    smoothing, row_loss, row_hit, (float*)grad, (long)ldg);
    else
        AVSR_LAUNCH((ce_smooth_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, stream, (const bf16_t*)logits, (long)ld, target, ignore_id, V,
                    smoothing, row_loss, row_hit, (bf16_t*)grad, (long)ldg);
    AVSR_CHECK_LAUNCH("ce_smooth");
    return 0;
}

extern "C" int avsr_sum_scale(const float* a, int n, float scale, float* out, int finite_only, hipStream_t stream) {
    AVSR_LAUNCH(sum_scale_kernel, dim3(1), dim3(256), 0, stream, a, n, scale, out, finite_only);
    AVSR_CHECK_LAUNCH("sum_scale");
    return 0;
}

extern "C" int avsr_prepare_targets(const int64_t* ys_pad, int B, int L, int64_t sos, int64_t eos, int64_t ignore_id,
                                    int64_t* ys_in, int64_t* ys_out, uint8_t* mask, int64_t* n_tokens, hipStream_t stream) {
    AVSR_REQUIRE(L >= 1 && L <= 4096, "prepare_targets: label width out of range");
    if (B <= 0) return 0;
    const size_t lds = (size_t)2 * L * sizeof(int64_t) + ((L + 1 + 7) / 8) * 8;
    AVSR_LAUNCH(prepare_targets_kernel, dim3(B), dim3(256), lds, stream, ys_pad, B, L, sos, eos, ignore_id, ys_in, ys_out, mask, n_tokens);
    AVSR_CHECK_LAUNCH("prepare_targets");
    return 0;
}

extern "C" int avsr_embed_fwd(const int64_t* ids, const float* table, const float* pe, float* out, int64_t rows, int L,
                              int D, float scale, float drop_p, uint64_t seed, const uint64_t* seed_dev, hipStream_t stream) {
    AVSR_REQUIRE(D % 8 == 0, "embed: D must be a multiple of 8");
    if (rows <= 0) return 0;
    long nb = (rows * (D >> 3) + 255) / 256;
    AVSR_LAUNCH(embed_fwd_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, stream, ids, table, pe, out, (long)rows, L, D, scale,
                drop_p, seed, seed_dev);
    AVSR_CHECK_LAUNCH("embed_fwd");
    return 0;
}

extern "C" int avsr_embed_bwd(const int64_t* ids, const float* dout, float* dtable, int64_t rows, int D, float scale,
                              float drop_p, uint64_t seed, const uint64_t* seed_dev, hipStream_t stream) {
    if (rows <= 0) return 0;
    long nb = (rows * D + 255) / 256;
    AVSR_LAUNCH(embed_bwd_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, stream, ids, dout, dtable, (long)rows, D, scale, drop_p,
                seed, seed_dev);
    AVSR_CHECK_LAUNCH("embed_bwd");
    return 0;
}

extern "C" int avsr_log_softmax(const float* x, int64_t ld, float* lse_ws, float* out, int64_t rows, int V,
                                hipStream_t stream) {
    if (rows <= 0) return 0;
    int rc = avsr_row_lse(x, 0, ld, lse_ws, rows, V, stream);
    if (rc) return rc;
    AVSR_LAUNCH(sub_row_scalar_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, (long)ld, lse_ws, out, (long)rows, V);
    AVSR_CHECK_LAUNCH("log_softmax");
    return 0;
}
