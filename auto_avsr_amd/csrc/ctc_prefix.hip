// ctc_prefix.hip -- CTC prefix scores for beam search (hybrid CTC/attention decoding).
//
// Replaces the python `for t in range(start, end)` recursion of the reference's vectorised prefix scorer
// (espnet/nets/ctc_prefix_score.py:155-160 inside CTCPrefixScoreTH.__call__, :71-187; Watanabe et al. Algorithm 2):
// for every running hypothesis n and every candidate next token c = cand[n][s]
//     phi[t]   = r_prev[t][1][n]                                   if c == last token of n
//              = logaddexp(r_prev[t][0][n], r_prev[t][1][n])       otherwise
//     r[t][0]  = logaddexp(r[t-1][0], phi[t-1]) + logp[t][c]       (prefix ends in c at frame t)
//     r[t][1]  = logaddexp(r[t-1][0], r[t-1][1]) + logp[t][blank]  (prefix followed by blank)
//     psi      = logsumexp( r[start-1][0],  phi[max(t-1,0)] + logp[t][c]  for t in [start, T) )
// with start = max(len(prefix), 1) and r[0][0] = logp[0][c] for the empty prefix.  One thread per (n, s) walks the T
// frames with its state in registers (the recursion is sequential in t, embarrassingly parallel in (n, s): beam 40 x
// 60 candidates = 2400 independent chains); r is written out as the next step's state.
#include <math.h>
#include "prims.h"
#include "avsr_hip.h"

namespace {

constexpr float LOGZERO = -10000000000.0f;

// log(e^a + e^b) on the hardware exp / log (v_exp_f32, v_log_f32): the recursion below is ONE dependent chain of these per
// frame and thread, so its instruction count is the kernel's run time (log1pf alone is ~40 dependent instructions); the
// absolute error of log(1 + x) through 1 + x is <= 6e-8, the forward variables are sums of O(1) log-probabilities
AVSR_DEV float logaddexp2(float a, float b) {
    const float m = fmaxf(a, b), d = -fabsf(a - b);
#ifdef AVSR_EMU
    return m + log1pf(avsr_exp(d));
#else
    return m + __logf(1.0f + __expf(d));
#endif
}

__global__ __launch_bounds__(64) void ctc_prefix_kernel(const float* __restrict__ logp, int T, int ldv,
                                                         const float* __restrict__ r_prev, const int64_t* __restrict__ last,
                                                         const int64_t* __restrict__ cand, int NH, int S, int out_len,
                                                         int blank, float* __restrict__ r_new, float* __restrict__ psi,
                                                         float* __restrict__ psi_eos) {
    const int id = blockIdx.x * 64 + threadIdx.x;  // (one wave per block: 2 400 chains spread over 38 CUs instead of 10)
    if (id >= NH * S) return;
    const int n = id / S, s = id - n * S;
    const int c = (int)cand[id];
    const bool same = c == (int)last[n];
    const int start = out_len > 1 ? out_len : 1;
    const size_t st = (size_t)2 * NH;  // floats between consecutive frames of r_prev
    const size_t so = (size_t)2 * NH * S;
    float rn = LOGZERO, rb = LOGZERO;  // r[t-1][0], r[t-1][1]
    float pm = LOGZERO, ps = 0.f;      // running logsumexp (max, scaled sum) of the psi terms
    auto acc = [&](float v) {
        if (v > pm) { ps = ps * avsr_exp(pm - v) + 1.f; pm = v; }
        else ps += avsr_exp(v - pm);
    };
    float phi_prev = 0.f;  // phi[t-1]
    // the recursion is sequential in t, its INPUTS are not: the loads of U frames are issued together, then the U dependent
    // updates run on registers (one frame at a time the loop was a chain of exposed L2 round trips: ~1 us per frame)
    constexpr int U = 8;
    for (int t0 = 0; t0 < T; t0 += U) {
        float p0s[U], p1s[U], xcs[U], xbs[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = min(t0 + u, T - 1);
            p0s[u] = r_prev[t * st + n];
            p1s[u] = r_prev[t * st + NH + n];
            xcs[u] = logp[(size_t)t * ldv + c];
            xbs[u] = logp[(size_t)t * ldv + blank];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t0 + u;
            if (t >= T) break;
            const float p0 = p0s[u], p1 = p1s[u];
            const float phi = same ? p1 : logaddexp2(p0, p1);
            const float xc = xcs[u], xb = xbs[u];
            float n0 = LOGZERO, n1 = LOGZERO;
            if (t == 0 && out_len == 0) n0 = xc;
            if (t >= start) {
                n0 = logaddexp2(rn, phi_prev) + xc;
                n1 = logaddexp2(rn, rb) + xb;
                acc((t == 0 ? phi : phi_prev) + xc);
            }
            if (t == start - 1) acc(n0);  // r[start-1][0]
            r_new[t * so + (size_t)n * S + s] = n0;
            r_new[t * so + (size_t)NH * S + (size_t)n * S + s] = n1;
            rn = n0;
            rb = n1;
            phi_prev = phi;
            if (s == 0 && t == T - 1) psi_eos[n] = logaddexp2(p0, p1);
        }
    }
    psi[id] = pm + logf(ps);
}

}  // namespace

// logp: [T][ldv] f32 log-softmax rows of ONE utterance; r_prev: [T][2][NH] (state of the running hypotheses: log prob of
// the prefix ending in non-blank / blank at frame t); last: [NH] last token of each prefix; cand: [NH][S] candidate next
// tokens; out_len = len(prefix) - 1 (sos excluded).  Outputs: r_new [T][2][NH][S], psi [NH][S] (log prefix probability of
// prefix + candidate), psi_eos [NH] (log probability of the prefix as a complete label sequence).
extern "C" int avsr_ctc_prefix_score(const float* logp, int T, int V, int ldv, const float* r_prev, const int64_t* last,
                                     const int64_t* cand, int NH, int S, int out_len, int blank, float* r_new, float* psi,
                                     float* psi_eos, hipStream_t stream) {
    AVSR_REQUIRE(T > 0 && V > 0 && ldv >= V && blank >= 0 && blank < V && out_len >= 0, "ctc_prefix_score: bad dimensions");
    if (NH <= 0 || S <= 0) return 0;
    AVSR_LAUNCH(ctc_prefix_kernel, dim3((NH * S + 63) / 64), dim3(64), 0, stream, logp, T, ldv, r_prev, last, cand, NH, S,
                out_len, blank, r_new, psi, psi_eos);
    AVSR_CHECK_LAUNCH("ctc_prefix_score");
    return 0;
}
