// attention_kv.hip -- the key / value side of the attention backward for bf16 operands, on k-major tiles with LDS
// transpose reads:
//     dV[b,h] = Pd[b,h]^T dO[b,h]        dK[b,h] = dS[b,h]^T Qu[b,h]        dpos[h] += sum_b skew(dS[b,h])^T Qv[b,h]
// (attention.py:59-104,131-193 under autograd; pd / ds are what avsr_attention_bwd_dq stored, [B,H,Tq,lds] bf16 with zeros
// in the pad columns [Tk, lds)).  All three are TN contractions over the QUERY index -- the slow dimension of both
// operands -- so the tiles are copied to LDS exactly as they lie in HBM (64 k-rows x 128 bytes, 16-byte register-staged
// copies, next tile in flight while the current one is multiplied) and the 32x32x16 MFMA fragments are fetched with
// ds_read_b64_tr_b16, as in gemm_tn_kernel.h.  The generic path (gemm_core.h gemm_multi_kernel) transposes element by
// element into k-contiguous LDS tiles and gathers the skewed dS operand of the position term with eight guarded scalar
// loads per chunk: 61 us per encoder layer at T = 400 against 24 us for the whole forward kernel.
//
// The position term: block (b, h, band tile) contracts batch item b's queries and ADDS into dpos[h] (f32 atomics); of the
// Tq queries only those whose key index band + q - (Tq-1) falls into [0, Tk) are visited.  Its A operand
// A[q][band] = dS[b,q][band + q - (Tq-1)] is a row of dS shifted by a per-row offset, i.e. eight elements at a 2-byte-aligned
// address (out of reach of LDS-DMA): fetched as aligned dwords and funnel-shifted in registers.
//
// pd / ds pad columns [Tk, lds) are never written by the producer: the position term masks them, the other two contractions
// only let them reach output rows >= Tk, which are not stored.
// History: the first MI355X runs of this kernel ended in a GPU memory access fault.  Round 3 isolated it contraction by
// contraction (tools/kv_fault_probe.py, avsr_tune knob 11): dV / dK were correct, the f32 atomics of the dpos epilogue
// faulted -- the compiler had left their row pitch in an undefined scalar register (see kv_block below).
#include "gemm_core.h"
#include "avsr_hip.h"

namespace {

struct KvParams {
    const bf16_t *pd, *ds, *dout, *qu, *qv;
    bf16_t *dk, *dv;
    float* dpos;
    int lds, ldo, ldq, ldk, ldv, ldpos;
    long sbo, sbq, sbk, sbv;
    int B, H, Tq, Tk, skip;
    int det;  // deterministic mode: the position term of head h is formed by the blocks of batch item 0, which walk every batch item
};


constexpr int KV_OP_BYTES = 64 * 128, KV_STAGE_BYTES = 2 * KV_OP_BYTES;
constexpr size_t KV_LDS_BYTES = 2 * KV_STAGE_BYTES;  // two stages; also covers the 64 x 68 f32 epilogue tile

// One block of contraction `which` (0: dV, 1: dK, 2: dpos).  A template parameter, not a run-time value: with `which` as a
// wave-uniform run-time variable hipcc (ROCm 7.2) left the output pitch of the which == 2 path in an UNDEFINED scalar
// register (the s_mov of p.ldpos was tail-merged into the which == 0 path; found in the ISA after the kernel's dpos atomics
// faulted on the MI355X while the host-emulator build of the same source was correct).
template <int which>
AVSR_DEV void kv_block(const KvParams& p, char* smem) {
    const int m0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int wm = wave >> 1, wn = wave & 1;
    // every contraction runs per (batch item, head); the position term's blocks ADD their batch item's share into dpos[h]
    // (f32 atomics of the accumulate epilogue; the first version gave one block all B * Tq rows of a band tile: 156 long
    // blocks were the launch's critical path, 37 us at B = 4, T = 400)
    const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
    const int M = which < 2 ? p.Tk : 2 * p.Tq - 1, K = p.Tq;
    if (m0 >= M) return;
    // (deterministic mode, which == 2: ONE block per (head, band tile) adds to dpos -- it contracts all batch items, in order)
    const int nb = (which == 2 && p.det) ? p.B : 1;
    if (which == 2 && p.det && b != 0) return;
    int t_lo = 0, nt = (K + 63) / 64;
    if (which == 2) {
        // band rows m0 .. m0+63 only meet queries q with a key index j = m + q - (Tq-1) in [0, Tk): skip the k-tiles outside
        const int q_lo = max(0, p.Tq - 1 - m0 - 63), q_hi = min(p.Tq, p.Tq - 1 - m0 + p.Tk);  // [q_lo, q_hi)
        if (q_hi <= q_lo) return;
        t_lo = q_lo / 64;
        nt = (q_hi + 63) / 64;
    }
    const bf16_t* Abase = (which == 0 ? p.pd : p.ds) + ((long)b * p.H + h) * p.Tq * p.lds;  // which < 2
    const bf16_t* Bbase = which == 0 ? p.dout + b * p.sbo + h * 64 : p.qu + b * p.sbq + h * 64;
    const int ldb = which == 0 ? p.ldo : p.ldq;

    const int kr0 = threadIdx.x >> 3, chunk = (threadIdx.x & 7) * 8;  // this thread's two k-rows (kr0, kr0 + 32), 8 columns
    bf16x8 ra[2], rb[2];
    int bcur = b;  // batch item of the tile being fetched (which == 2 in deterministic mode: 0 .. B-1)
    auto fetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int k = t * 64 + kr0 + 32 * i;
            ra[i] = rb[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (k >= K) continue;
            if (which < 2) {
                if (m0 + chunk < p.lds) ra[i] = *reinterpret_cast<const bf16x8*>(Abase + (long)k * p.lds + m0 + chunk);
                rb[i] = *reinterpret_cast<const bf16x8*>(Bbase + (long)k * ldb + chunk);
            } else {
                const int bb = bcur, q = k;
                const bf16_t* row = p.ds + (((long)bb * p.H + h) * p.Tq + q) * p.lds;
                const int j = m0 + chunk + q - (p.Tq - 1);  // key index of band column m0 + chunk for this query
                const int jb = j & ~1;  // even element index: a 4-byte-aligned address (rows start 16-byte aligned)
                if (j >= 0 && jb + 10 <= p.lds) {
                    // eight elements from a 2-byte-aligned position as five ALIGNED dwords + a 16-bit funnel shift for odd j
                    const uint32_t* w = reinterpret_cast<const uint32_t*>(row + jb);
                    uint32_t d[5];
#pragma unroll
                    for (int e = 0; e < 5; e++) d[e] = w[e];
                    uint32_t o[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = (j & 1) ? ((d[e] >> 16) | (d[e + 1] << 16)) : d[e];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        // columns [Tk, lds) of pd / ds are never written (torch.empty): mask them
                        ra[i][2 * e] = j + 2 * e < p.Tk ? (short)(o[e] & 0xffffu) : (short)0;
                        ra[i][2 * e + 1] = j + 2 * e + 1 < p.Tk ? (short)(o[e] >> 16) : (short)0;
                    }
                } else if (j > -8 && j < p.Tk) {
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        if (j + e >= 0 && j + e < p.Tk) ra[i][e] = (short)row[j + e];
                }
                rb[i] = *reinterpret_cast<const bf16x8*>(p.qv + bb * p.sbq + (long)q * p.ldq + h * 64 + chunk);
            }
        }
    };
    auto commit = [&](char* stage) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int kr = kr0 + 32 * i;
            *reinterpret_cast<bf16x8*>(stage + kr * 128 + chunk * 2) = ra[i];
            *reinterpret_cast<bf16x8*>(stage + KV_OP_BYTES + kr * 128 + chunk * 2) = rb[i];
        }
    };
    // 32 (m or n) x 16 (k) fragment of a k-major tile: columns c0 .. c0+31, k-step ks (layout as gemm_tn_kernel.h frag_async)
    auto frag = [&](const char* base, int c0, int ks) {
        const int g = lane >> 4, i = lane & 15;
        const bf16_t* t = reinterpret_cast<const bf16_t*>(base);
        const int row = ks * 16 + 8 * (g >> 1) + (i >> 2);
        const int col = c0 + 16 * (g & 1) + 4 * (i & 3);
        const bf16x4 lo = lds_tr16(t + row * 64 + col), hi = lds_tr16(t + (row + 4) * 64 + col);
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[0][0][r] = 0.f;
    // flat iteration over (batch item, k-tile): it -> (bb0 + it / span, t_lo + it % span); one batch item unless deterministic
    const int span = nt - t_lo, n_it = nb * span;
    auto fetch_it = [&](int it) {
        bcur = (nb > 1 ? it / span : b);
        fetch(t_lo + it % span);
    };
    fetch_it(0);
    commit(smem);
    __syncthreads();
    for (int it = 0; it < n_it; it++) {
        const char* As = smem + (it & 1) * KV_STAGE_BYTES;
        const char* Bs = As + KV_OP_BYTES;
        if (it + 1 < n_it) fetch_it(it + 1);  // in flight during the multiply
#pragma unroll
        for (int ks = 0; ks < 4; ks++) acc[0][0] = mfma32(frag(As, wm * 32, ks), frag(Bs, wn * 32, ks), acc[0][0]);
        if (it + 1 < n_it) commit(smem + ((it + 1) & 1) * KV_STAGE_BYTES);
        __syncthreads();
    }

    avsr_gemm_impl::Params q{};
    q.M = M; q.N = 64; q.alpha = 1.f; q.gate_scale = 1.f;
    if (which == 0) { q.C = p.dv + b * p.sbv + h * 64; q.c_dtype = 1; q.ldc = p.ldv; }
    else if (which == 1) { q.C = p.dk + b * p.sbk + h * 64; q.c_dtype = 1; q.ldc = p.ldk; }
    else { q.C = p.dpos + h * 64; q.c_dtype = 0; q.ldc = p.ldpos; q.accumulate = 1; }  // "+=" as the entry point documents
    avsr_gemm_impl::epilogue_lds<64, 64, 1, 1>(acc, q, m0, 0, wm * 32, wn * 32, 0, 0, smem);
}

__global__ __launch_bounds__(256) void attn_bwd_kv_fast_kernel(KvParams p) {
    AVSR_DYN_SMEM(smem);
    if ((p.skip >> blockIdx.z) & 1) return;  // avsr_tune knob 11: bit mask of contractions to skip (fault isolation)
    switch (blockIdx.z) {
        case 0: kv_block<0>(p, smem); break;
        case 1: kv_block<1>(p, smem); break;
        default: kv_block<2>(p, smem); break;
    }
}

}  // namespace

// bf16 fast path of avsr_attention_bwd_kv (gemm.hip); same arguments, dk_dim == 64.  Returns 0 when launched.
int avsr_attention_bwd_kv_fast(const void* pd, const void* ds, int lds, const void* dout, int ldo, int64_t sbo, const void* qu,
                               const void* qv, int ldq, int64_t sbq, void* dk, int ldk, int64_t sbk, void* dv, int ldv,
                               int64_t sbv, float* dpos, int ldpos, int B, int H, int Tq, int Tk, hipStream_t stream) {
    KvParams p{};
    p.pd = (const bf16_t*)pd; p.ds = (const bf16_t*)ds; p.dout = (const bf16_t*)dout;
    p.qu = (const bf16_t*)qu; p.qv = (const bf16_t*)qv;
    p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.dpos = dpos;
    p.lds = lds; p.ldo = ldo; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldpos = ldpos;
    p.sbo = sbo; p.sbq = sbq; p.sbk = sbk; p.sbv = sbv;
    p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.skip = avsr_tune_knobs[11];
    p.det = avsr_det() ? 1 : 0;
    const int mt_kv = (Tk + 63) / 64, mt_pos = dpos ? (2 * Tq - 1 + 63) / 64 : 0;
    dim3 grid(mt_kv > mt_pos ? mt_kv : mt_pos, B * H, dpos ? 3 : 2), block(256);
    AVSR_LAUNCH(attn_bwd_kv_fast_kernel, grid, block, KV_LDS_BYTES, stream, p);
    return 0;
}
