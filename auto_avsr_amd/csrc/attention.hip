// attention.hip -- fused (flash-style) multi-head attention for gfx950, with the
// Transformer-XL relative-position term of the Conformer encoder.
//
// Replaces the op sequence of
//   RelPositionMultiHeadedAttention.forward   attention.py:153-193  (+ rel_shift :131-151)
//   MultiHeadedAttention.forward_attention    attention.py:59-88    (mask -> softmax -> zero -> dropout -> @V)
//   MultiHeadedAttention.forward              attention.py:90-104   (decoder self / source attention)
// without ever materialising the (B,H,T,T) score or (B,H,T,2T-1) position tensors in the forward pass:
//   scores[i,j] = ( (q_i+u).k_j + (q_i+v).p[j-i+T-1] ) / sqrt(d_k)
// The rel_shift of the reference is the index map  bd[i,j] = G[i, j-i+T-1]  with G = (q+v) p^T; per
// (64-query, 64-key) tile only a 127-row band of p is needed, G is produced by MFMA into LDS and read
// back skewed.
//
// Work decomposition: grid (ceil(Tq/64), H, B); 256 threads = 4 waves, each wave owns 16 query rows and
// runs v_mfma_f32_16x16x32_bf16; online softmax with per-row statistics reduced by wave shuffles inside
// 16-lane groups; K, V^T and the position band are staged in LDS per key tile.  d_k is fixed to 64 (the
// only head size of the reference model).  NS = 2 runs every contraction on split hi/lo bf16 planes.
//
// The backward "dq" kernel recomputes the probabilities from the saved log-sum-exp, produces dQu / dQv
// and writes the (dropout-applied) probabilities and the scaled score gradients dS to HBM; dK, dV and the
// position-projection gradient are then batched TN GEMMs over those (gemm_core.h).
#include "prims.h"
#include "avsr_hip.h"

namespace {

constexpr int DK = 64;
constexpr int QT = 64;        // query rows per block
constexpr int KT = 64;        // keys per tile
constexpr int PITCH = DK + 8; // bf16 row pitch of k-contiguous LDS tiles (144 B)
constexpr int PB_ROWS = 144;  // staged position band rows (127 used, zero padded)
constexpr int G_PITCH = 84;   // f32 pitch of the per-wave G scratch (80 used)
constexpr int DG_PITCH = 104; // bf16 pitch of the per-wave skewed dS scratch (96 used)
constexpr float NEG_BIG = -1e30f;

struct AttnParams {
    const void* qu;   // [B,Tq,H,64] (q + pos_bias_u, or plain q when !RELPOS)
    const void* qv;   // [B,Tq,H,64] (q + pos_bias_v)
    const void* k;    // [B,Tk,H,64]
    const void* v;    // [B,Tk,H,64]
    const void* pos;  // [2*Tq-1, H*64] projected positional embeddings
    const uint8_t* mask;  // mask[b*mask_sb + i*mask_sq + j] != 0 means "attend"; null = no mask
    long mask_sb, mask_sq;
    void* out;   // [B,Tq,H,64]
    void* out2;  // forward, optional: bf16 twin of an f32 `out` (same strides)
    float* lse;  // [B,H,Tq]
    int B, H, Tq, Tk;
    int ldq, ldk, ldv, ldp, ldo;       // row strides (elements)
    long sbq, sbk, sbv, sbo;           // batch strides (elements)
    float scale, drop_p;
    uint64_t seed;
    const uint64_t* seed_dev;
    // backward only
    const void* dout;  // [B,Tq,H,64] strides as out
    void* dqu;         // [B,Tq,H,64] strides as qu
    void* dqv;
    void* pd;          // [B,H,Tq,lds] dropout-applied probabilities
    void* ds;          // [B,H,Tq,lds] scale * dS
    int lds;           // row pitch of pd/ds (multiple of 8, >= Tk)
    // backward, relative-position form, optional: instead of dqu / dqv write their SUM (the gradient of q itself,
    // attention.py:171-177: q_with_bias_u = q + pos_bias_u, q_with_bias_v = q + pos_bias_v) and add the column sums of the
    // two parts -- the gradients of pos_bias_u / pos_bias_v -- to du / dv [H*64] (f32, caller zeroes)
    void* dq_sum;      // [B,Tq,H,64] view, row pitch lddq, batch stride sbdq
    int lddq;
    long sbdq;
    float* du;
    float* dv;
};

AVSR_DEV void wave_sync() {
#ifdef AVSR_EMU
    size_t st;
    char z = 0;
    (void)emu::wave_gather(&z, 1, &st);
#else
    __builtin_amdgcn_wave_barrier();
#endif
}

// ---- block-cooperative staging of a [ROWS][64] k-contiguous tile: row r comes from src row (row0+r) if
// 0 <= row0+r < row_lim, else zeros.  LDS image [NS][ROWS][PITCH].
template <class T, int NS, int ROWS>
AVSR_DEV void stage_rows(bf16_t* lds, const T* src, long ld, int row0, int row_lim) {
    if (sizeof(T) == 2 && NS == 1) {
        // bf16 in, one bf16 plane out: a plain 16-byte copy (the generic path below unpacks to f32 and rounds back: ~30 VALU
        // instructions per chunk, a fifth of the instructions of the whole key-tile iteration)
        for (int id = threadIdx.x; id < ROWS * 8; id += 256) {
            const int r = id >> 3, c = (id & 7) * 8;
            const int gr = row0 + r;
            bf16x8 v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (gr >= 0 && gr < row_lim) v = *reinterpret_cast<const bf16x8*>(src + (long)gr * ld + c);
            *reinterpret_cast<bf16x8*>(lds + r * PITCH + c) = v;
        }
        return;
    }
    for (int id = threadIdx.x; id < ROWS * 8; id += 256) {
        const int r = id >> 3, c = (id & 7) * 8;
        const int gr = row0 + r;
        float v[8];
        if (gr >= 0 && gr < row_lim) {
            load8(src + (long)gr * ld + c, v);
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = 0.f;
        }
        bf16x8 pl[NS];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            bf16_t s[NS];
            split_bf16<NS>(v[e], s);
#pragma unroll
            for (int p = 0; p < NS; p++) pl[p][e] = (short)s[p];
        }
#pragma unroll
        for (int p = 0; p < NS; p++)
            *reinterpret_cast<bf16x8*>(lds + (size_t)p * ROWS * PITCH + r * PITCH + c) = pl[p];
    }
}
template <int NS>
AVSR_DEV Frag<NS> ldfrag(const bf16_t* lds, int plane_elems, int pitch, int row, int koff) {
    Frag<NS> f;
#pragma unroll
    for (int p = 0; p < NS; p++)
        f.p[p] = *reinterpret_cast<const bf16x8*>(lds + (size_t)p * plane_elems + row * pitch + koff);
    return f;
}
// B fragment [n][k] of a tile stored k-major ([k][pitch], n contiguous): columns n0 + lc, k = k0 + 8*quad .. +7, as
// two ds_read_b64_tr_b16 (each 16-lane group transposes a 4x16 block; see prims.h lds_tr16)
template <int NS>
AVSR_DEV Frag<NS> ldfrag_tr(const bf16_t* lds, int plane_elems, int pitch, int k0, int n0) {
    const int lane = threadIdx.x & 63, quad = lane >> 4, lc = lane & 15;
    Frag<NS> f;
#pragma unroll
    for (int p = 0; p < NS; p++) {
        const bf16_t* p0 = lds + (size_t)p * plane_elems + (k0 + 8 * quad + (lc >> 2)) * pitch + n0 + 4 * (lc & 3);
        const bf16x4 lo = lds_tr16(p0), hi = lds_tr16(p0 + 4 * pitch);
        f.p[p] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    return f;
}
// A fragment straight from HBM: 8 consecutive elements of one row
template <class T, int NS>
AVSR_DEV Frag<NS> gfrag(const T* p, bool valid) {
    float v[8];
    if (valid) load8(p, v);
    else
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = 0.f;
    Frag<NS> f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        bf16_t s[NS];
        split_bf16<NS>(v[e], s);
#pragma unroll
        for (int q = 0; q < NS; q++) f.p[q][e] = (short)s[q];
    }
    return f;
}

AVSR_DEV float group16_max(float v) {
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
AVSR_DEV float group16_sum(float v) {
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// Register-staged prefetch of the NEXT key tile (bf16, one plane): the K / V / position-band rows of tile kt+1 are fetched
// into registers while tile kt is being multiplied and stored to LDS at the top of the next iteration -- a block pays the
// global-memory latency of its operands once, not once per key tile (synchronous staging: load, wait, store, barrier,
// compute; measured 37 -> 24 us on the T = 400 forward).
template <bool RELPOS>
struct TilePrefetch {
    static constexpr int PB_CH = RELPOS ? (PB_ROWS * 8 + 255) / 256 : 0;  // 16-byte chunks per thread
    bf16x8 rk[2], rv[2], rp[PB_CH > 0 ? PB_CH : 1];
    int tid = threadIdx.x & 255;  // thread index inside the 256-thread group that shares the staged tile
    AVSR_DEV void fetch(const bf16_t* kk, long ldk, const bf16_t* vv, long ldv, const bf16_t* pos, long ldp, int j0, int Tk,
                        int prow0, int plim) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int id = tid + 256 * c, r = id >> 3, ch = (id & 7) * 8, gr = j0 + r;
            rk[c] = rv[c] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (gr < Tk) {
                rk[c] = *reinterpret_cast<const bf16x8*>(kk + (long)gr * ldk + ch);
                rv[c] = *reinterpret_cast<const bf16x8*>(vv + (long)gr * ldv + ch);
            }
        }
        if (RELPOS) {
#pragma unroll
            for (int c = 0; c < PB_CH; c++) {
                const int id = tid + 256 * c, r = id >> 3, ch = (id & 7) * 8, gr = prow0 + r;
                rp[c] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (id < PB_ROWS * 8 && gr >= 0 && gr < plim) rp[c] = *reinterpret_cast<const bf16x8*>(pos + (long)gr * ldp + ch);
            }
        }
    }
    AVSR_DEV void commit(bf16_t* Ks, bf16_t* Vs, bf16_t* Pb) const {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int id = tid + 256 * c, r = id >> 3, ch = (id & 7) * 8;
            *reinterpret_cast<bf16x8*>(Ks + r * PITCH + ch) = rk[c];
            *reinterpret_cast<bf16x8*>(Vs + r * PITCH + ch) = rv[c];
        }
        if (RELPOS) {
#pragma unroll
            for (int c = 0; c < PB_CH; c++) {
                const int id = tid + 256 * c, r = id >> 3, ch = (id & 7) * 8;
                if (id < PB_ROWS * 8) *reinterpret_cast<bf16x8*>(Pb + r * PITCH + ch) = rp[c];
            }
        }
    }
};

template <class T, int NS, bool RELPOS, bool BWD>
struct Attn {
    // LDS carve (bf16 elements unless noted)
    static constexpr int KS_E = NS * KT * PITCH;               // K tile  [key][d]
    static constexpr int VS_E = NS * KT * PITCH;               // V tile  [key][d]
    static constexpr int PB_E = RELPOS ? NS * PB_ROWS * PITCH : 0;
    static constexpr int PS_E = 4 * NS * 16 * PITCH;           // per wave P (fwd) / dS (bwd) as A operand
    // skewed dS (bwd): one bf16 plane fits in the wave's G scratch once the scores are formed, two planes do not
    static constexpr int DG_E = (RELPOS && BWD && NS > 1) ? 4 * NS * 16 * DG_PITCH : 0;
    static_assert(16 * DG_PITCH * 2 <= 16 * G_PITCH * 4, "dS skew scratch must fit the G scratch");
    static constexpr int G_F = RELPOS ? 4 * 16 * G_PITCH : 0;  // f32
    static constexpr size_t LDS_BYTES = (size_t)(KS_E + VS_E + PB_E + PS_E + DG_E) * 2 + (size_t)G_F * 4;

    static AVSR_DEV void run(const AttnParams& p, char* smem) {
        bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
        bf16_t* Vs = Ks + KS_E;
        bf16_t* Pb = Vs + VS_E;
        bf16_t* Ps = Pb + PB_E;
        bf16_t* DG = Ps + PS_E;
        float* Gs = reinterpret_cast<float*>(DG + DG_E);

        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int quad = lane >> 4, lc = lane & 15;
        const int i0 = blockIdx.x * QT, h = blockIdx.y, b = blockIdx.z;
        const int Tq = p.Tq, Tk = p.Tk;
        const T* qu = reinterpret_cast<const T*>(p.qu) + b * p.sbq + h * DK;
        const T* qv = RELPOS ? reinterpret_cast<const T*>(p.qv) + b * p.sbq + h * DK : nullptr;
        const T* kk = reinterpret_cast<const T*>(p.k) + b * p.sbk + h * DK;
        const T* vv = reinterpret_cast<const T*>(p.v) + b * p.sbv + h * DK;
        const T* pos = RELPOS ? reinterpret_cast<const T*>(p.pos) + h * DK : nullptr;
        const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
        const uint64_t seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);

        // ---- per-wave A fragments (rows i0 + 16w + lc)
        const int arow = i0 + 16 * w + lc;
        Frag<NS> fa_u[2], fa_v[2], fa_do[2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            fa_u[ks] = gfrag<T, NS>(qu + (long)arow * p.ldq + ks * 32 + 8 * quad, arow < Tq);
            if (RELPOS) fa_v[ks] = gfrag<T, NS>(qv + (long)arow * p.ldq + ks * 32 + 8 * quad, arow < Tq);
        }
        // ---- per-row state; C-layout rows of this lane: rr = 4*quad + r
        float m_run[4], l_run[4], lse_r[4], delta[4];
        f32x4 acc0[4], acc1[4];  // fwd: acc0 = O ; bwd: acc0 = dQu, acc1 = dQv  (4 d-tiles of 16)
#pragma unroll
        for (int n = 0; n < 4; n++) {
            acc0[n] = f32x4{0, 0, 0, 0};
            acc1[n] = f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            m_run[r] = NEG_BIG;
            l_run[r] = 0.f;
            lse_r[r] = 0.f;
            delta[r] = 0.f;
        }
        if (BWD) {
            const T* dout = reinterpret_cast<const T*>(p.dout) + b * p.sbo + h * DK;
            const T* outp = reinterpret_cast<const T*>(p.out) + b * p.sbo + h * DK;
            float part = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                fa_do[ks] = gfrag<T, NS>(dout + (long)arow * p.ldo + ks * 32 + 8 * quad, arow < Tq);
                if (arow < Tq) {
                    float a[8], c[8];
                    load8(dout + (long)arow * p.ldo + ks * 32 + 8 * quad, a);
                    load8(outp + (long)arow * p.ldo + ks * 32 + 8 * quad, c);
#pragma unroll
                    for (int e = 0; e < 8; e++) part += a[e] * c[e];
                }
            }
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);  // every lane with the same lc now holds delta(row lc)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                delta[r] = __shfl(part, 4 * quad + r);
                const int ig = i0 + 16 * w + 4 * quad + r;
                lse_r[r] = ig < Tq ? p.lse[((long)b * p.H + h) * Tq + ig] : 0.f;
            }
        }

        const int ntiles = (Tk + KT - 1) / KT;
        const bool wave_active = i0 + 16 * w < Tq;
        const bool row_mask = p.mask && p.mask_sq != 0;  // false: one mask row per batch item (padding mask)
        // bf16, one plane: next tile prefetched into registers -- except in the relative-position backward kernel, where the
        // 40 extra VGPRs push the wave past 256 (288) and cost the second resident block per CU (54 -> 62 us measured)
        constexpr bool PREF = sizeof(T) == 2 && NS == 1 && !(RELPOS && BWD);
        TilePrefetch<RELPOS> tp;
        const bf16_t* kk16 = reinterpret_cast<const bf16_t*>(kk);
        const bf16_t* vv16 = reinterpret_cast<const bf16_t*>(vv);
        const bf16_t* pos16 = reinterpret_cast<const bf16_t*>(pos);
        if (PREF) tp.fetch(kk16, p.ldk, vv16, p.ldv, pos16, p.ldp, 0, Tk, -i0 + Tq - 1 - 63, 2 * Tq - 1);
        for (int kt = 0; kt < ntiles; kt++) {
            const int j0 = kt * KT;
            __syncthreads();
            if (PREF) {
                tp.commit(Ks, Vs, Pb);
            } else {
                stage_rows<T, NS, KT>(Ks, kk, p.ldk, j0, Tk);
                stage_rows<T, NS, KT>(Vs, vv, p.ldv, j0, Tk);
                if (RELPOS) stage_rows<T, NS, PB_ROWS>(Pb, pos, p.ldp, j0 - i0 + Tq - 1 - 63, 2 * Tq - 1);
            }
            __syncthreads();
            if (PREF && kt + 1 < ntiles)
                tp.fetch(kk16, p.ldk, vv16, p.ldv, pos16, p.ldp, j0 + KT, Tk, j0 + KT - i0 + Tq - 1 - 63, 2 * Tq - 1);
            if (!wave_active) continue;  // wave-uniform: this wave's 16 query rows are all past Tq

            // ---- scores: (q+u).k
            f32x4 s[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                s[j] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
                    s[j] = mma16<NS>(fa_u[ks], ldfrag<NS>(Ks, KT * PITCH, PITCH, j * 16 + lc, ks * 32 + 8 * quad), s[j]);
            }
            // ---- (q+v).p band, skewed through LDS
            const int sb = 48 - 16 * w;
            float* Gw = Gs + w * 16 * G_PITCH;
            if (RELPOS) {
#pragma unroll
                for (int t = 0; t < 5; t++) {
                    f32x4 g = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 2; ks++)
                        g = mma16<NS>(fa_v[ks],
                                      ldfrag<NS>(Pb, PB_ROWS * PITCH, PITCH, sb + t * 16 + lc, ks * 32 + 8 * quad), g);
#pragma unroll
                    for (int r = 0; r < 4; r++) Gw[(4 * quad + r) * G_PITCH + t * 16 + lc] = g[r];
                }
                wave_sync();
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int rr = 4 * quad + r;
                        s[j][r] += Gw[rr * G_PITCH + (j * 16 + lc) - rr + 15];
                    }
            }
            // ---- mask + softmax pieces
            bool valid[4][4], colok[4];
            float pr[4][4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int jg = j0 + j * 16 + lc;
                colok[j] = jg < Tk;
                if (colok[j] && p.mask && !row_mask) colok[j] = p.mask[(long)b * p.mask_sb + jg] != 0;
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int ig = i0 + 16 * w + 4 * quad + r;
                const long mrow = row_mask ? (long)b * p.mask_sb + (long)(ig < Tq ? ig : 0) * p.mask_sq : 0;
                float mx = NEG_BIG;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int jg = j0 + j * 16 + lc;
                    bool ok = colok[j];
                    if (ok && row_mask) ok = p.mask[mrow + jg] != 0;
                    valid[j][r] = ok;
                    s[j][r] = ok ? s[j][r] * p.scale : NEG_BIG;
                    mx = fmaxf(mx, s[j][r]);
                }
                if (!BWD) {
                    mx = group16_max(mx);
                    const float m_new = fmaxf(m_run[r], mx);
                    const float corr = avsr_exp(m_run[r] - m_new);
                    float rs = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        pr[j][r] = valid[j][r] ? avsr_exp(s[j][r] - m_new) : 0.f;
                        rs += pr[j][r];
                    }
                    rs = group16_sum(rs);
                    l_run[r] = l_run[r] * corr + rs;
                    m_run[r] = m_new;
#pragma unroll
                    for (int n = 0; n < 4; n++) acc0[n][r] *= corr;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) pr[j][r] = valid[j][r] ? avsr_exp(s[j][r] - lse_r[r]) : 0.f;
                }
            }
            // dropout keep-scale per element (1 when p == 0)
            float keep[4][4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int ig = i0 + 16 * w + 4 * quad + r;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int jg = j0 + j * 16 + lc;
                    keep[j][r] = dropout_scale(seed, (((uint64_t)b * p.H + h) * Tq + ig) * (uint64_t)Tk + jg,
                                               p.drop_p, inv_keep);
                }
            }
            bf16_t* Pw = Ps + (size_t)w * 16 * PITCH;  // plane stride = 4*16*PITCH
            constexpr int PS_PLANE = 4 * 16 * PITCH;
            if (!BWD) {
                // ---- P (dropout applied) -> LDS as A operand, then O += P V
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        bf16_t sp[NS];
                        split_bf16<NS>(pr[j][r] * keep[j][r], sp);
#pragma unroll
                        for (int q = 0; q < NS; q++) Pw[(size_t)q * PS_PLANE + (4 * quad + r) * PITCH + j * 16 + lc] = sp[q];
                    }
                wave_sync();
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const Frag<NS> fa = ldfrag<NS>(Pw, PS_PLANE, PITCH, lc, ks * 32 + 8 * quad);
#pragma unroll
                    for (int n = 0; n < 4; n++)
                        acc0[n] = mma16<NS>(fa, ldfrag_tr<NS>(Vs, KT * PITCH, PITCH, ks * 32, n * 16), acc0[n]);
                }
            } else {
                // ---- dP = dO V^T ; dS = P * (keep*dP - delta) * scale
                f32x4 dp[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    dp[j] = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 2; ks++)
                        dp[j] = mma16<NS>(fa_do[ks], ldfrag<NS>(Vs, KT * PITCH, PITCH, j * 16 + lc, ks * 32 + 8 * quad), dp[j]);
                }
                T* pd_g = reinterpret_cast<T*>(p.pd) + (((long)b * p.H + h) * Tq) * p.lds;
                T* ds_g = reinterpret_cast<T*>(p.ds) + (((long)b * p.H + h) * Tq) * p.lds;
                bf16_t* DGw = NS == 1 ? reinterpret_cast<bf16_t*>(Gw) : DG + (size_t)w * 16 * DG_PITCH;
                constexpr int DG_PLANE = 4 * 16 * DG_PITCH;  // only used when NS > 1
                // 2-byte outputs leave through the wave's LDS tile as 16-byte row chunks (rows are lds-pitched, lds % 8
                // == 0; columns in [Tk, lds) receive zeros); 4-byte outputs (test paths) are stored per element
                constexpr bool CHUNKED = sizeof(T) == 2 && NS == 1;
                auto store_tile = [&](T* dst) {
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const int id = lane + 64 * c, rr = id >> 3, ch = id & 7;
                        const int ig = i0 + 16 * w + rr, jg = j0 + 8 * ch;
                        const bf16x8 v8 = *reinterpret_cast<const bf16x8*>(Pw + rr * PITCH + 8 * ch);
                        if (ig < Tq && jg < p.lds) *reinterpret_cast<bf16x8*>(dst + (long)ig * p.lds + jg) = v8;
                    }
                };
                if (CHUNKED) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            Pw[(4 * quad + r) * PITCH + j * 16 + lc] = f2bf(pr[j][r] * keep[j][r]);
                    wave_sync();
                    store_tile(pd_g);
                }
                wave_sync();  // the G scratch has been consumed (scores) and the P tile has been read back
                if (RELPOS) {  // zero the skew scratch (16 x 104 per plane per wave)
                    for (int q = 0; q < NS; q++)
                        for (int id = lane; id < 16 * DG_PITCH / 8; id += 64)
                            *reinterpret_cast<bf16x8*>(DGw + (size_t)q * DG_PLANE + id * 8) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    wave_sync();
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int rr = 4 * quad + r, jj = j * 16 + lc;
                        const int ig = i0 + 16 * w + rr, jg = j0 + jj;
                        const float dsv = pr[j][r] * (keep[j][r] * dp[j][r] - delta[r]) * p.scale;
                        if (!CHUNKED && ig < Tq && jg < Tk) {
                            Elem<T>::st(pd_g + (long)ig * p.lds + jg, pr[j][r] * keep[j][r]);
                            Elem<T>::st(ds_g + (long)ig * p.lds + jg, dsv);
                        }
                        bf16_t sp[NS];
                        split_bf16<NS>(dsv, sp);
#pragma unroll
                        for (int q = 0; q < NS; q++) {
                            Pw[(size_t)q * PS_PLANE + rr * PITCH + jj] = sp[q];
                            if (RELPOS) DGw[(size_t)q * DG_PLANE + rr * DG_PITCH + jj - rr + 15] = sp[q];
                        }
                    }
                wave_sync();
                if (CHUNKED) store_tile(ds_g);
                // dQu += dS K   (contraction over keys: K fragments are key-strided in LDS)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const Frag<NS> fa = ldfrag<NS>(Pw, PS_PLANE, PITCH, lc, ks * 32 + 8 * quad);
#pragma unroll
                    for (int n = 0; n < 4; n++)
                        acc0[n] = mma16<NS>(fa, ldfrag_tr<NS>(Ks, KT * PITCH, PITCH, ks * 32, n * 16), acc0[n]);
                }
                if (RELPOS) {  // dQv += dG Pband
#pragma unroll
                    for (int ks = 0; ks < 3; ks++) {
                        const Frag<NS> fa = ldfrag<NS>(DGw, DG_PLANE, DG_PITCH, lc, ks * 32 + 8 * quad);
#pragma unroll
                        for (int n = 0; n < 4; n++)
                            acc1[n] = mma16<NS>(fa, ldfrag_tr<NS>(Pb, PB_ROWS * PITCH, PITCH, sb + ks * 32, n * 16), acc1[n]);
                    }
                }
            }
        }

        // ---- epilogue
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int ig = i0 + 16 * w + 4 * quad + r;
            if (ig >= Tq) continue;
            if (!BWD) {
                const float inv = l_run[r] > 0.f ? 1.f / l_run[r] : 0.f;
                T* o = reinterpret_cast<T*>(p.out) + b * p.sbo + (long)ig * p.ldo + h * DK;
#pragma unroll
                for (int n = 0; n < 4; n++) Elem<T>::st(o + n * 16 + lc, acc0[n][r] * inv);
                if (p.out2) {
                    bf16_t* o2 = reinterpret_cast<bf16_t*>(p.out2) + b * p.sbo + (long)ig * p.ldo + h * DK;
#pragma unroll
                    for (int n = 0; n < 4; n++) o2[n * 16 + lc] = f2bf(acc0[n][r] * inv);
                }
                if (lc == 0) p.lse[((long)b * p.H + h) * Tq + ig] = l_run[r] > 0.f ? m_run[r] + logf(l_run[r]) : 0.f;
            } else if (RELPOS && p.dq_sum) {
                T* dq = reinterpret_cast<T*>(p.dq_sum) + b * p.sbdq + (long)ig * p.lddq + h * DK;
#pragma unroll
                for (int n = 0; n < 4; n++) Elem<T>::st(dq + n * 16 + lc, acc0[n][r] + acc1[n][r]);
            } else {
                T* dqu = reinterpret_cast<T*>(p.dqu) + b * p.sbq + (long)ig * p.ldq + h * DK;
#pragma unroll
                for (int n = 0; n < 4; n++) Elem<T>::st(dqu + n * 16 + lc, acc0[n][r]);
                if (RELPOS) {
                    T* dqv = reinterpret_cast<T*>(p.dqv) + b * p.sbq + (long)ig * p.ldq + h * DK;
#pragma unroll
                    for (int n = 0; n < 4; n++) Elem<T>::st(dqv + n * 16 + lc, acc1[n][r]);
                }
            }
        }
        if (BWD && RELPOS && p.dq_sum) {
            // gradients of the two position biases: column sums of dQu / dQv over this block's query rows.  Rows past Tq
            // hold exact zeros (their dS is masked), so every accumulator row counts.  Wave: 4 rows per lane, then the 4
            // quads; block: through LDS (the key / value tiles are dead); one atomic per column and block.
            float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][64]
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 4; n++) {
                float su = 0.f, sv = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const bool ok = i0 + 16 * w + 4 * quad + r < Tq;
                    su += ok ? acc0[n][r] : 0.f;
                    sv += ok ? acc1[n][r] : 0.f;
                }
                su += __shfl_xor(su, 16);
                su += __shfl_xor(su, 32);
                sv += __shfl_xor(sv, 16);
                sv += __shfl_xor(sv, 32);
                if (quad == 0) {
                    red[(w * 2 + 0) * 64 + n * 16 + lc] = su;
                    red[(w * 2 + 1) * 64 + n * 16 + lc] = sv;
                }
            }
            __syncthreads();
            if (threadIdx.x < 128) {
                const int which = threadIdx.x >> 6, d = threadIdx.x & 63;
                const float t = (red[(0 * 2 + which) * 64 + d] + red[(1 * 2 + which) * 64 + d]) +
                                (red[(2 * 2 + which) * 64 + d] + red[(3 * 2 + which) * 64 + d]);
                atomicAdd((which ? p.dv : p.du) + h * DK + d, t);
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Forward pass, bf16, TRANSPOSED formulation: the score tile is computed as S^T = K Q^T (keys along the MFMA M
// dimension, the wave's 16 query rows along N), so that a lane ends up with 16 scores OF ONE query row (lane & 15) -- keys
// 16j + 4*quad + r -- and
//   * the probabilities are already laid out as the B operand of O^T = V^T P^T (8 keys per lane and k-step, with the key
//     order of the contraction permuted consistently on the V^T side): P never travels through LDS (the generic kernel
//     converts, stores 2 bytes per element, synchronises and reads it back as an A operand);
//   * the row statistics live in ONE register each per lane (not four), the running sum stays a per-lane partial until
//     the end, and the per-tile maximum needs two cross-quad exchanges instead of 4 x 4 within-row shuffle steps;
//   * scale * log2(e) is folded into the score scale (v_exp_f32 is an exp2), the key-padding mask is an additive bias
//     staged once per key tile, the relative-position band product leaves as 16-byte LDS stores, and the output row
//     pieces leave as 8-byte stores.
// The key-tile iteration of the generic kernel is ~1400 VALU/LDS instructions per wave against 26 MFMAs (416 clocks) --
// at two waves per SIMD the kernel time IS that instruction count.  Same grid, same staging, same
// arithmetic (online softmax in f32, dropout keep mask by (row, key) index), so the backward kernel pairs with it unchanged.
// F16 = 1: q / k / v / position band and the probabilities are IEEE half (v_mfma_f32_16x16x32_f16) -- the mixed mode's forward;
// `out` is then f16 and out2 (optional) its bf16 twin for the backward pass.
// KSP = 2 (round 6): the KEY range is split over two groups of four waves (512 threads): group g runs the loop below over its half of
// the key tiles on LDS buffers of its own, and the two (m, l, O) states meet through LDS at the end.  A block's time is the serial
// chain over its key tiles (7 at T = 400) at one or two waves per SIMD -- 336 blocks for 256 CUs leave most wave slots empty -- so
// halving the chain is worth more than the merge costs (T = 400 forward: profiles/r6_attention_ksplit.txt).
template <bool RELPOS, int F16 = 0, int KSP = 1>
struct AttnFwdT {
    static constexpr int KS_E = KT * PITCH, VS_E = KT * PITCH, PB_E = RELPOS ? PB_ROWS * PITCH : 0;
    static constexpr int G_F = RELPOS ? 4 * 16 * G_PITCH : 0;  // f32, per wave [16 q][G_PITCH]
    static constexpr size_t GROUP_BYTES = (size_t)(KS_E + VS_E + PB_E) * 2 + (size_t)(G_F + KT) * 4;
    static constexpr int MERGE_F = 20;  // floats a lane of group 1 hands over: 16 accumulators, running maximum, row sum (+ 2: 16-byte aligned rows)
    static constexpr size_t LDS_BYTES = KSP * GROUP_BYTES + (KSP > 1 ? (size_t)256 * MERGE_F * 4 : 0);
    static_assert(GROUP_BYTES % 16 == 0, "group buffers stay 16-byte aligned");

    static AVSR_DEV void run(const AttnParams& p, char* smem) {
        const int grp = KSP > 1 ? (int)(threadIdx.x >> 8) : 0, tid = threadIdx.x & 255;
        bf16_t* Ks = reinterpret_cast<bf16_t*>(smem + grp * GROUP_BYTES);
        bf16_t* Vs = Ks + KS_E;
        bf16_t* Pb = Vs + VS_E;
        float* Gs = reinterpret_cast<float*>(Pb + PB_E);
        float* bias = Gs + G_F;  // [KT] additive key mask of the staged tile (0 / NEG_BIG)

        const int lane = tid & 63, w = tid >> 6;
        const int quad = lane >> 4, lc = lane & 15;
        const int i0 = blockIdx.x * QT, h = blockIdx.y, b = blockIdx.z;
        const int Tq = p.Tq, Tk = p.Tk;
        const bf16_t* qu = reinterpret_cast<const bf16_t*>(p.qu) + b * p.sbq + h * DK;
        const bf16_t* qv = RELPOS ? reinterpret_cast<const bf16_t*>(p.qv) + b * p.sbq + h * DK : nullptr;
        const bf16_t* kk = reinterpret_cast<const bf16_t*>(p.k) + b * p.sbk + h * DK;
        const bf16_t* vv = reinterpret_cast<const bf16_t*>(p.v) + b * p.sbv + h * DK;
        const bf16_t* pos = RELPOS ? reinterpret_cast<const bf16_t*>(p.pos) + h * DK : nullptr;
        const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
        const uint64_t seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
        const float scale2 = p.scale * 1.4426950408889634f;  // scores in log2 units

        // this lane's query row and its Q fragments (B operand: n = lc, k chunk = quad)
        const int qrow = i0 + 16 * w + lc;
        const bool q_ok = qrow < Tq;
        bf16x8 fq_u[2], fq_v[2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            fq_u[ks] = fq_v[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (q_ok) {
                fq_u[ks] = *reinterpret_cast<const bf16x8*>(qu + (long)qrow * p.ldq + ks * 32 + 8 * quad);
                if (RELPOS) fq_v[ks] = *reinterpret_cast<const bf16x8*>(qv + (long)qrow * p.ldq + ks * 32 + 8 * quad);
            }
        }
        float m_run = NEG_BIG, l_part = 0.f;
        f32x4 acc[4];  // O^T[d = 16n + 4*quad + r][q = lc]
#pragma unroll
        for (int n = 0; n < 4; n++) acc[n] = f32x4{0, 0, 0, 0};

        const int ntiles = (Tk + KT - 1) / KT;
        const bool wave_active = i0 + 16 * w < Tq;
        const bool row_mask = p.mask && p.mask_sq != 0;  // false: one mask row per batch item (key-padding mask)
        const long mrow = row_mask ? (long)b * p.mask_sb + (long)(q_ok ? qrow : 0) * p.mask_sq : 0;
        const uint64_t drop_row = (((uint64_t)b * p.H + h) * Tq + (q_ok ? qrow : 0)) * (uint64_t)Tk;
        TilePrefetch<RELPOS> tp;  // next tile's K / V / band rows in registers while this one is multiplied
        tp.tid = tid;
        float rbias = 0.f;
        auto fetch_bias = [&](int j0) {
            if (tid < KT) {
                const int jg = j0 + tid;
                bool ok = jg < Tk;
                if (ok && p.mask && !row_mask) ok = p.mask[(long)b * p.mask_sb + jg] != 0;
                rbias = ok ? 0.f : NEG_BIG;
            }
        };
        // this group's key tiles [kt_begin, kt_end); every group makes `per` trips (the barriers are block-wide)
        const int per = (ntiles + KSP - 1) / KSP, kt_begin = grp * per, kt_end = min(ntiles, kt_begin + per);
        if (kt_begin < kt_end) {
            tp.fetch(kk, p.ldk, vv, p.ldv, pos, p.ldp, kt_begin * KT, Tk, kt_begin * KT - i0 + Tq - 1 - 63, 2 * Tq - 1);
            fetch_bias(kt_begin * KT);
        }
        for (int it = 0; it < per; it++) {
            const int kt = kt_begin + it, j0 = kt * KT;
            const bool live = kt < kt_end;
            __syncthreads();  // every wave is done reading the previous tile
            if (live) {
                tp.commit(Ks, Vs, Pb);
                if (tid < KT) bias[tid] = rbias;
            }
            __syncthreads();
            if (kt + 1 < kt_end) {
                tp.fetch(kk, p.ldk, vv, p.ldv, pos, p.ldp, j0 + KT, Tk, j0 + KT - i0 + Tq - 1 - 63, 2 * Tq - 1);
                fetch_bias(j0 + KT);
            }
            if (!wave_active || !live) continue;  // wave-uniform: this wave's 16 query rows are all past Tq / no tile left for this group

            // ---- S^T tiles: st[j][r] = score(key 16j + 4quad + r, query lc)
            f32x4 st[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                st[j] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
                    st[j] = mfma16x<F16>(ldfrag<1>(Ks, KT * PITCH, PITCH, j * 16 + lc, ks * 32 + 8 * quad).p[0], fq_u[ks], st[j]);
            }
            if (RELPOS) {
                // G^T = Pband (q+v)^T: lane holds G[q = lc][band 16t + 4quad .. +3] -> one 16-byte store per band tile
                const int sb = 48 - 16 * w;
                float* Gw = Gs + w * 16 * G_PITCH;
#pragma unroll
                for (int t = 0; t < 5; t++) {
                    f32x4 g = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 2; ks++)
                        g = mfma16x<F16>(ldfrag<1>(Pb, PB_ROWS * PITCH, PITCH, sb + t * 16 + lc, ks * 32 + 8 * quad).p[0], fq_v[ks], g);
                    *reinterpret_cast<f32x4*>(Gw + lc * G_PITCH + t * 16 + 4 * quad) = g;
                }
                wave_sync();
                // bd[q, key] = G[q, key - q + 15] (rel_shift as an index map, inside this wave's 16 x 80 band window)
                const float* grow = Gw + lc * G_PITCH - lc + 15 + 4 * quad;
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) st[j][r] += grow[j * 16 + r];
            }
            // ---- scale (log2 units) + mask, tile maximum of this query row
            float mx = NEG_BIG;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x4 bj = *reinterpret_cast<const f32x4*>(bias + j * 16 + 4 * quad);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float sv = st[j][r] * scale2 + bj[r];
                    if (row_mask) {
                        const int jg = j0 + j * 16 + 4 * quad + r;
                        if (jg < Tk && p.mask[mrow + jg] == 0) sv = NEG_BIG;
                    }
                    sv = fmaxf(sv, NEG_BIG);
                    st[j][r] = sv;
                    mx = fmaxf(mx, sv);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float m_safe = m_new <= 0.5f * NEG_BIG ? 0.f : m_new;  // nothing attended yet: exp2(NEG_BIG - 0) = 0
            const float corr = exp2f(m_run - m_safe);                    // (0 when m_run is still NEG_BIG)
            float rs = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    st[j][r] = exp2f(st[j][r] - m_safe);
                    rs += st[j][r];
                }
            l_part = l_part * corr + rs;
            m_run = m_new;
#pragma unroll
            for (int n = 0; n < 4; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[n][r] *= corr;
            if (p.drop_p > 0.f) {
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        st[j][r] *= dropout_scale(seed, drop_row + (uint64_t)(j0 + j * 16 + 4 * quad + r), p.drop_p, inv_keep);
            }
            // ---- O^T += V^T P^T.  k-step ks contracts the 32 keys of score tiles 2ks and 2ks+1; inside it the lane's 8 keys
            // are {16(2ks) + 4quad + r} and {16(2ks+1) + 4quad + r}, and the V^T fragment is fetched in the same order
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                bf16x8 pb;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    pb[r] = f32_to_raw16<F16>(st[2 * ks][r]);
                    pb[4 + r] = f32_to_raw16<F16>(st[2 * ks + 1][r]);
                }
#pragma unroll
                for (int n = 0; n < 4; n++) {
                    const bf16_t* p0 = Vs + (ks * 32 + 4 * quad + (lc >> 2)) * PITCH + n * 16 + 4 * (lc & 3);
                    const bf16x4 lo = lds_tr16(p0), hi = lds_tr16(p0 + 16 * PITCH);
                    acc[n] = mfma16x<F16>(bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}, pb, acc[n]);
                }
            }
        }
        // ---- epilogue: full row sum, normalise, 8-byte row pieces
        float l = l_part;
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        if (KSP > 1) {
            // the two key halves meet: group 1 hands (O, m, l) over through LDS, group 0 rescales both to the common maximum
            float* mb = reinterpret_cast<float*>(smem + KSP * GROUP_BYTES) + tid * MERGE_F;
            if (grp == 1) {
#pragma unroll
                for (int n = 0; n < 4; n++) *reinterpret_cast<f32x4*>(mb + 4 * n) = acc[n];
                mb[16] = m_run;
                mb[17] = l;
            }
            __syncthreads();
            if (grp == 1) return;
            const float m1 = mb[16], l1 = mb[17];
            const float m_new = fmaxf(m_run, m1);
            const float m_safe = m_new <= 0.5f * NEG_BIG ? 0.f : m_new;
            const float c0 = exp2f(m_run - m_safe), c1 = exp2f(m1 - m_safe);
#pragma unroll
            for (int n = 0; n < 4; n++) {
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(mb + 4 * n);
#pragma unroll
                for (int r = 0; r < 4; r++) acc[n][r] = acc[n][r] * c0 + a1[r] * c1;
            }
            l = l * c0 + l1 * c1;
            m_run = m_new;
        }
        if (!q_ok) return;
        const float inv = l > 0.f ? 1.f / l : 0.f;
        bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + b * p.sbo + (long)qrow * p.ldo + h * DK;
        bf16_t* o2 = (F16 && p.out2) ? reinterpret_cast<bf16_t*>(p.out2) + b * p.sbo + (long)qrow * p.ldo + h * DK : nullptr;
#pragma unroll
        for (int n = 0; n < 4; n++) {
            bf16x4 v4;
#pragma unroll
            for (int r = 0; r < 4; r++) v4[r] = f32_to_raw16<F16>(acc[n][r] * inv);
            *reinterpret_cast<bf16x4*>(o + n * 16 + 4 * quad) = v4;
            if (o2) {
#pragma unroll
                for (int r = 0; r < 4; r++) v4[r] = (short)f2bf(acc[n][r] * inv);
                *reinterpret_cast<bf16x4*>(o2 + n * 16 + 4 * quad) = v4;
            }
        }
        if (quad == 0) p.lse[((long)b * p.H + h) * Tq + qrow] = l > 0.f ? (m_run + log2f(l)) * 0.6931471805599453f : 0.f;
    }
};

template <bool RELPOS, int F16 = 0, int KSP = 1>
__global__ __launch_bounds__(256 * KSP) void attn_fwd_t_kernel(AttnParams p) {
    AVSR_DYN_SMEM(smem);
    AttnFwdT<RELPOS, F16, KSP>::run(p, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward "dq" pass, bf16, in the same transposed formulation (see AttnFwdT): S^T = K Q^T and dP^T = V dO^T put 16 keys of
// ONE query row into each lane, so P, dP and dS are elementwise register work, dS is already the B operand of
// dQu^T += K^T dS^T (key order of the contraction permuted consistently on the K^T side), delta / lse are one register
// each, pd = dropout(P) and ds = scale * dS leave as 8-byte row pieces straight from registers, and -- with 40 VGPRs less
// than the generic kernel -- the next key tile fits into registers as a prefetch.  Only the position term still crosses
// LDS: dQv += skew(dS) Pband needs dS re-indexed by (key - query), a lane-dependent shift.
template <bool RELPOS>
struct AttnBwdT {
    static constexpr int KS_E = KT * PITCH, VS_E = KT * PITCH, PB_E = RELPOS ? PB_ROWS * PITCH : 0;
    static constexpr int G_F = RELPOS ? 4 * 16 * G_PITCH : 0;  // f32 per wave [16 q][G_PITCH]; re-used as the bf16 dS skew scratch
    static constexpr size_t LDS_BYTES = (size_t)(KS_E + VS_E + PB_E) * 2 + (size_t)(G_F + KT) * 4 + 4 * 2 * 64 * 4;
    static_assert(16 * DG_PITCH * 2 <= 16 * G_PITCH * 4, "dS skew scratch must fit the G scratch");

    static AVSR_DEV void run(const AttnParams& p, char* smem) {
        bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
        bf16_t* Vs = Ks + KS_E;
        bf16_t* Pb = Vs + VS_E;
        float* Gs = reinterpret_cast<float*>(Pb + PB_E);
        float* bias = Gs + G_F;  // [KT]
        float* red = bias + KT;  // [4 waves][2][64] (epilogue)

        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int quad = lane >> 4, lc = lane & 15;
        const int i0 = blockIdx.x * QT, h = blockIdx.y, b = blockIdx.z;
        const int Tq = p.Tq, Tk = p.Tk;
        const bf16_t* qu = reinterpret_cast<const bf16_t*>(p.qu) + b * p.sbq + h * DK;
        const bf16_t* qv = RELPOS ? reinterpret_cast<const bf16_t*>(p.qv) + b * p.sbq + h * DK : nullptr;
        const bf16_t* kk = reinterpret_cast<const bf16_t*>(p.k) + b * p.sbk + h * DK;
        const bf16_t* vv = reinterpret_cast<const bf16_t*>(p.v) + b * p.sbv + h * DK;
        const bf16_t* pos = RELPOS ? reinterpret_cast<const bf16_t*>(p.pos) + h * DK : nullptr;
        const bf16_t* dout = reinterpret_cast<const bf16_t*>(p.dout) + b * p.sbo + h * DK;
        const bf16_t* outp = reinterpret_cast<const bf16_t*>(p.out) + b * p.sbo + h * DK;
        const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
        const uint64_t seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
        constexpr float LOG2E = 1.4426950408889634f;
        const float scale2 = p.scale * LOG2E;

        const int qrow = i0 + 16 * w + lc;
        const bool q_ok = qrow < Tq;
        bf16x8 fq_u[2], fq_v[2], fdo[2];
        float delta = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            fq_u[ks] = fq_v[ks] = fdo[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (q_ok) {
                const int d0 = ks * 32 + 8 * quad;
                fq_u[ks] = *reinterpret_cast<const bf16x8*>(qu + (long)qrow * p.ldq + d0);
                if (RELPOS) fq_v[ks] = *reinterpret_cast<const bf16x8*>(qv + (long)qrow * p.ldq + d0);
                fdo[ks] = *reinterpret_cast<const bf16x8*>(dout + (long)qrow * p.ldo + d0);
                const bf16x8 o8 = *reinterpret_cast<const bf16x8*>(outp + (long)qrow * p.ldo + d0);
#pragma unroll
                for (int e = 0; e < 8; e++) delta += bf2f((bf16_t)fdo[ks][e]) * bf2f((bf16_t)o8[e]);
            }
        }
        delta += __shfl_xor(delta, 16);
        delta += __shfl_xor(delta, 32);  // delta(query lc) in all four quads
        const float lse2 = q_ok ? p.lse[((long)b * p.H + h) * Tq + qrow] * LOG2E : 0.f;
        f32x4 acc_u[4], acc_v[4];  // dQu^T / dQv^T [d = 16n + 4*quad + r][q = lc]
#pragma unroll
        for (int n = 0; n < 4; n++) acc_u[n] = acc_v[n] = f32x4{0, 0, 0, 0};

        const int ntiles = (Tk + KT - 1) / KT;
        const bool wave_active = i0 + 16 * w < Tq;
        const bool row_mask = p.mask && p.mask_sq != 0;
        const long mrow = row_mask ? (long)b * p.mask_sb + (long)(q_ok ? qrow : 0) * p.mask_sq : 0;
        const uint64_t drop_row = (((uint64_t)b * p.H + h) * Tq + (q_ok ? qrow : 0)) * (uint64_t)Tk;
        bf16_t* pd_g = reinterpret_cast<bf16_t*>(p.pd) + (((long)b * p.H + h) * Tq + (q_ok ? qrow : 0)) * p.lds;
        bf16_t* ds_g = reinterpret_cast<bf16_t*>(p.ds) + (((long)b * p.H + h) * Tq + (q_ok ? qrow : 0)) * p.lds;

        TilePrefetch<RELPOS> tp;
        tp.tid = threadIdx.x;
        float rbias = 0.f;
        auto fetch_bias = [&](int j0) {
            if (threadIdx.x < KT) {
                const int jg = j0 + threadIdx.x;
                bool ok = jg < Tk;
                if (ok && p.mask && !row_mask) ok = p.mask[(long)b * p.mask_sb + jg] != 0;
                rbias = ok ? 0.f : NEG_BIG;
            }
        };
        tp.fetch(kk, p.ldk, vv, p.ldv, pos, p.ldp, 0, Tk, -i0 + Tq - 1 - 63, 2 * Tq - 1);
        fetch_bias(0);
        for (int kt = 0; kt < ntiles; kt++) {
            const int j0 = kt * KT;
            __syncthreads();
            tp.commit(Ks, Vs, Pb);
            if (threadIdx.x < KT) bias[threadIdx.x] = rbias;
            __syncthreads();
            if (kt + 1 < ntiles) {
                tp.fetch(kk, p.ldk, vv, p.ldv, pos, p.ldp, j0 + KT, Tk, j0 + KT - i0 + Tq - 1 - 63, 2 * Tq - 1);
                fetch_bias(j0 + KT);
            }
            if (!wave_active) continue;

            // ---- S^T and dP^T tiles: [key 16j + 4quad + r][query lc]
            f32x4 st[4], dp[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                st[j] = dp[j] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    st[j] = mfma16(ldfrag<1>(Ks, KT * PITCH, PITCH, j * 16 + lc, ks * 32 + 8 * quad).p[0], fq_u[ks], st[j]);
                    dp[j] = mfma16(ldfrag<1>(Vs, KT * PITCH, PITCH, j * 16 + lc, ks * 32 + 8 * quad).p[0], fdo[ks], dp[j]);
                }
            }
            const int sb = 48 - 16 * w;
            float* Gw = Gs + w * 16 * G_PITCH;
            if (RELPOS) {
#pragma unroll
                for (int t = 0; t < 5; t++) {
                    f32x4 g = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 2; ks++)
                        g = mfma16(ldfrag<1>(Pb, PB_ROWS * PITCH, PITCH, sb + t * 16 + lc, ks * 32 + 8 * quad).p[0], fq_v[ks], g);
                    *reinterpret_cast<f32x4*>(Gw + lc * G_PITCH + t * 16 + 4 * quad) = g;
                }
                wave_sync();
                const float* grow = Gw + lc * G_PITCH - lc + 15 + 4 * quad;
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) st[j][r] += grow[j * 16 + r];
                wave_sync();  // the G scratch is consumed: it becomes the dS skew scratch below
            }
            // ---- P = exp(S - lse), dS = P (keep dP - delta) scale; pd / ds row pieces to HBM
            bf16x8 dsb[2];
            bf16_t* DGw = reinterpret_cast<bf16_t*>(Gw);
            if (RELPOS) {  // zero the skew scratch (16 x DG_PITCH bf16 per wave)
                for (int id = lane; id < 16 * DG_PITCH / 8; id += 64)
                    *reinterpret_cast<bf16x8*>(DGw + id * 8) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                wave_sync();
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x4 bj = *reinterpret_cast<const f32x4*>(bias + j * 16 + 4 * quad);
                bf16x4 pd4, ds4;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int jg = j0 + j * 16 + 4 * quad + r;
                    float sv = st[j][r] * scale2 + bj[r];
                    if (row_mask && jg < Tk && p.mask[mrow + jg] == 0) sv = NEG_BIG;
                    const float pr = sv > 0.5f * NEG_BIG ? exp2f(sv - lse2) : 0.f;
                    const float keep = dropout_scale(seed, drop_row + (uint64_t)jg, p.drop_p, inv_keep);
                    const float dsv = pr * (keep * dp[j][r] - delta) * p.scale;
                    pd4[r] = (short)f2bf(pr * keep);
                    ds4[r] = (short)f2bf(dsv);
                    if (RELPOS) DGw[lc * DG_PITCH + j * 16 + 4 * quad + r - lc + 15] = (bf16_t)ds4[r];
                }
                const int jg0 = j0 + j * 16 + 4 * quad;
                if (q_ok && jg0 < p.lds) {
                    *reinterpret_cast<bf16x4*>(pd_g + jg0) = pd4;
                    *reinterpret_cast<bf16x4*>(ds_g + jg0) = ds4;
                }
                // B operand of the key contraction: k-step j/2 takes tiles 2ks and 2ks+1
#pragma unroll
                for (int r = 0; r < 4; r++) dsb[j >> 1][(j & 1) * 4 + r] = ds4[r];
            }
            // ---- dQu^T += K^T dS^T (K^T fragments: transposed reads in the permuted key order)
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int n = 0; n < 4; n++) {
                    const bf16_t* p0 = Ks + (ks * 32 + 4 * quad + (lc >> 2)) * PITCH + n * 16 + 4 * (lc & 3);
                    const bf16x4 lo = lds_tr16(p0), hi = lds_tr16(p0 + 16 * PITCH);
                    acc_u[n] = mfma16(bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}, dsb[ks], acc_u[n]);
                }
            if (RELPOS) {  // dQv^T += Pband^T dG^T, dG[q][band] = dS[q][key = band + q - 15] read back as the B operand
                wave_sync();
#pragma unroll
                for (int ks = 0; ks < 3; ks++) {
                    const bf16x8 dgb = *reinterpret_cast<const bf16x8*>(DGw + lc * DG_PITCH + ks * 32 + 8 * quad);
#pragma unroll
                    for (int n = 0; n < 4; n++)
                        acc_v[n] = mfma16(ldfrag_tr<1>(Pb, PB_ROWS * PITCH, PITCH, sb + ks * 32, n * 16).p[0], dgb, acc_v[n]);
                }
            }
        }

        // ---- epilogue: dQ row pieces (8 bytes), optionally summed, + the position-bias gradients
        const bool sum_out = RELPOS && p.dq_sum;
        if (q_ok) {
            if (sum_out) {
                bf16_t* dq = reinterpret_cast<bf16_t*>(p.dq_sum) + b * p.sbdq + (long)qrow * p.lddq + h * DK;
#pragma unroll
                for (int n = 0; n < 4; n++) {
                    bf16x4 v4;
#pragma unroll
                    for (int r = 0; r < 4; r++) v4[r] = (short)f2bf(acc_u[n][r] + acc_v[n][r]);
                    *reinterpret_cast<bf16x4*>(dq + n * 16 + 4 * quad) = v4;
                }
            } else {
                bf16_t* dqu = reinterpret_cast<bf16_t*>(p.dqu) + b * p.sbq + (long)qrow * p.ldq + h * DK;
                bf16_t* dqv = RELPOS ? reinterpret_cast<bf16_t*>(p.dqv) + b * p.sbq + (long)qrow * p.ldq + h * DK : nullptr;
#pragma unroll
                for (int n = 0; n < 4; n++) {
                    bf16x4 a4, b4;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        a4[r] = (short)f2bf(acc_u[n][r]);
                        b4[r] = (short)f2bf(acc_v[n][r]);
                    }
                    *reinterpret_cast<bf16x4*>(dqu + n * 16 + 4 * quad) = a4;
                    if (RELPOS) *reinterpret_cast<bf16x4*>(dqv + n * 16 + 4 * quad) = b4;
                }
            }
        }
        if (sum_out) {
            // du / dv: column sums over this block's queries.  Lane (quad, lc) holds d = 16n + 4quad + r of query lc: sum the 16
            // lanes of a quad (rows past Tq hold exact zeros), then the four waves through LDS, one atomic per column
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 4; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float su = acc_u[n][r], sv = acc_v[n][r];
#pragma unroll
                    for (int m = 8; m >= 1; m >>= 1) {
                        su += __shfl_xor(su, m);
                        sv += __shfl_xor(sv, m);
                    }
                    if (lc == 0) {
                        red[(w * 2 + 0) * 64 + n * 16 + 4 * quad + r] = su;
                        red[(w * 2 + 1) * 64 + n * 16 + 4 * quad + r] = sv;
                    }
                }
            __syncthreads();
            if (threadIdx.x < 128) {
                const int which = threadIdx.x >> 6, d = threadIdx.x & 63;
                const float t = (red[(0 * 2 + which) * 64 + d] + red[(1 * 2 + which) * 64 + d]) +
                                (red[(2 * 2 + which) * 64 + d] + red[(3 * 2 + which) * 64 + d]);
                atomicAdd((which ? p.dv : p.du) + h * DK + d, t);
            }
        }
    }
};

template <bool RELPOS>
__global__ __launch_bounds__(256) void attn_bwd_t_kernel(AttnParams p) {
    AVSR_DYN_SMEM(smem);
    AttnBwdT<RELPOS>::run(p, smem);
}

template <class T, int NS, bool RELPOS, bool BWD>
__global__ __launch_bounds__(256) void attn_kernel(AttnParams p) {
    AVSR_DYN_SMEM(smem);
    Attn<T, NS, RELPOS, BWD>::run(p, smem);
}

template <bool BWD>
int launch_attn(const AttnParams& p, int dtype, int precise, bool relpos, hipStream_t stream) {
    dim3 grid((p.Tq + QT - 1) / QT, p.H, p.B), block(256), block2(512);
    // forward, 16-bit operands: split the key range over two wave groups once a block walks >= 4 key tiles (knob 8 = 2: never; 3: always)
    // Only where the grid leaves CUs idle AND two blocks still fit a CU: the plain (decoder) form -- source attention has
    // ceil(65 / 64) x B x H = 96 blocks walking 7 key tiles.  The relative-position form needs 61 KB per group (position band +
    // per-wave skew scratch): two groups = one block per CU, and its ~330 blocks would then run in two rounds of half-length
    // blocks -- nothing gained (knob 8 = 3 forces the split there too, for measurements).
    const bool ksp2 = !BWD && avsr_tune_knobs[8] != 2 &&
                      (avsr_tune_knobs[8] == 3 || (p.pos == nullptr && (p.Tk + KT - 1) / KT >= 4 && (long)grid.x * grid.y * grid.z <= 256));
#define AVSR_ATTN_GO(TT, NSV, RP)                                                                        \
    AVSR_LAUNCH((attn_kernel<TT, NSV, RP, BWD>), grid, block, (Attn<TT, NSV, RP, BWD>::LDS_BYTES), stream, p)
    if (precise) {
        if (dtype != 0) return -1;
        if (relpos) AVSR_ATTN_GO(float, 2, true); else AVSR_ATTN_GO(float, 2, false);
    } else if (dtype == 1) {
        if (!BWD && avsr_tune_knobs[8] != 1) {  // knob 8 = 1: the generic kernel (A/B runs)
            if (ksp2 && relpos) AVSR_LAUNCH((attn_fwd_t_kernel<true, 0, 2>), grid, block2, (AttnFwdT<true, 0, 2>::LDS_BYTES), stream, p);
            else if (ksp2) AVSR_LAUNCH((attn_fwd_t_kernel<false, 0, 2>), grid, block2, (AttnFwdT<false, 0, 2>::LDS_BYTES), stream, p);
            else if (relpos) AVSR_LAUNCH((attn_fwd_t_kernel<true>), grid, block, (AttnFwdT<true>::LDS_BYTES), stream, p);
            else AVSR_LAUNCH((attn_fwd_t_kernel<false>), grid, block, (AttnFwdT<false>::LDS_BYTES), stream, p);
        } else if (BWD && avsr_tune_knobs[9] != 1) {  // knob 9 = 1: the generic backward kernel
            if (relpos) AVSR_LAUNCH((attn_bwd_t_kernel<true>), grid, block, (AttnBwdT<true>::LDS_BYTES), stream, p);
            else AVSR_LAUNCH((attn_bwd_t_kernel<false>), grid, block, (AttnBwdT<false>::LDS_BYTES), stream, p);
        } else if (relpos) AVSR_ATTN_GO(bf16_t, 1, true);
        else AVSR_ATTN_GO(bf16_t, 1, false);
    } else if (dtype == 2) {  // f16 (forward only: the backward pass of the mixed mode runs on the bf16 twins)
        if (BWD) return -1;
        if (ksp2 && relpos) AVSR_LAUNCH((attn_fwd_t_kernel<true, 1, 2>), grid, block2, (AttnFwdT<true, 1, 2>::LDS_BYTES), stream, p);
        else if (ksp2) AVSR_LAUNCH((attn_fwd_t_kernel<false, 1, 2>), grid, block2, (AttnFwdT<false, 1, 2>::LDS_BYTES), stream, p);
        else if (relpos) AVSR_LAUNCH((attn_fwd_t_kernel<true, 1>), grid, block, (AttnFwdT<true, 1>::LDS_BYTES), stream, p);
        else AVSR_LAUNCH((attn_fwd_t_kernel<false, 1>), grid, block, (AttnFwdT<false, 1>::LDS_BYTES), stream, p);
    } else {
        if (relpos) AVSR_ATTN_GO(float, 1, true); else AVSR_ATTN_GO(float, 1, false);
    }
#undef AVSR_ATTN_GO
    return 0;
}

}  // namespace

static int attention_fwd_impl(const void* qu, const void* qv, const void* k, const void* v, const void* pos,
                              int dtype, int precise, const uint8_t* mask, int64_t mask_sb, int64_t mask_sq,
                              void* out, void* out2, float* lse, int B, int H, int Tq, int Tk, int dk, int ldq, int ldk,
                              int ldv, int ldp, int ldo, int64_t sbq, int64_t sbk, int64_t sbv, int64_t sbo,
                              float scale, float drop_p, uint64_t seed, const uint64_t* seed_dev, hipStream_t stream) {
    AVSR_REQUIRE(dk == DK, "attention: d_k must be 64");
    AVSR_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && (pos == nullptr || ldp % 8 == 0),
                 "attention: row strides must be multiples of 8");
    AVSR_REQUIRE(pos == nullptr || Tq == Tk, "attention: relative-position form needs Tq == Tk");
    if (B == 0 || Tq == 0) return 0;
    AttnParams p{};
    p.qu = qu; p.qv = qv; p.k = k; p.v = v; p.pos = pos;
    p.mask = mask; p.mask_sb = mask_sb; p.mask_sq = mask_sq;
    p.out = out; p.lse = lse;
    p.out2 = out2;
    p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldp = ldp; p.ldo = ldo;
    p.sbq = sbq; p.sbk = sbk; p.sbv = sbv; p.sbo = sbo;
    p.scale = scale; p.drop_p = drop_p; p.seed = seed; p.seed_dev = seed_dev;
    AVSR_REQUIRE(launch_attn<false>(p, dtype, precise, pos != nullptr, stream) == 0, "attention: bad dtype/precise combination");
    AVSR_CHECK_LAUNCH("attention_fwd");
    return 0;
}

extern "C" int avsr_attention_fwd(const void* qu, const void* qv, const void* k, const void* v, const void* pos,
                                  int dtype, int precise, const uint8_t* mask, int64_t mask_sb, int64_t mask_sq,
                                  void* out, float* lse, int B, int H, int Tq, int Tk, int dk, int ldq, int ldk,
                                  int ldv, int ldp, int ldo, int64_t sbq, int64_t sbk, int64_t sbv, int64_t sbo,
                                  float scale, float drop_p, uint64_t seed, const uint64_t* seed_dev, hipStream_t stream) {
    return attention_fwd_impl(qu, qv, k, v, pos, dtype, precise, mask, mask_sb, mask_sq, out, nullptr, lse, B, H, Tq, Tk, dk, ldq,
                              ldk, ldv, ldp, ldo, sbq, sbk, sbv, sbo, scale, drop_p, seed, seed_dev, stream);
}

// f16 forward (q / k / v / pos / out all IEEE half, v_mfma_f32_16x16x32_f16) + the bf16 twin out2 (may be NULL) of its output
extern "C" int avsr_attention_fwd_h16(const void* qu, const void* qv, const void* k, const void* v, const void* pos,
                                      const uint8_t* mask, int64_t mask_sb, int64_t mask_sq, void* out, void* out2, float* lse,
                                      int B, int H, int Tq, int Tk, int dk, int ldq, int ldk, int ldv, int ldp, int ldo,
                                      int64_t sbq, int64_t sbk, int64_t sbv, int64_t sbo, float scale, float drop_p, uint64_t seed,
                                      const uint64_t* seed_dev, hipStream_t stream) {
    return attention_fwd_impl(qu, qv, k, v, pos, 2, 0, mask, mask_sb, mask_sq, out, out2, lse, B, H, Tq, Tk, dk, ldq, ldk, ldv, ldp,
                              ldo, sbq, sbk, sbv, sbo, scale, drop_p, seed, seed_dev, stream);
}

// f32 (precise) forward + the bf16 twin of its output (same strides) in one pass
extern "C" int avsr_attention_fwd2(const void* qu, const void* qv, const void* k, const void* v, const void* pos,
                                   const uint8_t* mask, int64_t mask_sb, int64_t mask_sq, void* out, void* out2, float* lse,
                                   int B, int H, int Tq, int Tk, int dk, int ldq, int ldk, int ldv, int ldp, int ldo,
                                   int64_t sbq, int64_t sbk, int64_t sbv, int64_t sbo, float scale, float drop_p, uint64_t seed,
                                   const uint64_t* seed_dev, hipStream_t stream) {
    return attention_fwd_impl(qu, qv, k, v, pos, 0, 1, mask, mask_sb, mask_sq, out, out2, lse, B, H, Tq, Tk, dk, ldq, ldk, ldv, ldp,
                              ldo, sbq, sbk, sbv, sbo, scale, drop_p, seed, seed_dev, stream);
}

extern "C" int avsr_attention_bwd_dq(const void* qu, const void* qv, const void* k, const void* v, const void* pos,
                                     int dtype, int precise, const uint8_t* mask, int64_t mask_sb, int64_t mask_sq,
                                     const void* out, const float* lse, const void* dout, void* dqu, void* dqv,
                                     void* pd, void* ds, int lds, int B, int H, int Tq, int Tk, int dk, int ldq,
                                     int ldk, int ldv, int ldp, int ldo, int64_t sbq, int64_t sbk, int64_t sbv,
                                     int64_t sbo, float scale, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                                     void* dq_sum, int lddq, int64_t sbdq, float* du, float* dv, hipStream_t stream) {
    AVSR_REQUIRE(dq_sum == nullptr || (pos != nullptr && du != nullptr && dv != nullptr && lddq % 8 == 0),
                 "attention_bwd_dq: dq_sum needs the relative-position form, du, dv and a row pitch that is a multiple of 8");
    AVSR_REQUIRE(dq_sum != nullptr || dqu != nullptr, "attention_bwd_dq: no destination for the query gradient");
    AVSR_REQUIRE(dk == DK, "attention: d_k must be 64");
    AVSR_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && (pos == nullptr || ldp % 8 == 0),
                 "attention: row strides must be multiples of 8");
    AVSR_REQUIRE(pos == nullptr || Tq == Tk, "attention: relative-position form needs Tq == Tk");
    AVSR_REQUIRE(lds >= Tk && lds % 8 == 0, "attention: lds must cover Tk and be a multiple of 8");
    if (B == 0 || Tq == 0) return 0;
    AttnParams p{};
    p.qu = qu; p.qv = qv; p.k = k; p.v = v; p.pos = pos;
    p.mask = mask; p.mask_sb = mask_sb; p.mask_sq = mask_sq;
    p.out = const_cast<void*>(out); p.lse = const_cast<float*>(lse);
    p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldp = ldp; p.ldo = ldo;
    p.sbq = sbq; p.sbk = sbk; p.sbv = sbv; p.sbo = sbo;
    p.scale = scale; p.drop_p = drop_p; p.seed = seed; p.seed_dev = seed_dev;
    p.dout = dout; p.dqu = dqu; p.dqv = dqv; p.pd = pd; p.ds = ds; p.lds = lds;
    p.dq_sum = dq_sum; p.lddq = lddq; p.sbdq = sbdq; p.du = du; p.dv = dv;
    AVSR_REQUIRE(launch_attn<true>(p, dtype, precise, pos != nullptr, stream) == 0, "attention: bad dtype/precise combination");
    AVSR_CHECK_LAUNCH("attention_bwd_dq");
    return 0;
}
