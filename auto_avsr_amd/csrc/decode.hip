// decode.hip -- the hybrid CTC / attention beam search of the evaluation path, one decoding step per HOST CALL.
//
// What it replaces: one iteration of BatchBeamSearch.search (espnet/nets/batch_beam_search.py:208-349 over
// beam_search.py:330-406) with the scorers the reference wires (lightning.py:126-158): TransformerDecoder.batch_score
// (decoder/transformer_decoder.py:260-334 -> forward_one_step :226-258), CTCPrefixScorer.batch_score_partial
// (scorers/ctc.py:101-126 over ctc_prefix_score.py:71-187), LengthBonus (scorers/length_bonus.py), the pre-beam on the
// decoder scores and the top-k over beam x vocabulary.
//
// Why a host-side step function.  The python form of this loop (auto_avsr_amd/decoding.py: BatchBeamSearch._step) issues
// ~120 launches per emitted token through ctypes / ATen and is bound by that (2.2 ms per token at beam 40, the GPU idle
// most of the time), and -- like the reference -- it re-projects keys and values of ALL previous positions and of the
// whole encoder memory for every hypothesis at every step.  Here
//  * the step is one C call: ~75 launches issued back to back from C++, one 1 KB device-to-host copy, one stream sync;
//  * the source-attention K / V of every layer are projected ONCE per utterance (they do not depend on the hypothesis);
//  * self-attention K / V live in a per-layer cache indexed [position][slot]; a hypothesis carries the list of slots of
//    its ancestors (`anc`), so that re-ordering the beam copies a row of 4-byte indices per hypothesis instead of
//    gathering the cache -- the attention kernel below follows the indirection;
//  * only the NEW position of every hypothesis goes through LayerNorm / projections / FFN (n <= beam rows);
//  * arithmetic: the split-plane GEMM of the precise mode (three bf16 MFMAs per product, ~f32) and f32 everywhere else,
//    whatever numerical mode the encoder ran in -- at n <= 40 rows the step is launch-bound, cheaper operands buy nothing.
// Values are those of the reference's recomputation (same weights, same inputs per row); the python path stays as the
// implementation of the generic scorer API and as the cross-check (tests/test_decoding.py).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "prims.h"
#include "avsr_hip.h"

#ifdef AVSR_EMU
static inline int avsr_copy_to_host_sync(void* dst, const void* src, size_t n, hipStream_t) {
    memcpy(dst, src, n);
    return 0;
}
#else
static inline int avsr_copy_to_host_sync(void* dst, const void* src, size_t n, hipStream_t s) {
    hipError_t e = hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return (int)e;
}
#endif

namespace {

constexpr float LOGZERO = -10000000000.0f;
constexpr int MAX_BEAM = 128;

struct IdxList {
    int n;
    int idx[MAX_BEAM];
};

AVSR_DEV float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
AVSR_DEV float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// Single-query attention (attention.py:59-104 with one query row): block = (hypothesis b, head h), d_k = 64, `len` keys.
// Key / value row j of hypothesis b lives at kv + j * step_j + slot * step_slot (+ koff / voff) + h * 64 with
//     slot = anc == NULL ? 0 : (j == len - 1 ? b : anc[b * ld_anc + j])
// -- self-attention: the [position][slot] cache, the hypothesis' own ancestry; source attention: the utterance's memory
// projection, shared by every hypothesis.  16 lanes share a key (one float4 of the 64 dimensions each), a block of four
// waves takes 16 keys per iteration; scores are parked in LDS, softmax in f32 with expf.
__global__ __launch_bounds__(1024) void dec_attn_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ kv, long step_j,
                                                       long step_slot, int koff, int voff, const int* __restrict__ anc, int ld_anc,
                                                       int len, float scale, float* __restrict__ out, long ldo) {
    AVSR_DYN_SMEM(smem);
    const int nwv = blockDim.x >> 6;             // 4, 8 or 16 waves: 16 keys per wave and chunk
    float* sc = reinterpret_cast<float*>(smem);  // [len] scores, then probabilities
    float* part = sc + len;                      // [nwv][64]
    __shared__ float red[16];
    const int b = blockIdx.x, h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const f32x4 q4 = *reinterpret_cast<const f32x4*>(q + (size_t)b * ldq + h * 64 + 4 * c);
    auto row_of = [&](int j) -> const float* {
        const long slot = anc ? (j == len - 1 ? b : anc[(size_t)b * ld_anc + j]) : 0;
        return kv + (size_t)j * step_j + slot * step_slot + h * 64 + 4 * c;
    };
    // four keys per lane and pass: the loads of a chunk of 64 keys are requested together (a loop of one key per iteration is a
    // chain of exposed L2 round trips: 8 us for 100 keys)
    for (int j0 = 0; j0 < len; j0 += 16 * nwv) {
        f32x4 k4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + 4 * nwv * u + 4 * wave + g;
            k4[u] = j < len ? *reinterpret_cast<const f32x4*>(row_of(j) + koff) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + 4 * nwv * u + 4 * wave + g;
            float s = q4[0] * k4[u][0] + q4[1] * k4[u][1] + q4[2] * k4[u][2] + q4[3] * k4[u][3];
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m);
            if (c == 0 && j < len) sc[j] = s * scale;
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int j = threadIdx.x; j < len; j += blockDim.x) m = fmaxf(m, sc[j]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
    for (int w = 1; w < nwv; w++) m = fmaxf(m, red[w]);
    __syncthreads();
    float l = 0.f;
    for (int j = threadIdx.x; j < len; j += blockDim.x) {
        const float p = expf(sc[j] - m);
        sc[j] = p;
        l += p;
    }
    l = wave_sum(l);
    if (lane == 0) red[wave] = l;
    __syncthreads();
    l = 0.f;
    for (int w = 0; w < nwv; w++) l += red[w];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < len; j0 += 16 * nwv) {
        f32x4 v4[4];
        float p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + 4 * nwv * u + 4 * wave + g;
            v4[u] = j < len ? *reinterpret_cast<const f32x4*>(row_of(j) + voff) : f32x4{0.f, 0.f, 0.f, 0.f};
            p[u] = j < len ? sc[j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[e] += p[u] * v4[u][e];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        acc[e] += __shfl_xor(acc[e], 16);
        acc[e] += __shfl_xor(acc[e], 32);
    }
    if (g == 0) {
#pragma unroll
        for (int e = 0; e < 4; e++) part[wave * 64 + 4 * c + e] = acc[e];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = 0.f;
        for (int w = 0; w < nwv; w++) v += part[w * 64 + threadIdx.x];
        out[(size_t)b * ldo + h * 64 + threadIdx.x] = v / l;
    }
}

// Source attention of a decoding step (transformer_decoder.py:109-117): every hypothesis attends to the SAME projected memory,
// so a block takes HB hypotheses of one head and reads each key / value row once for all of them (with one hypothesis per
// block the 480 blocks of a beam-40 step pull 98 MB through the L2s at T = 400: 35 us per launch).  Layout as dec_attn_kernel:
// 16 lanes per key, 16 keys per block iteration, four iterations' loads in flight; wave w owns the softmax of hypothesis w.
constexpr int SRC_HB = 4;
__global__ __launch_bounds__(1024) void dec_src_attn_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ kv, long step_j,
                                                           int voff, int n, int len, float scale, float* __restrict__ out, long ldo) {
    AVSR_DYN_SMEM(smem);
    const int nwv = blockDim.x >> 6;             // 4, 8 or 16 waves
    float* sc = reinterpret_cast<float*>(smem);  // [SRC_HB][len]
    float* part = sc + SRC_HB * len;             // [nwv][SRC_HB][64]
    __shared__ float s_l[SRC_HB];
    const int b0 = blockIdx.x * SRC_HB, h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    f32x4 q4[SRC_HB];
#pragma unroll
    for (int i = 0; i < SRC_HB; i++)
        q4[i] = b0 + i < n ? *reinterpret_cast<const f32x4*>(q + (size_t)(b0 + i) * ldq + h * 64 + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* base = kv + h * 64 + 4 * c;
    for (int j0 = 0; j0 < len; j0 += 16 * nwv) {
        f32x4 k4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + 4 * nwv * u + 4 * wave + g;
            k4[u] = j < len ? *reinterpret_cast<const f32x4*>(base + (size_t)j * step_j) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + 4 * nwv * u + 4 * wave + g;
#pragma unroll
            for (int i = 0; i < SRC_HB; i++) {
                float s = q4[i][0] * k4[u][0] + q4[i][1] * k4[u][1] + q4[i][2] * k4[u][2] + q4[i][3] * k4[u][3];
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m);
                if (c == 0 && j < len) sc[i * len + j] = s * scale;
            }
        }
    }
    __syncthreads();
    if (wave < SRC_HB) {  // softmax of hypothesis `wave`
        float* row = sc + wave * len;
        float m = -INFINITY;
        for (int j = lane; j < len; j += 64) m = fmaxf(m, row[j]);
        m = wave_max(m);
        float l = 0.f;
        for (int j = lane; j < len; j += 64) {
            const float p = expf(row[j] - m);
            row[j] = p;
            l += p;
        }
        l = wave_sum(l);
        if (lane == 0) s_l[wave] = l;
    }
    __syncthreads();
    f32x4 acc[SRC_HB];
#pragma unroll
    for (int i = 0; i < SRC_HB; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < len; j0 += 16 * nwv) {
        f32x4 v4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + 4 * nwv * u + 4 * wave + g;
            v4[u] = j < len ? *reinterpret_cast<const f32x4*>(base + (size_t)j * step_j + voff) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + 4 * nwv * u + 4 * wave + g;
            if (j < len) {
#pragma unroll
                for (int i = 0; i < SRC_HB; i++) {
                    const float p = sc[i * len + j];
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[i][e] += p * v4[u][e];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < SRC_HB; i++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            acc[i][e] += __shfl_xor(acc[i][e], 16);
            acc[i][e] += __shfl_xor(acc[i][e], 32);
        }
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < SRC_HB; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) part[(wave * SRC_HB + i) * 64 + 4 * c + e] = acc[i][e];
    }
    __syncthreads();
    if (threadIdx.x < 64 * SRC_HB) {
        const int i = threadIdx.x >> 6, d = threadIdx.x & 63;
        if (b0 + i < n) {
            float v = 0.f;
            for (int w = 0; w < nwv; w++) v += part[(w * SRC_HB + i) * 64 + d];
            out[(size_t)(b0 + i) * ldo + h * 64 + d] = v / s_l[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The linear layers of a decoding step (transformer_decoder.py:84-126 on ONE position per hypothesis): C[M][N] =
// act(LN?(A)[M][K] . W[N][K]^T + bias) + resid with M <= beam rows.  On the tiled GEMM (64 x 64 tile, 12 - 48 sequential
// k-tiles behind a two-stage ring) such a launch takes 25 us whatever M is: with 12 - 48 blocks on 256 CUs it is a chain of
// dependent HBM round trips.  Here a block owns 16 weight rows and 64 NW columns of K (all of K = 768: NW = 12 waves), wave w
// its own 64 columns; split hi / lo bf16 planes are formed in registers, three MFMAs per product (the precise mode's
// arithmetic); the NW partial tiles meet in LDS.
// LayerNorm in front of a sub-layer's first projection (pre-norm blocks) is applied to the A fragments; the row statistics it
// needs were left behind by whatever produced the rows (st_out below: per-row sum and sum of squares of every block's 16
// columns, summed by the consumer) -- the 19 stand-alone LayerNorm launches of a step disappear and no kernel reads its input
// twice.
// History of the kernel (MI355X, per launch at M = 40, K = 768): 32-row MFMA fragments streamed by every lane from its own row:
// 43 us -- 32 distinct cache lines per wave-instruction, the CU's address path (one line per cycle) was the bound -- and 19 us
// more when the LayerNorm statistics re-read A; the same tile staged by LDS-DMA (8 lanes per 128-byte row piece, XOR-swizzled
// wave-private regions, two phases): 9.4 us; the 16-row form below (whole cache lines per four lanes, no staging, one round
// trip): 9.5 us.  The last two differ in everything but the result: what remains is the price of ANY dependent launch that
// touches fresh HBM lines here (a 40-block row-sum of 120 KB takes 6 us), not of the data path.
struct SkinnyArgs {
    const float* A;
    long lda;
    const float* W;
    long ldw;
    const float *bias, *ln_g, *ln_b;
    float eps;
    const float* st_in;  // LN: [M][st_in_nt][2] partial (sum, sum of squares) of every row of A
    int st_in_nt;
    const float* resid;
    long ldr;
    float* C;
    long ldc;
    float* st_out;  // [M][gridDim.x][2] or NULL: the same statistics of the rows written here
    int M, N, K, act;
    int Z;           // K slices (blockIdx.z); Z > 1: raw partial sums to partial[z][M][N], finished by rowsum_kernel
    float* partial;
};

AVSR_DEV void split8(const float* x, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const bf16_t h = f2bf(x[e]);
        hi[e] = (short)h;
        lo[e] = (short)f2bf(x[e] - bf2f(h));
    }
}

// v_mfma_f32_16x16x32, no staging.  In the 16-row fragment layout a lane holds 8 consecutive k of row (lane & 15) and the four lanes lane, lane + 16, +32, +48 hold the
// next 8 each: a wave-instruction reads 16 rows x 128 contiguous bytes -- whole cache lines, four lanes per line -- so the
// fragments can come straight from global memory without the one-line-per-lane address traffic that sank the first 32-row
// version.  A block owns 16 weight rows (twice the blocks of the staged kernel: 48 - 316 of them), wave w its 64 columns of K
// = two MFMA steps; every operand of a wave is requested up front (4 + 12 16-byte loads per lane) and arrives in ONE round trip
// instead of the staged kernel's two DMA phases.
constexpr int SK16_ROWS = 48;  // three 16-row tiles per block

template <int NW>
__global__ __launch_bounds__(64 * NW) void skinny16_kernel(SkinnyArgs a) {
    AVSR_DYN_SMEM(smem);
    float* red = reinterpret_cast<float*>(smem);  // [NW][48][17]
    __shared__ float s_mean[SK16_ROWS], s_rstd[SK16_ROWS];
    constexpr int NT = 64 * NW, RP = 17;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * SK16_ROWS;
    const int rows = min(SK16_ROWS, a.M - m0);
    const int li = lane & 15, kq = lane >> 4;
    const int kb = (blockIdx.z * NW + wave) * 64 + 8 * kq;  // this lane's first column
    // every operand of this wave, requested before anything else (LayerNorm statistics below overlap the round trip)
    const bool nv = n0 + li < a.N;
    const float* wrow = a.W + (size_t)(nv ? n0 + li : 0) * a.ldw + kb;
    float w8[2][8], x8[3][2][8];
#pragma unroll
    for (int s = 0; s < 2; s++) load8(wrow + 32 * s, w8[s]);
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int r = li + 16 * t;
        const float* arow = a.A + (size_t)(m0 + min(r, rows - 1)) * a.lda + kb;
#pragma unroll
        for (int s = 0; s < 2; s++) load8(arow + 32 * s, x8[t][s]);
    }
    if (a.ln_g) {  // row statistics from the producer's partial sums: 16 lanes per row
        const int sub = tid & 15;
        for (int r0 = 0; r0 < rows; r0 += NT >> 4) {
            const int r = r0 + (tid >> 4);
            float s1 = 0.f, s2 = 0.f;
            if (r < rows)
                for (int j = sub; j < a.st_in_nt; j += 16) {
                    const float* q = a.st_in + ((size_t)(m0 + r) * a.st_in_nt + j) * 2;
                    s1 += q[0];
                    s2 += q[1];
                }
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) {
                s1 += __shfl_xor(s1, m);
                s2 += __shfl_xor(s2, m);
            }
            if (r < rows && sub == 0) {
                const float mean = s1 / (float)a.K;
                s_mean[r] = mean;
                s_rstd[r] = 1.0f / sqrtf(fmaxf(s2 / (float)a.K - mean * mean, 0.f) + a.eps);
            }
        }
        __syncthreads();
    }
    f32x4 acc[3];
#pragma unroll
    for (int t = 0; t < 3; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; s++) {
        if (!nv) {
#pragma unroll
            for (int e = 0; e < 8; e++) w8[s][e] = 0.f;
        }
        bf16x8 whi, wlo;
        split8(w8[s], whi, wlo);
        float g8v[8], b8v[8];
        if (a.ln_g) {
            load8(a.ln_g + kb + 32 * s, g8v);
            load8(a.ln_b + kb + 32 * s, b8v);
        }
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int r = li + 16 * t;
            if (a.ln_g) {
                const float mean = r < rows ? s_mean[r] : 0.f, rstd = r < rows ? s_rstd[r] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) x8[t][s][e] = (x8[t][s][e] - mean) * rstd * g8v[e] + b8v[e];
            }
            bf16x8 ahi, alo;
            split8(x8[t][s], ahi, alo);
            acc[t] = mfma16(alo, whi, acc[t]);
            acc[t] = mfma16(ahi, wlo, acc[t]);
            acc[t] = mfma16(ahi, whi, acc[t]);
        }
    }
    float* mine = red + (size_t)wave * SK16_ROWS * RP;
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) mine[(16 * t + 4 * kq + r) * RP + li] = acc[t][r];
    __syncthreads();
    for (int idx = tid; idx < SK16_ROWS * 16; idx += NT) {  // (wave-uniform trip count: 768 and NT are multiples of 64)
        const int row = idx >> 4, col = idx & 15;
        const bool ok = row < rows && n0 + col < a.N;
        float v = 0.f;
        if (ok) {
#pragma unroll
            for (int w = 0; w < NW; w++) v += red[((size_t)w * SK16_ROWS + row) * RP + col];
            if (a.Z > 1) {
                a.partial[((size_t)blockIdx.z * a.M + m0 + row) * a.N + n0 + col] = v;
            } else {
                if (a.bias) v += a.bias[n0 + col];
                if (a.act == 1) v = fmaxf(v, 0.f);
                if (a.resid) v += a.resid[(size_t)(m0 + row) * a.ldr + n0 + col];
                a.C[(size_t)(m0 + row) * a.ldc + n0 + col] = v;
            }
        }
        if (a.st_out) {  // (never with Z > 1)
            float s1 = v, s2 = v * v;
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) {
                s1 += __shfl_xor(s1, m);
                s2 += __shfl_xor(s2, m);
            }
            if (col == 0 && row < rows) {
                float* q = a.st_out + ((size_t)(m0 + row) * gridDim.x + blockIdx.x) * 2;
                q[0] = s1;
                q[1] = s2;
            }
        }
    }
}

// C[m][n] = sum_z partial[z][m][n] + bias[n] + resid[m][n]: the K slices of a split contraction, summed in slice order; block =
// row; st_out [M][1][2]: the row's (sum, sum of squares) for the LayerNorm that follows
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ partial, int Z, int M, int N, const float* __restrict__ bias,
                                                     const float* __restrict__ resid, long ldr, float* __restrict__ C, long ldc,
                                                     float* __restrict__ st_out) {
    __shared__ float red[2][4];
    const int m = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s1 = 0.f, s2 = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        float v = 0.f;
        for (int z = 0; z < Z; z++) v += partial[((size_t)z * M + m) * N + n];
        if (bias) v += bias[n];
        if (resid) v += resid[(size_t)m * ldr + n];
        C[(size_t)m * ldc + n] = v;
        s1 += v;
        s2 += v * v;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        red[0][wave] = s1;
        red[1][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0 && st_out) {
        st_out[2 * m] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        st_out[2 * m + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// x[r] = embed[last[r]] * scale + pe  (transformer_decoder.py:186-189, embedding.py:78-87; one position for every hypothesis) and
// the row's LayerNorm statistics; block = row
__global__ __launch_bounds__(256) void dec_embed_kernel(const int64_t* __restrict__ last, const float* __restrict__ table,
                                                        const float* __restrict__ pe, float scale, int D, float* __restrict__ x,
                                                        float* __restrict__ st_out) {
    __shared__ float red[2][4];
    const int m = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* row = table + (size_t)last[m] * D;
    float s1 = 0.f, s2 = 0.f;
    for (int n = threadIdx.x; n < D; n += 256) {
        const float v = row[n] * scale + pe[n];
        x[(size_t)m * D + n] = v;
        s1 += v;
        s2 += v * v;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        red[0][wave] = s1;
        red[1][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        st_out[2 * m] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        st_out[2 * m + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Top-S selection by radix select on the order-preserving integer image of the floats (keys in LDS).  The
// digits are taken from key - min(key), most significant byte of the SPAN first: log-probabilities share sign and exponent,
// so the top byte of the raw key is the same for nearly every element -- a histogram pass on it is thousands of LDS atomics
// on one address (measured: 57 us for 5 049 values); relative to the span the first digit already spreads.
AVSR_DEV unsigned f2key(float f) {
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
AVSR_DEV unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
AVSR_DEV unsigned umax(unsigned a, unsigned b) { return a > b ? a : b; }
AVSR_DEV int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

constexpr int SEL_NT = 1024;  // threads of the selection kernels (the per-element passes are latency chains: more lanes, fewer trips)

struct RadixScratch {
    int hist[256];
    unsigned wmin[SEL_NT / 64], wmax[SEL_NT / 64];
    unsigned prefix, mask;
    int remaining;
    int wtot[2][SEL_NT / 64];
};

// keys[0 .. n): finds the S-th largest key `thr`; returns the number of keys above it, `eq_take` = how many keys equal to it
// belong to the selection (taken in index order).  S <= n.  All SEL_NT threads call it.
AVSR_DEV void radix_select(const unsigned* keys, int n, int S, RadixScratch& sc, unsigned& thr, int& n_gt, int& eq_take) {
    constexpr int NWV = SEL_NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned lo = 0xffffffffu, hi = 0u;
    for (int i = tid; i < n; i += SEL_NT) {
        lo = umin(lo, keys[i]);
        hi = umax(hi, keys[i]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        lo = umin(lo, (unsigned)__shfl_xor((int)lo, m));
        hi = umax(hi, (unsigned)__shfl_xor((int)hi, m));
    }
    if (lane == 0) {
        sc.wmin[wave] = lo;
        sc.wmax[wave] = hi;
    }
    if (tid == 0) {
        sc.prefix = 0u;
        sc.mask = 0u;
        sc.remaining = S;
    }
    __syncthreads();
    unsigned kmin = sc.wmin[0], kmax = sc.wmax[0];
#pragma unroll
    for (int w = 1; w < NWV; w++) {
        kmin = umin(kmin, sc.wmin[w]);
        kmax = umax(kmax, sc.wmax[w]);
    }
    const unsigned span = kmax - kmin;
    int bits = 0;
    while (bits < 32 && (span >> bits) != 0u) bits++;
    const int npass = (bits + 7) / 8;
    for (int pass = npass - 1; pass >= 0; pass--) {
        if (tid < 256) sc.hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = sc.prefix, mask = sc.mask;
        for (int i = tid; i < n; i += SEL_NT) {
            const unsigned k = keys[i] - kmin;
            if ((k & mask) == prefix) atomicAdd(&sc.hist[(k >> (8 * pass)) & 255u], 1);
        }
        __syncthreads();
        // digit d with  #(digits > d) < remaining <= #(digits >= d):  thread t < 256 looks at digit 255 - t; inclusive scan from the top
        const int rem = sc.remaining, own = tid < 256 ? sc.hist[255 - tid] : 0;
        int incl = tid < 256 ? wave_incl_scan(own, lane) : 0;  // (waves 0 - 3 whole: wave-uniform)
        if (tid < 256 && lane == 63) sc.wtot[0][wave] = incl;
        __syncthreads();
        if (tid < 256) {
            for (int w = 0; w < wave; w++) incl += sc.wtot[0][w];
            if (incl >= rem && incl - own < rem) {  // exactly one thread
                sc.remaining = rem - (incl - own);  // entries still to take among the keys that share the prefix extended by this digit
                sc.prefix = prefix | ((unsigned)(255 - tid) << (8 * pass));
                sc.mask = mask | (255u << (8 * pass));
            }
        }
        __syncthreads();
    }
    thr = sc.prefix + kmin;
    eq_take = sc.remaining;
    n_gt = S - eq_take;
}

// the selected keys' indices in index order (greater-than-threshold ones first, then the ties): out[0 .. S)
AVSR_DEV void radix_compact(const unsigned* keys, int n, unsigned thr, int n_gt, int eq_take, RadixScratch& sc, int* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + SEL_NT - 1) / SEL_NT;
    const int lo = min(n, tid * per), hi = min(n, lo + per);
    int ngt = 0, neq = 0;
    for (int i = lo; i < hi; i++) {
        ngt += keys[i] > thr;
        neq += keys[i] == thr;
    }
    const int sg = wave_incl_scan(ngt, lane), se = wave_incl_scan(neq, lane);
    if (lane == 63) {
        sc.wtot[0][wave] = sg;
        sc.wtot[1][wave] = se;
    }
    __syncthreads();
    int og = sg - ngt, oe = se - neq;
    for (int w = 0; w < wave; w++) {
        og += sc.wtot[0][w];
        oe += sc.wtot[1][w];
    }
    for (int i = lo; i < hi; i++) {
        const unsigned k = keys[i];
        if (k > thr) out[og++] = i;
        else if (k == thr) {
            if (oe < eq_take) out[n_gt + oe] = i;
            oe++;
        }
    }
}

// Pre-beam (beam_search.py:240-262 with pre_beam_score_key = "decoder" / "full"): the S best tokens of every row of the
// decoder's scores, as a SET -- fused with the log-softmax that produces those scores (transformer_decoder.py:256:
// logp = logits - logsumexp(logits), written for the selection kernel; the order of a row is that of its logits).
__global__ __launch_bounds__(SEL_NT) void logsoftmax_prebeam_kernel(const float* __restrict__ logits, float* __restrict__ logp, long ld, int V,
                                                                 int S, int64_t* __restrict__ cand) {
    AVSR_DYN_SMEM(smem);
    unsigned* keys = reinterpret_cast<unsigned*>(smem);  // [V]
    int* picked = reinterpret_cast<int*>(keys + V);      // [S]
    __shared__ RadixScratch sc;
    __shared__ float redf[SEL_NT / 64];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + (size_t)row * ld;
    float m = -INFINITY;
    for (int i = tid; i < V; i += SEL_NT) {
        const float v = x[i];
        keys[i] = f2key(v);
        m = fmaxf(m, v);
    }
    m = wave_max(m);
    if (lane == 0) redf[wave] = m;
    __syncthreads();
    m = redf[0];
    for (int w = 1; w < SEL_NT / 64; w++) m = fmaxf(m, redf[w]);
    __syncthreads();
    float l = 0.f;
    for (int i = tid; i < V; i += SEL_NT) l += expf(x[i] - m);
    l = wave_sum(l);
    if (lane == 0) redf[wave] = l;
    __syncthreads();
    float lsum = 0.f;
    for (int w = 0; w < SEL_NT / 64; w++) lsum += redf[w];
    const float lse = m + logf(lsum);
    float* y = logp + (size_t)row * ld;
    for (int i = tid; i < V; i += SEL_NT) y[i] = x[i] - lse;
    unsigned thr;
    int n_gt, eq_take;
    radix_select(keys, V, S, sc, thr, n_gt, eq_take);
    radix_compact(keys, V, thr, n_gt, eq_take, sc, picked);
    __syncthreads();
    for (int i = tid; i < S; i += SEL_NT) cand[(size_t)row * S + i] = picked[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// Scores of the viable extensions and the beam's K best (beam_search.py:264-297, batch_beam_search.py:107-129): hypothesis b
// extended by one of its S candidates or by <eos> (scorers/ctc.py gives <eos> its complete-sequence probability whether
// or not the pre-beam kept it; every other token outside the candidates carries the CTC scorer's LOGZERO and cannot
// reach a beam of K <= n * (S - 1) -- the host refuses configurations where it could).  Arithmetic order as the python
// form: ((w_dec * logp + w_len) + w_ctc * (log_psi - s_prev)) + score, f32.  ONE block: radix select of the
// K best values, then a rank sort of those K (value descending, entry ascending).
struct SelectArgs {
    const float* logp;  // [n][ld] decoder log-probabilities
    long ld;
    const int64_t* cand;  // [n][S]
    const float *psi, *psi_eos, *s_prev, *score;
    int n, S, K;
    int eos, blank, has_len;
    float w_dec, w_ctc, w_len;
    int* sel;     // [K][4]: prev, tok, candidate column (-1: <eos> outside the candidates), spare
    float* selv;  // [K][4]: total, decoder term logp, ctc term (log_psi - s_prev), ctc log_psi
};

__global__ __launch_bounds__(SEL_NT) void beam_select_kernel(SelectArgs a) {
    AVSR_DYN_SMEM(smem);
    const int NE = a.n * (a.S + 1);
    float* val = reinterpret_cast<float*>(smem);            // [NE]
    unsigned* keys = reinterpret_cast<unsigned*>(val + NE);  // [NE]
    int* picked = reinterpret_cast<int*>(keys + NE);         // [K]
    int* cnd = picked + a.K;                                 // [n][S] candidate tokens
    int* has_eos = cnd + a.n * a.S;                          // [n] <eos> is among the candidates of hypothesis b
    __shared__ RadixScratch sc;
    const int tid = threadIdx.x;
    for (int i = tid; i < a.n; i += SEL_NT) has_eos[i] = 0;
    __syncthreads();
    for (int i = tid; i < a.n * a.S; i += SEL_NT) {
        const int t = (int)a.cand[i];
        cnd[i] = t;
        if (t == a.eos) has_eos[i / a.S] = 1;
    }
    __syncthreads();
    auto decode = [&](int e, int& b, int& c, int& tok, float& dec, float& ctc_rel, float& log_psi) -> bool {
        b = e / (a.S + 1);
        c = e - b * (a.S + 1);
        bool valid = true;
        if (c < a.S) tok = cnd[b * a.S + c];
        else {
            tok = a.eos;
            valid = !has_eos[b];
            c = -1;
        }
        log_psi = tok == a.blank ? LOGZERO : (tok == a.eos ? a.psi_eos[b] : a.psi[(size_t)b * a.S + c]);
        ctc_rel = log_psi - a.s_prev[b];
        dec = a.logp[(size_t)b * a.ld + tok];
        return valid;
    };
    for (int e = tid; e < NE; e += SEL_NT) {
        int b, c, tok;
        float dec, ctc_rel, log_psi, v = -INFINITY;
        if (decode(e, b, c, tok, dec, ctc_rel, log_psi)) {
            float w = a.w_dec * dec;
            if (a.has_len) w = w + a.w_len;
            w = w + a.w_ctc * ctc_rel;
            v = w + a.score[b];
            if (!(v == v)) v = -INFINITY;
        }
        val[e] = v;
        keys[e] = f2key(v);
    }
    __syncthreads();
    unsigned thr;
    int n_gt, eq_take;
    radix_select(keys, NE, a.K, sc, thr, n_gt, eq_take);
    radix_compact(keys, NE, thr, n_gt, eq_take, sc, picked);
    __syncthreads();
    for (int r = tid; r < a.K; r += SEL_NT) {
        const int e = picked[r];
        const float v = val[e];
        int rank = 0;
        for (int j = 0; j < a.K; j++) {
            const int ej = picked[j];
            const float vj = val[ej];
            rank += vj > v || (vj == v && ej < e);
        }
        int b, c, tok;
        float dec, ctc_rel, log_psi;
        decode(e, b, c, tok, dec, ctc_rel, log_psi);
        a.sel[4 * rank + 0] = b;
        a.sel[4 * rank + 1] = tok;
        a.sel[4 * rank + 2] = c;
        a.sel[4 * rank + 3] = 0;
        a.selv[4 * rank + 0] = v;
        a.selv[4 * rank + 1] = dec;
        a.selv[4 * rank + 2] = ctc_rel;
        a.selv[4 * rank + 3] = log_psi;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Beam state (device): hypothesis-major token and ancestry tables + per-hypothesis scalars + the CTC forward variables.
struct BeamBuf {
    int64_t* yseq;  // [beam][ldy]
    int* anc;       // [beam][ldy]: slot of the hypothesis' ancestor at position j (the row its K / V were written to)
    int64_t* last;  // [beam] last token
    float* sc;      // [5][beam]: total, decoder sum, ctc sum, length sum, ctc log prefix probability s
    float* r;       // [T][2][n] (pitch = current n)
};

// new hypothesis k = hypothesis prev[k] extended by tok[k]  (batch_beam_search.py:131-176 merge + select states)
__global__ __launch_bounds__(256) void beam_update_kernel(BeamBuf src, BeamBuf dst, int ldy, int beam, int L, int n_src, int K, int T, int S,
                                                          const int* __restrict__ sel, const float* __restrict__ selv,
                                                          const float* __restrict__ r_new, float* __restrict__ host_row) {
    const int k = blockIdx.x, tid = threadIdx.x;
    const int prev = sel[4 * k], tok = sel[4 * k + 1];
    int pos = sel[4 * k + 2];
    if (pos < 0) pos = S - 1;  // <eos> outside the candidates: the hypothesis ends here, its CTC state is never read
    for (int j = tid; j < L; j += 256) {
        dst.yseq[(size_t)k * ldy + j] = src.yseq[(size_t)prev * ldy + j];
        if (j < L - 1) dst.anc[(size_t)k * ldy + j] = src.anc[(size_t)prev * ldy + j];
    }
    if (tid == 0) {
        dst.yseq[(size_t)k * ldy + L] = tok;
        dst.anc[(size_t)k * ldy + L - 1] = prev;
        dst.last[k] = tok;
        const float total = selv[4 * k], dec = selv[4 * k + 1], ctc_rel = selv[4 * k + 2], log_psi = selv[4 * k + 3];
        const float s_dec = src.sc[1 * beam + prev] + dec, s_ctc = src.sc[2 * beam + prev] + ctc_rel, s_len = src.sc[3 * beam + prev] + 1.f;
        dst.sc[0 * beam + k] = total;
        dst.sc[1 * beam + k] = s_dec;
        dst.sc[2 * beam + k] = s_ctc;
        dst.sc[3 * beam + k] = s_len;
        dst.sc[4 * beam + k] = log_psi;
        float* h = host_row + 8 * k;
        h[0] = (float)tok;
        h[1] = (float)prev;
        h[2] = total;
        h[3] = s_dec;
        h[4] = s_ctc;
        h[5] = s_len;
        h[6] = 0.f;
        h[7] = 0.f;
    }
    // CTC forward variables of (prev, candidate column pos): r_new [T][2][n_src][S] -> dst.r [T][2][K]
    for (int i = tid; i < 2 * T; i += 256) dst.r[(size_t)i * K + k] = r_new[((size_t)i * n_src + prev) * S + pos];
}

// hypotheses keep.idx[0 .. keep.n) survive (ended ones are taken off the beam, batch_beam_search.py:178-206)
__global__ __launch_bounds__(256) void beam_keep_kernel(BeamBuf src, BeamBuf dst, int ldy, int beam, int L, int n_src, int T, IdxList keep) {
    const int k = blockIdx.x, tid = threadIdx.x, from = keep.idx[k];
    for (int j = tid; j < L; j += 256) {
        dst.yseq[(size_t)k * ldy + j] = src.yseq[(size_t)from * ldy + j];
        if (j < L - 1) dst.anc[(size_t)k * ldy + j] = src.anc[(size_t)from * ldy + j];
    }
    if (tid < 5) dst.sc[tid * beam + k] = src.sc[tid * beam + from];
    if (tid == 5) dst.last[k] = src.last[from];
    for (int i = tid; i < 2 * T; i += 256) dst.r[(size_t)i * keep.n + k] = src.r[(size_t)i * n_src + from];
}

__global__ void beam_init_kernel(BeamBuf st, int beam, int T, int sos, const float* __restrict__ r_init) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid == 0) {
        st.yseq[0] = sos;
        st.last[0] = sos;
        for (int i = 0; i < 5; i++) st.sc[i * beam] = 0.f;
    }
    if (tid < 2 * T) st.r[tid] = r_init[tid];
}

// ---------------------------------------------------------------------------------------------------------------------
struct Layer {
    const float *n1g, *n1b, *wqkv, *bqkv, *wo, *bo, *n2g, *n2b, *wq2, *bq2, *wkv2, *bkv2, *wo2, *bo2, *n3g, *n3b, *w1, *b1, *w2, *b2;
};

struct Session {
    int D, H, FF, V, nl, beam, S, sos, eos, blank, has_len;
    float w_dec, w_ctc, w_len, emb_scale, eps;
    const float *embed, *pe;
    int pe_rows;
    std::vector<Layer> layers;
    const float *ang, *anb, *wout, *bout;
    // per utterance
    int T = 0, Lmax = 0, ldy = 0, ldv = 0, n = 0, L = 0, cur = 0;
    const float* ctc_logp = nullptr;
    int ld_ctc = 0;
    BeamBuf st[2];
    std::vector<float*> cache;   // per layer [Lmax][beam][3 D]: q | k | v of the position's row
    std::vector<float*> memkv;   // per layer [T][2 D]
    float *x, *x1, *x2, *h, *att, *q2, *ff, *part, *stx, *st1, *st2, *mean, *rstd, *logits, *logp, *lse, *psi, *psi_eos, *r_new, *selv, *host_dev;
    int64_t* cand;
    int* sel;
};

struct Carver {
    char* base;
    size_t off = 0;
    template <class T> T* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

void carve(Session& s, Carver& c, int T, int Lmax) {
    const size_t beam = s.beam, D = s.D;
    const int ldy = Lmax + 2, ldv = (s.V + 7) / 8 * 8;
    s.ldy = ldy;
    s.ldv = ldv;
    for (int i = 0; i < 2; i++) {
        s.st[i].yseq = c.take<int64_t>(beam * ldy);
        s.st[i].anc = c.take<int>(beam * ldy);
        s.st[i].last = c.take<int64_t>(beam);
        s.st[i].sc = c.take<float>(5 * beam);
        s.st[i].r = c.take<float>((size_t)2 * T * beam);
    }
    s.cache.resize(s.nl);
    s.memkv.resize(s.nl);
    for (int l = 0; l < s.nl; l++) {
        s.cache[l] = c.take<float>((size_t)Lmax * beam * 3 * D);
        s.memkv[l] = c.take<float>((size_t)T * 2 * D);
    }
    s.x = c.take<float>(beam * D);
    s.x1 = c.take<float>(beam * D);
    s.x2 = c.take<float>(beam * D);
    s.h = c.take<float>(beam * D);
    s.att = c.take<float>(beam * D);
    s.q2 = c.take<float>(beam * D);
    s.ff = c.take<float>(beam * (size_t)s.FF);
    s.part = c.take<float>((size_t)8 * beam * D);
    s.stx = c.take<float>(beam * (D / 16) * 2);
    s.st1 = c.take<float>(beam * (D / 16) * 2);
    s.st2 = c.take<float>(beam * (D / 16) * 2);  // K slices of the FFN's second contraction (at most 8)
    s.mean = c.take<float>(beam);
    s.rstd = c.take<float>(beam);
    s.logits = c.take<float>(beam * (size_t)ldv);
    s.logp = c.take<float>(beam * (size_t)ldv);
    s.lse = c.take<float>(beam);
    s.cand = c.take<int64_t>(beam * (size_t)s.S);
    s.psi = c.take<float>(beam * (size_t)s.S);
    s.psi_eos = c.take<float>(beam);
    s.r_new = c.take<float>((size_t)2 * T * beam * s.S);
    s.sel = c.take<int>(4 * beam);
    s.selv = c.take<float>(4 * beam);
    s.host_dev = c.take<float>(8 * beam);
}

int gemm(const float* A, int lda, const float* B, int M, int N, int K, const float* bias, int act, const float* resid, int ldr, float* C,
         int ldc, hipStream_t stream) {
    return avsr_gemm_f32s_nt(A, lda, B, K, M, N, K, bias, act, nullptr, 0, 0, 1.f, 0.f, 0, nullptr, 1.f, nullptr, resid, 0, ldr, C, 0, ldc,
                             0, 1, 0, nullptr, 0, nullptr, 0, stream);
}

// per-row (sum, sum of squares) partials left by the producer of a [M][D] activation for the LayerNorm that consumes it
struct RowStats {
    float* buf;  // [M][nt][2]
    int nt;
};

// C = act(LN?(A) W^T + bias) + resid for the <= beam rows of a decoding step (skinny16_kernel).  ln_g == NULL: no LayerNorm;
// otherwise `in` holds the statistics of A's rows.  out (may be NULL): receives the statistics of C's rows.
int skinny(const float* A, int lda, const float* W, int M, int N, int K, const float* bias, const float* ln_g, const float* ln_b, float eps,
           const RowStats* in, int act, const float* resid, int ldr, float* C, int ldc, RowStats* out, float* partial, hipStream_t stream) {
    // K slices across blocks when K exceeds the 16 waves x 64 columns of a block (the FFN's second contraction: K = 2048 /
    // 3072): partial sums, finished by rowsum_kernel
    auto ok_nw = [](int nw) { return nw >= 1 && nw <= 12 && (nw <= 4 || nw % 2 == 0); };  // the instantiated block sizes
    int Z = 1;
    while (Z <= 8 && !(K % (64 * Z) == 0 && ok_nw(K / (64 * Z)))) Z++;
    if (Z > 8 || (Z > 1 && (ln_g || act != 0 || !partial))) {
        avsr_set_error("beam_step: unsupported contraction length");
        return 1;
    }
    const int nw = K / (64 * Z);
    const dim3 grid((N + 15) / 16, (M + SK16_ROWS - 1) / SK16_ROWS, Z);
    SkinnyArgs a{A, lda, W, K, bias, ln_g, ln_b, eps, in ? in->buf : nullptr, in ? in->nt : 0, resid, ldr, C, ldc,
                 (out && Z == 1) ? out->buf : nullptr, M, N, K, act, Z, partial};
    if (out) out->nt = Z == 1 ? (int)grid.x : 1;
#define SKINNY_CASE(NW) AVSR_LAUNCH((skinny16_kernel<NW>), grid, dim3(64 * NW), (size_t)NW * SK16_ROWS * 17 * sizeof(float), stream, a)
    switch (nw) {
        case 1: SKINNY_CASE(1); break;
        case 2: SKINNY_CASE(2); break;
        case 3: SKINNY_CASE(3); break;
        case 4: SKINNY_CASE(4); break;
        case 6: SKINNY_CASE(6); break;
        case 8: SKINNY_CASE(8); break;
        case 10: SKINNY_CASE(10); break;
        case 12: SKINNY_CASE(12); break;
        default: avsr_set_error("beam_step: unsupported contraction length"); return 1;
    }
#undef SKINNY_CASE
    if (Z > 1)
        AVSR_LAUNCH(rowsum_kernel, dim3(M), dim3(256), 0, stream, (const float*)partial, Z, M, N, bias, resid, (long)ldr, C, (long)ldc,
                    out ? out->buf : (float*)nullptr);
    return 0;
}

}  // namespace

#define DEC_TRY(call)             \
    do {                          \
        const int rc__ = (call);  \
        if (rc__ != 0) return rc__; \
    } while (0)

// The linear layer of a decoding step on its own (tests, microbenchmarks): C = act(LN?(A) W^T + bias) + resid, M <= 128 rows.
// st_in [M][st_in_nt][2]: per-row (sum, sum of squares) partials of A (LN only); st_out [M][ceil(N / 16)][2] or NULL (one partial per 16-column block: size it from out->nt): the same
// for the rows of C; partial: >= 8 * M * N floats, needed when K > 768 (K slices).  Returns the number of partials per row of
// st_out through st_out_nt (may be NULL).
extern "C" int avsr_decode_linear(const float* A, int lda, const float* W, int M, int N, int K, const float* bias, const float* ln_g,
                                  const float* ln_b, float eps, const float* st_in, int st_in_nt, int act, const float* resid, int ldr,
                                  float* C, int ldc, float* st_out, int* st_out_nt, float* partial, hipStream_t stream) {
    RowStats in{const_cast<float*>(st_in), st_in_nt}, out{st_out, 0};
    const int rc = skinny(A, lda, W, M, N, K, bias, ln_g, ln_b, eps, ln_g ? &in : nullptr, act, resid, ldr, C, ldc, st_out ? &out : nullptr,
                          partial, stream);
    if (rc != 0) return rc;
    AVSR_CHECK_LAUNCH("decode_linear");
    if (st_out_nt) *st_out_nt = out.nt;
    return 0;
}

// cfg: D, H, FF, V, n_layers, beam, S (pre-beam size), sos, eos, blank, has_length_bonus, pe_rows
// fcfg: w_decoder, w_ctc, w_length_bonus, embedding scale (sqrt(D), embedding.py:84), LayerNorm eps
// w: embed [V][D], pe [pe_rows][D], then per layer 20 pointers in the order of `struct Layer` (wqkv = rows of linear_q, linear_k,
// linear_v stacked; wkv2 = source attention's linear_k, linear_v stacked), then after_norm gamma, beta, output_layer W [V][D], b
extern "C" int64_t avsr_beam_create(const int32_t* cfg, const float* fcfg, const void* const* w, int n_w) {
    const int nl = cfg[4];
    if (n_w != 2 + 20 * nl + 4 || cfg[0] % 64 != 0 || cfg[0] != 64 * cfg[1] || cfg[2] % 64 != 0 || cfg[5] < 2 || cfg[5] > MAX_BEAM ||
        cfg[6] < 1 || cfg[6] > cfg[3] || (int64_t)cfg[5] * (cfg[6] + 1) > 8192 || cfg[5] > cfg[6] - 1) {
        avsr_set_error("beam_create: unsupported configuration (d_k = 64, beam in [2, 128], beam <= pre-beam - 1, beam * (pre-beam + 1) <= 8192)");
        return 0;
    }
    Session* s = new (std::nothrow) Session();
    if (!s) return 0;
    s->D = cfg[0]; s->H = cfg[1]; s->FF = cfg[2]; s->V = cfg[3]; s->nl = nl; s->beam = cfg[5]; s->S = cfg[6];
    s->sos = cfg[7]; s->eos = cfg[8]; s->blank = cfg[9]; s->has_len = cfg[10]; s->pe_rows = cfg[11];
    s->w_dec = fcfg[0]; s->w_ctc = fcfg[1]; s->w_len = fcfg[2]; s->emb_scale = fcfg[3]; s->eps = fcfg[4];
    const float* const* p = reinterpret_cast<const float* const*>(w);
    s->embed = p[0];
    s->pe = p[1];
    s->layers.resize(nl);
    for (int l = 0; l < nl; l++) memcpy(&s->layers[l], p + 2 + 20 * l, sizeof(Layer));
    static_assert(sizeof(Layer) == 20 * sizeof(void*), "Layer is 20 pointers");
    s->ang = p[2 + 20 * nl];
    s->anb = p[3 + 20 * nl];
    s->wout = p[4 + 20 * nl];
    s->bout = p[5 + 20 * nl];
    return (int64_t)(intptr_t)s;
}

extern "C" int avsr_beam_destroy(int64_t h) {
    delete reinterpret_cast<Session*>((intptr_t)h);
    return 0;
}

extern "C" int64_t avsr_beam_workspace_bytes(int64_t h, int T, int Lmax) {
    Session tmp = *reinterpret_cast<Session*>((intptr_t)h);
    Carver c{nullptr};
    carve(tmp, c, T, Lmax);
    return (int64_t)c.off + 256;
}

// New utterance: memory [T][D] f32 (encoder output), ctc_logp [T][ld_ctc] f32 log-softmax of the CTC head, r_init [T][2] the
// CTC state of the empty prefix (ctc_prefix_score.py:60-66).  Projects the memory's K / V for every layer, resets the beam to <sos>.
extern "C" int avsr_beam_begin(int64_t h, const float* memory, int T, const float* ctc_logp, int ld_ctc, const float* r_init, void* ws,
                               int64_t ws_bytes, int Lmax, hipStream_t stream) {
    Session& s = *reinterpret_cast<Session*>((intptr_t)h);
    AVSR_REQUIRE(T >= 1 && Lmax >= 1 && Lmax + 1 <= s.pe_rows, "beam_begin: bad lengths (position table too short?)");
    Carver c{reinterpret_cast<char*>(ws)};
    carve(s, c, T, Lmax);
    AVSR_REQUIRE((int64_t)c.off <= ws_bytes, "beam_begin: workspace too small");
    s.T = T;
    s.Lmax = Lmax;
    s.ctc_logp = ctc_logp;
    s.ld_ctc = ld_ctc;
    s.n = 1;
    s.L = 1;
    s.cur = 0;
    for (int l = 0; l < s.nl; l++)
        DEC_TRY(gemm(memory, s.D, s.layers[l].wkv2, T, 2 * s.D, s.D, s.layers[l].bkv2, 0, nullptr, 0, s.memkv[l], 2 * s.D, stream));
    AVSR_LAUNCH(beam_init_kernel, dim3((2 * T + 255) / 256), dim3(256), 0, stream, s.st[0], s.beam, T, s.sos, r_init);
    AVSR_CHECK_LAUNCH("beam_begin");
    return 0;
}

// One decoding step for the n running hypotheses (all of length L): decoder pass over the new position, pre-beam, CTC
// prefix scores, top-K, new beam state.  host_out [K][8] f32: token, parent, total score, decoder / ctc / length-bonus
// sums, 0, 0 -- valid on return (the call synchronises the stream).  Returns K through n_out.
extern "C" int avsr_beam_step(int64_t h, float* host_out, int* n_out, hipStream_t stream) {
    Session& s = *reinterpret_cast<Session*>((intptr_t)h);
    AVSR_REQUIRE(s.n >= 1 && s.L <= s.Lmax, "beam_step: no running hypotheses / maximum length reached");
    const int n = s.n, L = s.L, D = s.D, beam = s.beam;
    BeamBuf& st = s.st[s.cur];
    BeamBuf& nx = s.st[s.cur ^ 1];
    const float scale = 1.0f / sqrtf(64.f);
    // embedding of the last token at position L - 1 (transformer_decoder.py:186-189, embedding.py:78-87) + its row statistics
    RowStats sx{s.stx, 1}, s1{s.st1, 0}, s2{s.st2, 0};
    AVSR_LAUNCH(dec_embed_kernel, dim3(n), dim3(256), 0, stream, (const int64_t*)st.last, s.embed, s.pe + (size_t)(L - 1) * D, s.emb_scale, D,
                s.x, sx.buf);
    float* x = s.x;
    for (int l = 0; l < s.nl; l++) {
        const Layer& w = s.layers[l];
        float* row = s.cache[l] + (size_t)(L - 1) * beam * 3 * D;  // this position's q | k | v rows, slot b = hypothesis b
        DEC_TRY(skinny(x, D, w.wqkv, n, 3 * D, D, w.bqkv, w.n1g, w.n1b, s.eps, &sx, 0, nullptr, 0, row, 3 * D, nullptr, nullptr, stream));
        const int wv_self = L <= 64 ? 4 : (L <= 256 ? 8 : 16), wv_src = s.T <= 64 ? 4 : (s.T <= 256 ? 8 : 16);  // waves per attention block
        AVSR_LAUNCH(dec_attn_kernel, dim3(n, s.H), dim3(64 * wv_self), (size_t)(L + 64 * wv_self) * sizeof(float), stream, (const float*)row, (long)3 * D,
                    (const float*)s.cache[l], (long)beam * 3 * D, (long)3 * D, D, 2 * D, (const int*)st.anc, s.ldy, L, scale, s.att, (long)D);
        DEC_TRY(skinny(s.att, D, w.wo, n, D, D, w.bo, nullptr, nullptr, 0.f, nullptr, 0, x, D, s.x1, D, &s1, nullptr, stream));
        DEC_TRY(skinny(s.x1, D, w.wq2, n, D, D, w.bq2, w.n2g, w.n2b, s.eps, &s1, 0, nullptr, 0, s.q2, D, nullptr, nullptr, stream));
        AVSR_LAUNCH(dec_src_attn_kernel, dim3((n + SRC_HB - 1) / SRC_HB, s.H), dim3(64 * wv_src), (size_t)SRC_HB * (s.T + 64 * wv_src) * sizeof(float), stream,
                    (const float*)s.q2, (long)D, (const float*)s.memkv[l], (long)2 * D, D, n, s.T, scale, s.att, (long)D);
        DEC_TRY(skinny(s.att, D, w.wo2, n, D, D, w.bo2, nullptr, nullptr, 0.f, nullptr, 0, s.x1, D, s.x2, D, &s2, nullptr, stream));
        DEC_TRY(skinny(s.x2, D, w.w1, n, s.FF, D, w.b1, w.n3g, w.n3b, s.eps, &s2, 1, nullptr, 0, s.ff, s.FF, nullptr, nullptr, stream));
        DEC_TRY(skinny(s.ff, s.FF, w.w2, n, D, s.FF, w.b2, nullptr, nullptr, 0.f, nullptr, 0, s.x2, D, s.x, D, &sx, s.part, stream));
    }
    DEC_TRY(skinny(x, D, s.wout, n, s.V, D, s.bout, s.ang, s.anb, s.eps, &sx, 0, nullptr, 0, s.logits, s.ldv, nullptr, nullptr, stream));
    AVSR_LAUNCH(logsoftmax_prebeam_kernel, dim3(n), dim3(SEL_NT), (size_t)(s.V + s.S) * sizeof(unsigned), stream, (const float*)s.logits, s.logp,
                (long)s.ldv, s.V, s.S, s.cand);
    DEC_TRY(avsr_ctc_prefix_score(s.ctc_logp, s.T, s.V, s.ld_ctc, st.r, st.last, s.cand, n, s.S, L - 1, s.blank, s.r_new, s.psi,
                                  s.psi_eos, stream));
    const int NE = n * (s.S + 1);
    const int K = beam;  // n * V >= beam always; the viable entries n * (S - 1) >= beam by the create-time check
    SelectArgs a{s.logp, (long)s.ldv, s.cand, s.psi, s.psi_eos, st.sc + 4 * beam, st.sc, n, s.S, K, s.eos, s.blank, s.has_len,
                 s.w_dec, s.w_ctc, s.w_len, s.sel, s.selv};
    AVSR_LAUNCH(beam_select_kernel, dim3(1), dim3(SEL_NT), (size_t)(2 * NE + K + n * s.S + n) * 4, stream, a);
    AVSR_LAUNCH(beam_update_kernel, dim3(K), dim3(256), 0, stream, st, nx, s.ldy, beam, L, n, K, s.T, s.S, (const int*)s.sel,
                (const float*)s.selv, (const float*)s.r_new, s.host_dev);
    AVSR_CHECK_LAUNCH("beam_step");
    const int e = avsr_copy_to_host_sync(host_out, s.host_dev, (size_t)K * 8 * sizeof(float), stream);
    if (e != 0) {
        avsr_set_error2("beam_step", hipGetErrorString((hipError_t)e));
        return 2;
    }
    s.cur ^= 1;
    s.n = K;
    s.L = L + 1;
    *n_out = K;
    return 0;
}

// take the hypotheses NOT listed off the beam (keep: ascending indices into the current beam)
extern "C" int avsr_beam_keep(int64_t h, const int32_t* keep, int n_keep, hipStream_t stream) {
    Session& s = *reinterpret_cast<Session*>((intptr_t)h);
    AVSR_REQUIRE(n_keep >= 0 && n_keep <= s.n, "beam_keep: bad count");
    if (n_keep == s.n) return 0;
    if (n_keep > 0) {
        IdxList il;
        il.n = n_keep;
        for (int i = 0; i < n_keep; i++) {
            AVSR_REQUIRE(keep[i] >= 0 && keep[i] < s.n, "beam_keep: index out of range");
            il.idx[i] = keep[i];
        }
        AVSR_LAUNCH(beam_keep_kernel, dim3(n_keep), dim3(256), 0, stream, s.st[s.cur], s.st[s.cur ^ 1], s.ldy, s.beam, s.L, s.n, s.T, il);
        AVSR_CHECK_LAUNCH("beam_keep");
        s.cur ^= 1;
    }
    s.n = n_keep;
    return 0;
}

// token sequences of the current beam: host_yseq [n][ldy] int64, the first L entries of every row valid (synchronises);
// returns the row pitch through ldy_out and the length through L_out
extern "C" int avsr_beam_fetch_yseq(int64_t h, int64_t* host_yseq, int* ldy_out, int* L_out, hipStream_t stream) {
    Session& s = *reinterpret_cast<Session*>((intptr_t)h);
    const int e = avsr_copy_to_host_sync(host_yseq, s.st[s.cur].yseq, (size_t)s.n * s.ldy * sizeof(int64_t), stream);
    if (e != 0) {
        avsr_set_error2("beam_fetch_yseq", hipGetErrorString((hipError_t)e));
        return 2;
    }
    *ldy_out = s.ldy;
    *L_out = s.L;
    return 0;
}
