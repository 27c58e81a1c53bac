// optim.hip -- the optimizer step of the reference training loop as three multi-tensor launches:
//   global gradient norm  ->  clip coefficient (+ step counter, learning rate)  ->  AdamW on every parameter.
//
// Replaces, for all 671 parameter tensors at once (lightning.py:48-52, train.py:41, cosine.py:6-25):
//   torch.nn.utils.clip_grad_norm_(params, 10.0)            Trainer(gradient_clip_val=10.0)
//   torch.optim.AdamW(lr, betas=(0.9, 0.98), weight_decay)  .step()
//   WarmupCosineScheduler.step()                            (per optimizer step)
// The foreach implementation behind those calls is ~10 elementwise launches per tensor group and streams the 1 GB of
// parameters / 1 GB of gradients / 2 GB of moments several times; here every element is read once and written once
// (p, m, v) -- the step is HBM-bound: 16 B read + 12 B written per parameter.
// Everything that changes from step to step (step count, learning rate, clip coefficient) lives in device memory and
// is produced on the device, so the step is capturable in a hipGraph and needs no host synchronisation.
#include <math.h>
#include "prims.h"
#include "avsr_hip.h"

namespace {

constexpr int OPT_CHUNK = 4096;  // elements per block: 256 threads x 4 float4

struct OptEntry {  // 48 bytes
    float* p;
    const float* g;
    float* m;
    float* v;
    long numel;
    int blk0, pad;
};

AVSR_DEV const OptEntry& find_entry(const OptEntry* table, int n, int blk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {  // last entry with blk0 <= blk
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].blk0 <= blk) lo = mid; else hi = mid - 1;
    }
    return table[lo];
}

// per-step scalars of the AdamW update, derived from the device-resident state
struct AdamScalars {
    float coef, decay, step_size, inv_sqrt_bc2;
};
AVSR_DEV AdamScalars adam_scalars(const float* state, float beta1, float beta2, float weight_decay) {
    const float step = state[0], lr = state[1];
    const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
    return AdamScalars{state[3], 1.f - lr * weight_decay, lr / bc1, 1.f / sqrtf(bc2)};
}
// torch.optim.AdamW (decoupled weight decay, no amsgrad) on one element, g := coef * grad
AVSR_DEV void adam_update(float& p, float g, float& m, float& v, const AdamScalars& s, float beta1, float beta2, float eps) {
    const float gk = g * s.coef;
    p *= s.decay;
    m = beta1 * m + (1.f - beta1) * gk;
    v = beta2 * v + (1.f - beta2) * gk * gk;
    const float denom = sqrtf(v) * s.inv_sqrt_bc2 + eps;
    p -= s.step_size * (m / denom);
}

// dst_i[0 .. numel_i) = scale * src_i[...] for every table entry {p = dst, g = src, m = optional bf16 dst, numel, blk0} (v unused):
// the gradients of one data-parallel bucket gathered into the bucket's flat buffer (and pre-divided by the world size) in ONE launch.
// m != NULL (round 5, the bf16 wire format): the scaled gradient goes to the bf16 buffer m INSTEAD of p -- the bucket travels as
// bf16 and is widened into p after the all-reduce, so the f32 image of the local gradients and the cast pass over it are skipped.
// Gradients may be dword-aligned only (views into other buffers): scalar accesses at the ragged ends.
__global__ __launch_bounds__(256) void multi_copy_scale_kernel(const OptEntry* __restrict__ table, int n, float scale) {
    const OptEntry e = find_entry(table, n, blockIdx.x);
    const long base = (long)(blockIdx.x - e.blk0) * OPT_CHUNK;
    const bool vec = (((uintptr_t)e.p | (uintptr_t)e.g) & 15) == 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const long i = base + (threadIdx.x + 256 * j) * 4;
        if (i >= e.numel) break;
        bf16_t* nb = reinterpret_cast<bf16_t*>(e.m);
        if (vec && i + 4 <= e.numel) {
            f32x4 v = *reinterpret_cast<const f32x4*>(e.g + i);
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] *= scale;
            if (nb) {
                bf16x4 o;
#pragma unroll
                for (int k = 0; k < 4; k++) o[k] = (short)f2bf(v[k]);
                *reinterpret_cast<bf16x4*>(nb + i) = o;  // (bucket slices start 16-byte aligned in f32: 8-byte aligned here)
            } else {
                *reinterpret_cast<f32x4*>(e.p + i) = v;
            }
        } else {
            for (int k = 0; k < 4 && i + k < e.numel; k++) {
                if (nb) nb[i + k] = f2bf(scale * e.g[i + k]);
                else e.p[i + k] = scale * e.g[i + k];
            }
        }
    }
}

// partial[block] = sum of squares of the block's chunk of gradients
__global__ __launch_bounds__(256) void multi_sumsq_kernel(const OptEntry* __restrict__ table, int n, float* __restrict__ partial) {
    __shared__ float red[4];
    const OptEntry e = find_entry(table, n, blockIdx.x);
    const long base = (long)(blockIdx.x - e.blk0) * OPT_CHUNK;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const long i = base + (threadIdx.x + 256 * j) * 4;
        if (i + 3 < e.numel && (((uintptr_t)(e.g + i)) & 15) == 0) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(e.g + i);
            s += g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + g[3] * g[3];
        } else {
            for (int k = 0; k < 4; k++)
                if (i + k < e.numel) s += e.g[i + k] * e.g[i + k];
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// state[0] = step (incremented here), state[1] = lr of this step, state[2] = gradient norm, state[3] = clip coefficient
// lr = base_lr * (step < warmup ? step / warmup : 0.5 (1 + cos(pi (step - warmup) / (total - warmup))))   cosine.py:20-25
__global__ __launch_bounds__(1024) void clip_coef_kernel(const float* __restrict__ partial, int nparts, float max_norm,
                                                        float base_lr, float warmup_steps, float total_steps,
                                                        float* __restrict__ state) {
    __shared__ double red[16];
    // ~61 000 partials for the 250 M-parameter model, ONE block (the result is a scalar): every trip of the loop is a
    // memory round trip, so the block is 1024 threads wide with eight independent loads per thread and trip -- 8 trips
    // instead of the 239 of a 256-thread single-load loop (84 us of a 22 ms step)
    double acc[8];
#pragma unroll
    for (int u = 0; u < 8; u++) acc[u] = 0.0;
    for (int i0 = threadIdx.x; i0 < nparts; i0 += 1024 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = (i0 + 1024 * u < nparts) ? partial[i0 + 1024 * u] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; u++) acc[u] += (double)v[u];
    }
    double s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    // wave reduction in double through two float halves is overkill: shuffle the double directly
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; w++) tot += red[w];
        const float norm = (float)sqrt(tot);
        const float step = state[0] + 1.f;
        float f;
        if (total_steps <= 0.f) f = 1.f;  // constant learning rate
        else if (step < warmup_steps) f = step / warmup_steps;
        else f = 0.5f * (1.f + cosf(3.14159265358979323846f * (step - warmup_steps) / (total_steps - warmup_steps)));
        state[0] = step;
        state[1] = base_lr * f;
        state[2] = norm;
        // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1; max_norm <= 0 disables clipping
        const float c = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.f;
        state[3] = c < 1.f ? c : 1.f;
    }
}

// torch.optim.AdamW (decoupled weight decay, no amsgrad) with g := coef * grad
__global__ __launch_bounds__(256) void multi_adamw_kernel(const OptEntry* __restrict__ table, int n,
                                                          const float* __restrict__ state, float beta1, float beta2,
                                                          float eps, float weight_decay) {
    const OptEntry e = find_entry(table, n, blockIdx.x);
    const long base = (long)(blockIdx.x - e.blk0) * OPT_CHUNK;
    const AdamScalars sc = adam_scalars(state, beta1, beta2, weight_decay);
    // p, m, v are allocations of their own; the gradient may be a view at any dword offset (DDP gradient buckets pack
    // parameters back to back, and e.g. a 5049-element bias shifts everything behind it by 4 bytes)
    const bool aligned = ((((uintptr_t)e.p) | ((uintptr_t)e.m) | ((uintptr_t)e.v)) & 15) == 0;
    const bool g_aligned = (((uintptr_t)e.g) & 15) == 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const long i = base + (threadIdx.x + 256 * j) * 4;
        if (i >= e.numel) break;
        float p[4], g[4], m[4], v[4];
        const bool vec = aligned && i + 3 < e.numel;
        const int cnt = vec ? 4 : (int)((e.numel - i) < 4 ? (e.numel - i) : 4);
        if (vec) {
            const f32x4 pp = *reinterpret_cast<const f32x4*>(e.p + i);
            const f32x4 mm = *reinterpret_cast<const f32x4*>(e.m + i), vv = *reinterpret_cast<const f32x4*>(e.v + i);
            f32x4 gg;
            if (g_aligned) gg = *reinterpret_cast<const f32x4*>(e.g + i);
            else gg = f32x4{e.g[i], e.g[i + 1], e.g[i + 2], e.g[i + 3]};
#pragma unroll
            for (int k = 0; k < 4; k++) { p[k] = pp[k]; g[k] = gg[k]; m[k] = mm[k]; v[k] = vv[k]; }
        } else {
            for (int k = 0; k < cnt; k++) { p[k] = e.p[i + k]; g[k] = e.g[i + k]; m[k] = e.m[i + k]; v[k] = e.v[i + k]; }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k >= cnt) break;
            adam_update(p[k], g[k], m[k], v[k], sc, beta1, beta2, eps);
        }
        if (vec) {
            *reinterpret_cast<f32x4*>(e.p + i) = f32x4{p[0], p[1], p[2], p[3]};
            *reinterpret_cast<f32x4*>(e.m + i) = f32x4{m[0], m[1], m[2], m[3]};
            *reinterpret_cast<f32x4*>(e.v + i) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
            for (int k = 0; k < cnt; k++) { e.p[i + k] = p[k]; e.m[i + k] = m[k]; e.v[i + k] = v[k]; }
        }
    }
}

// AdamW on 2-D weights whose bf16 operand copies the GEMMs read (gemm_fast.hip: AvsrCastEntry) are refreshed in the same
// pass: block = one 64x64 tile; the updated tile leaves as f32 (p, m, v), as bf16 [R][C] (dst) and, through LDS, as the
// transposed bf16 [C][ldT] (dstT, rows [R, limT) zero).  Saves the separate re-cast launch -- a second read of every
// f32 weight -- that an optimizer step otherwise forces before the next forward pass.
struct OptTileEntry {  // 80 bytes
    float* p;
    const float* g;
    float* m;
    float* v;
    bf16_t* dst;   // may be null
    bf16_t* dstT;  // may be null
    int R, C, ldT, blk0, tiles_c, limT;
    f16_t* dst16;  // may be null: two-plane IEEE-half [R][2][C] copy (forward operand of the mixed mode's f16 components)
};
static_assert(sizeof(OptTileEntry) == 80, "tile table rows are 80 bytes (auto_avsr_amd/optim.py)");

__global__ __launch_bounds__(256) void multi_adamw_cast_kernel(const OptTileEntry* __restrict__ table, int n,
                                                               const float* __restrict__ state, float beta1, float beta2,
                                                               float eps, float weight_decay) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 72];
    int lo = 0, hi = n - 1;
    while (lo < hi) {  // last entry with blk0 <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const OptTileEntry e = table[lo];
    const AdamScalars sc = adam_scalars(state, beta1, beta2, weight_decay);
    const int local = blockIdx.x - e.blk0;
    const int r0 = (local / e.tiles_c) * 64, c0 = (local % e.tiles_c) * 64;
    const bool vec_ok = (e.C % 8 == 0) && ((((uintptr_t)e.p) | ((uintptr_t)e.m) | ((uintptr_t)e.v)) & 15) == 0;
    const bool g_aligned = (((uintptr_t)e.g) & 15) == 0;  // the gradient may be a dword-aligned bucket view
    const int cc = (threadIdx.x & 7) * 8;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int r = (threadIdx.x >> 3) + 32 * half;
        const int gr = r0 + r, gc = c0 + cc;
        const long off = (long)gr * e.C + gc;
        float p[8], g[8], m[8], v[8];
        const bool vec = gr < e.R && vec_ok && gc + 8 <= e.C;
        const int cnt = gr < e.R ? (vec ? 8 : (e.C - gc < 8 ? (e.C - gc > 0 ? e.C - gc : 0) : 8)) : 0;
        if (vec) {
            load8(e.p + off, p);
            if (g_aligned) load8(e.g + off, g);
            else
#pragma unroll
                for (int k = 0; k < 8; k++) g[k] = e.g[off + k];
            load8(e.m + off, m);
            load8(e.v + off, v);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const bool ok = k < cnt;
                p[k] = ok ? e.p[off + k] : 0.f;
                g[k] = ok ? e.g[off + k] : 0.f;
                m[k] = ok ? e.m[off + k] : 0.f;
                v[k] = ok ? e.v[off + k] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < cnt) adam_update(p[k], g[k], m[k], v[k], sc, beta1, beta2, eps);
        if (vec) {
            store8(e.p + off, p);
            store8(e.m + off, m);
            store8(e.v + off, v);
            if (e.dst) store8(e.dst + off, p);
            if (e.dst16) {  // two-plane f16 image [R][2][C] (prims.h f2h_lo)
                store8(e.dst16 + off + (long)gr * e.C, p);
                store8_lo(e.dst16 + off + (long)(gr + 1) * e.C, p);
            }
        } else {
            for (int k = 0; k < cnt; k++) {
                e.p[off + k] = p[k];
                e.m[off + k] = m[k];
                e.v[off + k] = v[k];
                if (e.dst) e.dst[off + k] = f2bf(p[k]);
                if (e.dst16) {
                    e.dst16[off + (long)gr * e.C + k] = f2h(p[k]);
                    e.dst16[off + (long)(gr + 1) * e.C + k] = f2h_lo(p[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) tile[(cc + k) * 72 + r] = f2bf(p[k]);  // rows/columns past the matrix are zero
    }
    __syncthreads();
    if (e.dstT) {
        for (int id = threadIdx.x; id < 64 * 8; id += 256) {
            const int c = id >> 3, rr = (id & 7) * 8;
            const int gc = c0 + c, gr = r0 + rr;
            if (gc < e.C && gr < (e.limT ? e.limT : e.ldT))
                *reinterpret_cast<bf16x8*>(e.dstT + (long)gc * e.ldT + gr) = *reinterpret_cast<const bf16x8*>(tile + c * 72 + rr);
        }
    }
}

}  // namespace

// table: n entries of 48 bytes {float* p, const float* g, float* m, float* v, int64 numel, int blk0, 0} in device
// memory, blk0 = running sum of ceil(numel / 4096); total_blocks = the final sum.
// state: 4 floats in device memory {step, lr, grad_norm, clip_coef}; step starts at 0 and is incremented here.
// partial: total_blocks floats of scratch.  total_steps <= 0: constant learning rate base_lr.
extern "C" int avsr_adamw_step(const void* table, int n, int total_blocks, float* partial, float* state, float base_lr,
                               float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                               int64_t warmup_steps, int64_t total_steps, hipStream_t stream) {
    if (n <= 0 || total_blocks <= 0) return 0;
    AVSR_REQUIRE(table && partial && state, "adamw_step: null argument");
    const OptEntry* t = reinterpret_cast<const OptEntry*>(table);
    AVSR_LAUNCH(multi_sumsq_kernel, dim3(total_blocks), dim3(256), 0, stream, t, n, partial);
    AVSR_LAUNCH(clip_coef_kernel, dim3(1), dim3(1024), 0, stream, (const float*)partial, total_blocks, max_grad_norm, base_lr,
                (float)warmup_steps, (float)total_steps, state);
    AVSR_LAUNCH(multi_adamw_kernel, dim3(total_blocks), dim3(256), 0, stream, t, n, (const float*)state, beta1, beta2, eps,
                weight_decay);
    AVSR_CHECK_LAUNCH("adamw_step");
    return 0;
}

// The step in two halves, for an optimizer SHARDED over the data-parallel ranks (optim.ShardedAdamW: every rank owns 1 / N of every
// gradient bucket after a reduce-scatter and updates only that slice of the flat parameter buffers -- lightning.py:48-52 /
// train.py:41 semantics, 1 / N of the optimizer's HBM traffic per rank).  The global gradient norm needs the other ranks' shares:
//   avsr_multi_sumsq   sumsq[0] = sum of squares of this rank's gradient slices (table as above; partial: total_blocks floats)
//   [all-reduce of the one float across the ranks -- the caller]
//   avsr_adamw_apply   clip coefficient + step counter + learning rate from that global sum, then AdamW on the table's slices
__global__ __launch_bounds__(1024) void sum_partials_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 1024) s += (double)partial[i];
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; w++) tot += red[w];
        out[0] = (float)tot;
    }
}
extern "C" int avsr_multi_sumsq(const void* table, int n, int total_blocks, float* partial, float* sumsq, hipStream_t stream) {
    AVSR_REQUIRE(table && partial && sumsq && n > 0 && total_blocks > 0, "multi_sumsq: bad arguments");
    AVSR_LAUNCH(multi_sumsq_kernel, dim3(total_blocks), dim3(256), 0, stream, reinterpret_cast<const OptEntry*>(table), n, partial);
    AVSR_LAUNCH(sum_partials_kernel, dim3(1), dim3(1024), 0, stream, (const float*)partial, total_blocks, sumsq);
    AVSR_CHECK_LAUNCH("multi_sumsq");
    return 0;
}
extern "C" int avsr_adamw_apply(const void* table, int n, int total_blocks, const float* sumsq, float* state, float base_lr,
                                float beta1, float beta2, float eps, float weight_decay, float max_grad_norm, int64_t warmup_steps,
                                int64_t total_steps, hipStream_t stream) {
    AVSR_REQUIRE(table && sumsq && state && n > 0 && total_blocks > 0, "adamw_apply: bad arguments");
    AVSR_LAUNCH(clip_coef_kernel, dim3(1), dim3(1024), 0, stream, sumsq, 1, max_grad_norm, base_lr, (float)warmup_steps,
                (float)total_steps, state);
    AVSR_LAUNCH(multi_adamw_kernel, dim3(total_blocks), dim3(256), 0, stream, reinterpret_cast<const OptEntry*>(table), n,
                (const float*)state, beta1, beta2, eps, weight_decay);
    AVSR_CHECK_LAUNCH("adamw_apply");
    return 0;
}

// The same step with the bf16 operand copies of the 2-D weights refreshed in the update pass:
//   table / n / total_blocks   every parameter (48-byte entries as above): gradient norm
//   lin_table / lin_n / lin_blocks   the parameters updated by the linear kernel (same entry format, own blk0 numbering)
//   tile_table / tile_n / tile_blocks   80-byte entries {p, g, m, v, dst, dstT (bf16, may be 0), int R, C, ldT, blk0,
//       tiles_c, limT, 0, 0}: weights [R][C] updated tile-wise, blk0 = running sum of
//       ceil(max(R, limT ? limT : ldT) / 64) * ceil(C / 64) as in avsr_multi_cast_transpose
// Every parameter must be in exactly one of lin_table / tile_table.
extern "C" int avsr_adamw_cast_step(const void* table, int n, int total_blocks, const void* lin_table, int lin_n,
                                    int lin_blocks, const void* tile_table, int tile_n, int tile_blocks, float* partial,
                                    float* state, float base_lr, float beta1, float beta2, float eps, float weight_decay,
                                    float max_grad_norm, int64_t warmup_steps, int64_t total_steps, hipStream_t stream) {
    if (n <= 0 || total_blocks <= 0) return 0;
    AVSR_REQUIRE(table && partial && state, "adamw_cast_step: null argument");
    AVSR_REQUIRE((lin_n <= 0 || lin_table) && (tile_n <= 0 || tile_table), "adamw_cast_step: null table");
    const OptEntry* t = reinterpret_cast<const OptEntry*>(table);
    AVSR_LAUNCH(multi_sumsq_kernel, dim3(total_blocks), dim3(256), 0, stream, t, n, partial);
    AVSR_LAUNCH(clip_coef_kernel, dim3(1), dim3(1024), 0, stream, (const float*)partial, total_blocks, max_grad_norm, base_lr,
                (float)warmup_steps, (float)total_steps, state);
    if (tile_n > 0 && tile_blocks > 0)
        AVSR_LAUNCH(multi_adamw_cast_kernel, dim3(tile_blocks), dim3(256), 0, stream,
                    reinterpret_cast<const OptTileEntry*>(tile_table), tile_n, (const float*)state, beta1, beta2, eps,
                    weight_decay);
    if (lin_n > 0 && lin_blocks > 0)
        AVSR_LAUNCH(multi_adamw_kernel, dim3(lin_blocks), dim3(256), 0, stream, reinterpret_cast<const OptEntry*>(lin_table),
                    lin_n, (const float*)state, beta1, beta2, eps, weight_decay);
    AVSR_CHECK_LAUNCH("adamw_cast_step");
    return 0;
}

// dst_i = scale * src_i for n tensors in one launch.  table: n entries in the layout of avsr_adamw_step ({dst, src, -, -, numel,
// blk0}: 48 bytes each, blk0 = running sum of ceil(numel / 4096)); total_blocks = the final sum.  Used by the data-parallel
// gradient exchange (auto_avsr_amd/ddp.py): one launch gathers a bucket's gradients into its flat all-reduce buffer.
extern "C" int avsr_multi_copy_scale(const void* table, int n, int total_blocks, float scale, hipStream_t stream) {
    if (n <= 0 || total_blocks <= 0) return 0;
    AVSR_LAUNCH(multi_copy_scale_kernel, dim3(total_blocks), dim3(256), 0, stream, reinterpret_cast<const OptEntry*>(table), n, scale);
    AVSR_CHECK_LAUNCH("multi_copy_scale");
    return 0;
}
