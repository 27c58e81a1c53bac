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
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partial, int nparts, float max_norm,
                                                        float base_lr, float warmup_steps, float total_steps,
                                                        float* __restrict__ state) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += (double)partial[i];
    // wave reduction in double through two float halves is overkill: shuffle the double directly
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
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
    const float step = state[0], lr = state[1], coef = state[3];
    const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
    const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2), decay = 1.f - lr * weight_decay;
    const bool aligned = ((((uintptr_t)e.p) | ((uintptr_t)e.g) | ((uintptr_t)e.m) | ((uintptr_t)e.v)) & 15) == 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const long i = base + (threadIdx.x + 256 * j) * 4;
        if (i >= e.numel) break;
        float p[4], g[4], m[4], v[4];
        const bool vec = aligned && i + 3 < e.numel;
        const int cnt = vec ? 4 : (int)((e.numel - i) < 4 ? (e.numel - i) : 4);
        if (vec) {
            const f32x4 pp = *reinterpret_cast<const f32x4*>(e.p + i), gg = *reinterpret_cast<const f32x4*>(e.g + i);
            const f32x4 mm = *reinterpret_cast<const f32x4*>(e.m + i), vv = *reinterpret_cast<const f32x4*>(e.v + i);
#pragma unroll
            for (int k = 0; k < 4; k++) { p[k] = pp[k]; g[k] = gg[k]; m[k] = mm[k]; v[k] = vv[k]; }
        } else {
            for (int k = 0; k < cnt; k++) { p[k] = e.p[i + k]; g[k] = e.g[i + k]; m[k] = e.m[i + k]; v[k] = e.v[i + k]; }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k >= cnt) break;
            const float gk = g[k] * coef;
            p[k] *= decay;
            m[k] = beta1 * m[k] + (1.f - beta1) * gk;
            v[k] = beta2 * v[k] + (1.f - beta2) * gk * gk;
            const float denom = sqrtf(v[k]) * inv_sqrt_bc2 + eps;
            p[k] -= step_size * (m[k] / denom);
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
    AVSR_LAUNCH(clip_coef_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial, total_blocks, max_grad_norm, base_lr,
                (float)warmup_steps, (float)total_steps, state);
    AVSR_LAUNCH(multi_adamw_kernel, dim3(total_blocks), dim3(256), 0, stream, t, n, (const float*)state, beta1, beta2, eps,
                weight_decay);
    AVSR_CHECK_LAUNCH("adamw_step");
    return 0;
}
