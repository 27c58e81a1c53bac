// prims.h -- device primitives shared by every kernel of libavsr_hip.so (gfx950 / CDNA4).
//
// The kernels are written once, for the 64-wide CDNA4 wavefront and its MFMA
// units.  The only conditional in the tree is the AVSR_EMU hook below: with
// -DAVSR_EMU the same sources compile against tests/emu/hip_emu.h, a host-side
// SIMT emulator used by the CPU unit tests to debug indexing.  The shipped
// library is always the hipcc --offload-arch=gfx950 build.
#pragma once

extern "C" void avsr_note_launch(int err);
extern "C" int avsr_take_launch_error(void);
#ifdef AVSR_EMU
#include "hip_emu.h"
#define AVSR_LAUNCH(kern, grid, block, smem, stream, ...) \
    emu::launch((grid), (block), (smem), [=]() { kern(__VA_ARGS__); })
#define AVSR_DYN_SMEM(name) char* name = emu::dyn_smem()
#else
#include <hip/hip_runtime.h>
// PyTorch shares this thread's HIP runtime and may leave benign non-sticky codes (hipErrorNotReady from
// event queries) in the per-thread "last error" slot: clear it, launch, and record only our own result.
#define AVSR_LAUNCH(kern, grid, block, smem, stream, ...)                          \
    do {                                                                           \
        (void)hipGetLastError();                                                   \
        hipLaunchKernelGGL(kern, (grid), (block), (smem), (stream), __VA_ARGS__);  \
        avsr_note_launch((int)hipGetLastError());                                  \
    } while (0)
#define AVSR_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#endif

#include <stdint.h>

#define AVSR_DEV __device__ __forceinline__
#define AVSR_WAVE 64

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------- bf16 <-> f32
AVSR_DEV float bf2f(bf16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    return __builtin_bit_cast(float, u);
}
AVSR_DEV bf16_t f2bf(float f) {  // round to nearest even, NaN preserved
#ifndef AVSR_EMU
    return __builtin_bit_cast(bf16_t, (__bf16)f);  // v_cvt_pk_bf16_f32 on gfx950
#endif
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// IEEE half (1-5-10) as a STORAGE type of forward activations: the "mixed" numerical mode keeps the activations of the
// Conformer encoder in f16 -- 11 significant bits against the 8 of bf16, the same 2 bytes and the same MFMA rate
// (v_mfma_f32_*_f16) -- because the north-star bound on the logits (1e-3 relative) is not reachable with 8-bit significands
// (tools/precision_study.py).  Values are clamped to the finite range on conversion (activations that feed a contraction sit
// behind a LayerNorm / BatchNorm / softmax: |x| << 65504); gradients never use this type.
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
AVSR_DEV float h2f(f16_t h) { return (float)h; }
AVSR_DEV f16_t f2h(float f) {  // round to nearest even, saturating
#ifdef AVSR_EMU
    f = f > 65504.f ? 65504.f : (f < -65504.f ? -65504.f : f);
#else
    f = __builtin_amdgcn_fmed3f(f, -65504.f, 65504.f);
#endif
    return (f16_t)f;
}
AVSR_DEV short f2h_bits(float f) { return __builtin_bit_cast(short, f2h(f)); }
// Two-plane f16 image of a WEIGHT (round 5): hi = f16(w), lo = f16((w - hi) * 2^11).  The scale keeps the lo plane in the normal
// f16 range; consumers accumulate a * lo separately and fold it in as acc_lo * 2^-11 (gemm_fast_kernel.h, WP = 2).  Every f16
// forward copy of a weight is stored row-interleaved, [rows][2][K]: the hi row r at element 2 r K, its lo row K further on, so any
// row slice of a (concatenated) copy carries both planes and a one-plane consumer simply reads hi with pitch 2 K.
#define AVSR_H16_LO_SCALE 2048.0f
AVSR_DEV f16_t f2h_lo(float w) { return f2h((w - h2f(f2h(w))) * AVSR_H16_LO_SCALE); }

// Storage-type traits: activations live in HBM either as bf16 (bench mode), as f16 (forward pass of the mixed mode) or
// as f32 (parity mode, where GEMM operands are split into hi+lo bf16 halves).
template <class T> struct Elem;
template <> struct Elem<float> {
    static AVSR_DEV float ld(const float* p) { return *p; }
    static AVSR_DEV void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static AVSR_DEV float ld(const bf16_t* p) { return bf2f(*p); }
    static AVSR_DEV void st(bf16_t* p, float v) { *p = f2bf(v); }
};
template <> struct Elem<f16_t> {
    static AVSR_DEV float ld(const f16_t* p) { return h2f(*p); }
    static AVSR_DEV void st(f16_t* p, float v) { *p = f2h(v); }
};

// 8 consecutive elements -> 8 floats (16-byte / 32-byte vector loads)
AVSR_DEV void load8(const bf16_t* p, float* out) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = bf2f((bf16_t)v[i]);
}
AVSR_DEV void load8(const f16_t* p, float* out) {
    f16x8 v = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = (float)v[i];
}
AVSR_DEV void load8(const float* p, float* out) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        out[i] = a[i];
        out[i + 4] = b[i];
    }
}
AVSR_DEV void store8(bf16_t* p, const float* v) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = (short)f2bf(v[i]);
    *reinterpret_cast<bf16x8*>(p) = o;
}
AVSR_DEV void store8(f16_t* p, const float* v) {
    f16x8 o;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = f2h(v[i]);
    *reinterpret_cast<f16x8*>(p) = o;
}
AVSR_DEV void store8_lo(f16_t* p, const float* v) {  // the scaled lo plane of 8 weights
    float l[8];
#pragma unroll
    for (int i = 0; i < 8; i++) l[i] = (v[i] - h2f(f2h(v[i]))) * AVSR_H16_LO_SCALE;
    store8(p, l);
}
AVSR_DEV void store8(float* p, const float* v) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[i] = v[i];
        b[i] = v[i + 4];
    }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}

// ---------------------------------------------------------------- async global -> LDS (LDS-DMA)
// global_load_lds_dwordx4: lane l copies 16 bytes from its own global address to (wave-uniform LDS base + 16*l).
// The transfer is tracked by vmcnt; ordering against ds_read is the caller's job (counted s_waitcnt + barrier).
AVSR_DEV void glds16(const void* gptr, void* lds_wave_base) {
#ifdef AVSR_EMU
    memcpy(reinterpret_cast<char*>(lds_wave_base) + 16 * emu::lane_id(), gptr, 16);
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}
AVSR_DEV void block_barrier_raw() {  // s_barrier without the vmcnt(0) drain that __syncthreads() implies
#ifdef AVSR_EMU
    emu::sync_threads();
#else
    __builtin_amdgcn_s_barrier();
#endif
}
// wave index inside the block as a SCALAR (the compiler cannot see that threadIdx.x >> 6 is wave-uniform; with the
// scalar form per-wave guards become scalar branches and LDS-DMA bases need no v_readfirstlane)
AVSR_DEV int wave_id() {
#ifdef AVSR_EMU
    return (int)(threadIdx.x >> 6);
#else
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#endif
}
// compiler scheduling fence: no instruction is moved across it (keeps a hand-placed prefetch ahead of the MFMAs)
AVSR_DEV void sched_fence() {
#ifndef AVSR_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}
template <int N> AVSR_DEV void wait_vmcnt() {
#ifndef AVSR_EMU
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

// ds_read_b64_tr_b16: LDS transpose read.  Inside each 16-lane group, lane i supplies the address of 4 consecutive
// bf16 = the piece (row i/4, columns 4*(i%4)..+3) of a 4x16 block; it receives column i of that block (rows 0..3).
// Verified on gfx950 (tools/probes/tr_probe.hip).  Lets an MFMA fragment (8 consecutive k for one m) be read from a
// [k][m]-major LDS tile, i.e. from operands stored with the contraction index as the SLOW dimension.
AVSR_DEV bf16x4 lds_tr16(const bf16_t* p) {
#ifdef AVSR_EMU
    bf16x4 mine = *reinterpret_cast<const bf16x4*>(p);
    size_t stride;
    const unsigned char* all = emu::wave_gather(&mine, sizeof(mine), &stride);
    const int l = emu::lane_id(), i = l & 15, g = l >> 4;
    bf16x4 out;
    for (int j = 0; j < 4; j++) {
        bf16x4 src;
        memcpy(&src, all + (size_t)(16 * g + 4 * j + (i >> 2)) * stride, sizeof(src));
        out[j] = src[i & 3];
    }
    return out;
#else
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p);
#endif
}

// The same read issued WITHOUT the compiler knowing it is an LDS access.  hipcc orders every LDS-address-space load
// behind all pending LDS-DMA writes (it inserts s_waitcnt vmcnt(0) in front of the builtin above whenever a
// global_load_lds is in flight -- even one that targets another ring slot), which serialises an operand ring.  The
// asm form keeps the DMA in flight; in exchange the CALLER orders the read: lds_wait<N>() (s_waitcnt lgkmcnt(N),
// N = number of later LDS operations that may still be outstanding) followed by lds_tie() on every register that
// must not be consumed before the wait.  Reads are asm volatile, so they issue in program order.
AVSR_DEV bf16x4 lds_tr16_async(const bf16_t* p) {
#ifdef AVSR_EMU
    return lds_tr16(p);
#else
    const uint32_t addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) bf16_t*)p;
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
#endif
}
// Plain 16- / 8-byte LDS reads in the same caller-ordered form (tables that live beside an LDS-DMA ring).
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
AVSR_DEV i32x4 lds_read16_async(const void* p) {
#ifdef AVSR_EMU
    i32x4 v;
    memcpy(&v, p, 16);
    return v;
#else
    const uint32_t addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
#endif
}
AVSR_DEV i32x2 lds_read8_async(const void* p) {
#ifdef AVSR_EMU
    i32x2 v;
    memcpy(&v, p, 8);
    return v;
#else
    const uint32_t addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
    i32x2 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
    return v;
#endif
}
template <int N> AVSR_DEV void lds_wait() {
#ifndef AVSR_EMU
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
#endif
}
template <class T> AVSR_DEV void lds_tie(T& v) {  // makes every later use of v depend on the preceding lds_wait
#ifndef AVSR_EMU
    asm volatile("" : "+v"(v));
#endif
}

// ---------------------------------------------------------------- math
AVSR_DEV float avsr_exp(float x) {
#ifdef AVSR_EMU
    return expf(x);
#else
    return __expf(x);
#endif
}
// 1 / (1 + e^-x) with the hardware reciprocal (v_rcp_f32, 1 ulp): an IEEE division costs ~10 VALU instructions, and the
// BatchNorm + SiLU passes evaluate this once or more per activation element (the fused stem BN + pool kernel 9 times per
// output) -- they were ALU-bound on it, not HBM-bound.
AVSR_DEV float avsr_sigmoid(float x) {
#ifdef AVSR_EMU
    return 1.0f / (1.0f + avsr_exp(-x));
#else
    return __builtin_amdgcn_rcpf(1.0f + avsr_exp(-x));
#endif
}
AVSR_DEV float avsr_silu(float x) { return x * avsr_sigmoid(x); }

// ---------------------------------------------------------------- neighbour exchange across the 64 lanes of a wave
// lane i <- lane i-1 (up) / lane i+1 (down); the lane that has no such neighbour receives `fill`.  DPP wave shifts
// (wave_shr:1 / wave_shl:1, verified on gfx950 with tools/probes/dpp_probe.hip) move through the VALU data path in one
// instruction; __shfl_up / __shfl_down go through ds_bpermute, an LDS-crossbar round trip of ~100 clocks -- on the critical
// path of every step of a serial recursion (CTC alpha / beta).
AVSR_DEV float wave_up1(float x, float fill) {
#ifdef AVSR_EMU
    const float v = __shfl_up(x, 1);
    return emu::lane_id() == 0 ? fill : v;
#else
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x138, 0xf, 0xf, false));
#endif
}
AVSR_DEV float wave_down1(float x, float fill) {
#ifdef AVSR_EMU
    const float v = __shfl_down(x, 1);
    return emu::lane_id() == 63 ? fill : v;
#else
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x130, 0xf, 0xf, false));
#endif
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
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

// ---------------------------------------------------------------- MFMA
// v_mfma_f32_32x32x16_bf16: A lane l holds row (l&31), k = 8*(l>>5)+e; B lane l
// holds col (l&31), same k; D reg r of lane l is row (r&3)+8*(r>>2)+4*(l>>5), col l&31.
AVSR_DEV f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef AVSR_EMU
    struct P { bf16x8 a, b; } mine{a, b};
    size_t stride;
    const unsigned char* all = emu::wave_gather(&mine, sizeof(P), &stride);
    const int l = emu::lane_id();
    const int j = l & 31;
    for (int r = 0; r < 16; r++) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int kb = 0; kb < 2; kb++) {
            P pa, pb;
            memcpy(&pa, all + (size_t)(i + 32 * kb) * stride, sizeof(P));
            memcpy(&pb, all + (size_t)(j + 32 * kb) * stride, sizeof(P));
            for (int e = 0; e < 8; e++) acc += bf2f((bf16_t)pa.a[e]) * bf2f((bf16_t)pb.b[e]);
        }
        c[r] = acc;
    }
    return c;
#else
    typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, a),
                                                   __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
#endif
}
// The same MFMA with the accumulator pinned to ARCHITECTURAL VGPRs.  The compiler selects the AGPR form of an MFMA in kernels
// that may use more than 256 registers and "spills" the accumulators that do not fit the 256 AGPRs by swapping them through
// VGPRs around every use (32 v_accvgpr moves per MFMA); a kernel with more than 16 accumulators names the overflow here.
// (Inline asm: the compiler inserts no hazard no-ops around it -- keep dependent uses of the result tens of cycles away;
// back-to-back MFMAs on different accumulators and MFMA -> MFMA chains on the same one are interlocked by hardware.)
AVSR_DEV f32x16 mfma32_vgpr(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef AVSR_EMU
    return mfma32(a, b, c);
#else
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    return c;
#endif
}
// v_mfma_f32_16x16x32_bf16: A lane l holds row (l&15), k = 8*(l>>4)+e; B lane l
// holds col (l&15); D reg r of lane l is row 4*(l>>4)+r, col l&15.
AVSR_DEV f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
#ifdef AVSR_EMU
    struct P { bf16x8 a, b; } mine{a, b};
    size_t stride;
    const unsigned char* all = emu::wave_gather(&mine, sizeof(P), &stride);
    const int l = emu::lane_id();
    const int j = l & 15;
    for (int r = 0; r < 4; r++) {
        const int i = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int kb = 0; kb < 4; kb++) {
            P pa, pb;
            memcpy(&pa, all + (size_t)(i + 16 * kb) * stride, sizeof(P));
            memcpy(&pb, all + (size_t)(j + 16 * kb) * stride, sizeof(P));
            for (int e = 0; e < 8; e++) acc += bf2f((bf16_t)pa.a[e]) * bf2f((bf16_t)pb.b[e]);
        }
        c[r] = acc;
    }
    return c;
#else
    typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a),
                                                   __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
#endif
}

// The f16 forms of the same two MFMAs (v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16: identical shapes, lane layouts and
// rate).  Fragments travel as raw 16-bit lanes (bf16x8) through LDS and registers -- staging is a byte copy either way -- so
// FMT only selects the instruction: 0 = bf16, 1 = f16.
template <int FMT> AVSR_DEV float raw16_to_f32(short bits) {
    return FMT == 1 ? (float)__builtin_bit_cast(f16_t, bits) : bf2f((bf16_t)bits);
}
template <int FMT> AVSR_DEV short f32_to_raw16(float v) { return FMT == 1 ? f2h_bits(v) : (short)f2bf(v); }
template <int FMT> AVSR_DEV f32x16 mfma32x(bf16x8 a, bf16x8 b, f32x16 c) {
    if (FMT == 0) return mfma32(a, b, c);
#ifdef AVSR_EMU
    struct P { bf16x8 a, b; } mine{a, b};
    size_t stride;
    const unsigned char* all = emu::wave_gather(&mine, sizeof(P), &stride);
    const int l = emu::lane_id();
    const int j = l & 31;
    for (int r = 0; r < 16; r++) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int kb = 0; kb < 2; kb++) {
            P pa, pb;
            memcpy(&pa, all + (size_t)(i + 32 * kb) * stride, sizeof(P));
            memcpy(&pb, all + (size_t)(j + 32 * kb) * stride, sizeof(P));
            for (int e = 0; e < 8; e++) acc += raw16_to_f32<1>(pa.a[e]) * raw16_to_f32<1>(pb.b[e]);
        }
        c[r] = acc;
    }
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#endif
}
template <int FMT> AVSR_DEV f32x4 mfma16x(bf16x8 a, bf16x8 b, f32x4 c) {
    if (FMT == 0) return mfma16(a, b, c);
#ifdef AVSR_EMU
    struct P { bf16x8 a, b; } mine{a, b};
    size_t stride;
    const unsigned char* all = emu::wave_gather(&mine, sizeof(P), &stride);
    const int l = emu::lane_id();
    const int j = l & 15;
    for (int r = 0; r < 4; r++) {
        const int i = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int kb = 0; kb < 4; kb++) {
            P pa, pb;
            memcpy(&pa, all + (size_t)(i + 16 * kb) * stride, sizeof(P));
            memcpy(&pb, all + (size_t)(j + 16 * kb) * stride, sizeof(P));
            for (int e = 0; e < 8; e++) acc += raw16_to_f32<1>(pa.a[e]) * raw16_to_f32<1>(pb.b[e]);
        }
        c[r] = acc;
    }
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#endif
}

// Split-precision operands.  NS = 1: plain bf16.  NS = 2: value = hi + lo with
// both halves bf16 (16 mantissa bits); a product uses the three significant
// cross terms, giving ~2^-16 relative error -- the "parity mode" of DESIGN.md.
template <int NS> struct Frag { bf16x8 p[NS]; };

template <int NS>
AVSR_DEV f32x16 mma32(const Frag<NS>& a, const Frag<NS>& b, f32x16 c) {
    if (NS == 2) {
        c = mfma32(a.p[NS - 1], b.p[0], c);  // lo*hi
        c = mfma32(a.p[0], b.p[NS - 1], c);  // hi*lo
    }
    return mfma32(a.p[0], b.p[0], c);
}
template <int NS>
AVSR_DEV f32x4 mma16(const Frag<NS>& a, const Frag<NS>& b, f32x4 c) {
    if (NS == 2) {
        c = mfma16(a.p[NS - 1], b.p[0], c);
        c = mfma16(a.p[0], b.p[NS - 1], c);
    }
    return mfma16(a.p[0], b.p[0], c);
}

// split a float into NS bf16 planes
template <int NS> AVSR_DEV void split_bf16(float x, bf16_t* out) {
    bf16_t h = f2bf(x);
    out[0] = h;
    if (NS == 2) out[NS - 1] = f2bf(x - bf2f(h));
}

// "split8" as a STORAGE type of an activation (round 5): an f32-sized tensor whose every group of 8 consecutive elements holds
// the 8 hi bf16 followed by the 8 lo bf16 of the values it stands for (the layout csrc/gemm_split.hip consumes without a
// conversion pass: tiles with ACV = 2).  The producers are element-wise passes (BatchNorm + activation), the other consumers read
// it back as hi + lo (16 significant bits: what the split-plane contraction uses of the value anyway).  sizeof == 4, so pointer
// arithmetic in elements is that of the f32 tensor; only whole groups of 8 are addressable.
struct sp8_t { float raw; };
AVSR_DEV void load8(const sp8_t* p, float* out) {
    const bf16x8 hi = *reinterpret_cast<const bf16x8*>(p);
    const bf16x8 lo = *(reinterpret_cast<const bf16x8*>(p) + 1);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = bf2f((bf16_t)hi[i]) + bf2f((bf16_t)lo[i]);
}
AVSR_DEV void store8(sp8_t* p, const float* v) {
    bf16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        bf16_t pl[2];
        split_bf16<2>(v[i], pl);
        hi[i] = (short)pl[0];
        lo[i] = (short)pl[1];
    }
    *reinterpret_cast<bf16x8*>(p) = hi;
    *(reinterpret_cast<bf16x8*>(p) + 1) = lo;
}
template <> struct Elem<sp8_t> {  // (single elements of a split8 tensor are not addressable: callers keep counts multiples of 8)
    static AVSR_DEV float ld(const sp8_t*) { return 0.f; }
    static AVSR_DEV void st(sp8_t*, float) {}
};

// ---------------------------------------------------------------- stateless dropout RNG
// keep-mask for element `idx` of a tensor under (seed, p): a 32-bit mix of the 64-bit (seed, idx) counter; forward and
// backward recompute the same bits.
// Cost matters: the attention kernels draw one number per score element, the GEMM epilogues one per output.  A 64-bit
// splitmix (three 64-bit multiplies = ~9 quarter-rate 32-bit multiplies + carries per element) made the RNG ~8x the MFMA
// time of the attention inner loop.  Now the seed -- wave-uniform in every caller -- goes through the 64-bit mixer ONCE into
// a 32-bit key (scalar ALU, hoisted out of the element loops), and the per-element work is a 32-bit finalizer with two
// multiplies ("lowbias32", Wellons: bias 0.17 over all 2^32 inputs) of (low index word ^ key); the high index word
// (non-zero only beyond 2^32 elements) costs one more round.
AVSR_DEV uint32_t avsr_hash_key(uint64_t seed) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}
AVSR_DEV uint32_t avsr_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
AVSR_DEV uint32_t avsr_hash(uint64_t seed, uint64_t idx) {
    uint32_t x = avsr_mix32((uint32_t)idx ^ avsr_hash_key(seed));
    const uint32_t hi = (uint32_t)(idx >> 32);
    if (hi) x = avsr_mix32(x + hi * 0x9E3779B1u);
    return x;
}
// returns scale to apply: 0 if dropped, 1/(1-p) if kept; p == 0 -> 1
AVSR_DEV float dropout_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
    if (p <= 0.f) return 1.f;
    const float u = (float)(avsr_hash(seed, idx) >> 8) * (1.0f / 16777216.0f);
    return u < p ? 0.f : inv_keep;
}

// ---------------------------------------------------------------- status plumbing
extern int avsr_tune_knobs[32];  // common.hip (avsr_tune)
// Deterministic mode (avsr_tune knob 23 = 1; AVSR_DETERMINISTIC=1 / train.py --deterministic, round 6): every sum that the default
// build forms with floating-point atomics from SEVERAL blocks -- split-K weight gradients, bias / LayerNorm / depthwise parameter
// gradients, the position-projection gradient -- is formed by ONE block per output element in a fixed order instead (grid
// policies of the entry points: no k split, one block per column group) or by an ordered column-sum pass over the stored
// values (avsr_colsum_det); DESIGN.md lists the sites.  Two runs then give bit-identical losses and weights; it costs speed.
inline bool avsr_det() { return avsr_tune_knobs[23] != 0; }
// out[c] += sum_r src[r * ld + c] for c < cols, in a fixed order (one block per 256 columns): src f32 / bf16 / f16 (dtype 0 / 1 / 2)
int avsr_colsum_det(const void* src, int dtype, long ld, long rows, int cols, float* out, hipStream_t stream);
// out[r] += sum_c src[r * ld + c] for r < rows (bf16 source), fixed order
int avsr_rowsum_det_bf16(const void* src, long ld, long rows, long cols, float* out, hipStream_t stream);
extern "C" void avsr_set_error(const char* msg);
extern "C" void avsr_set_error2(const char* where, const char* what);
#define AVSR_CHECK_LAUNCH(name)                                                   \
    do {                                                                          \
        int e__ = avsr_take_launch_error();                                       \
        if (e__ != 0) {                                                           \
            avsr_set_error2(name, hipGetErrorString((hipError_t)e__));            \
            return 2;                                                             \
        }                                                                         \
    } while (0)
#define AVSR_REQUIRE(cond, msg)         \
    do {                                \
        if (!(cond)) {                  \
            avsr_set_error(msg);        \
            return 1;                   \
        }                               \
    } while (0)
