// hip_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny host-side SIMT emulator used to debug the kernels under
// auto_avsr_amd/csrc on a machine without a GPU.  The very same kernel sources
// are compiled with `clang++ -x c++ -DAVSR_EMU` against this header; every GPU
// thread becomes a fiber, a workgroup is a set of fibers run round-robin on one
// OS thread, and wavefront collectives (shuffles, MFMA) are rendezvous points
// between the 64 fibers of a wave.  Nothing in the product imports or links
// this: auto_avsr_amd/_lib.py only ever loads libavsr_hip.so (gfx950 code
// object) and raises if it is absent.  The emulator exists so that index /
// layout bugs are found in seconds here rather than in GPU-minutes.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {
void launch(dim3 grid, dim3 block, size_t dyn_smem, const std::function<void()>& body);
void sync_threads();
// Wave rendezvous: deposit `bytes` for this lane; returns pointer to the wave's
// staging area laid out as [64][stride] once every live lane has deposited.
const unsigned char* wave_gather(const void* mine, size_t bytes, size_t* stride);
char* dyn_smem();
int lane_id();
}  // namespace emu

inline void __syncthreads() { emu::sync_threads(); }

template <class T>
inline T emu_shfl_src(T v, int src_lane) {
    size_t stride;
    int me = emu::lane_id();
    const unsigned char* all = emu::wave_gather(&v, sizeof(T), &stride);
    T out;
    int s = src_lane;
    if (s < 0 || s > 63) s = me;
    std::memcpy(&out, all + (size_t)s * stride, sizeof(T));
    return out;
}
template <class T> inline T __shfl_xor(T v, int mask, int = 64) { return emu_shfl_src(v, emu::lane_id() ^ mask); }
template <class T> inline T __shfl_down(T v, int d, int = 64) { return emu_shfl_src(v, emu::lane_id() + d); }
template <class T> inline T __shfl_up(T v, int d, int = 64) { return emu_shfl_src(v, emu::lane_id() - d); }
template <class T> inline T __shfl(T v, int src, int = 64) { return emu_shfl_src(v, src & 63); }

inline float atomicAdd(float* p, float v) {
    uint32_t* ip = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED);
    for (;;) {
        float f;
        std::memcpy(&f, &old, 4);
        f += v;
        uint32_t nw;
        std::memcpy(&nw, &f, 4);
        if (__atomic_compare_exchange_n(ip, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
            float r;
            std::memcpy(&r, &old, 4);
            return r;
        }
    }
}
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
    std::memset(p, v, n);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsyncD2D(void* d, const void* s, size_t n, hipStream_t) {
    std::memcpy(d, s, n);
    return hipSuccess;
}

// HIP exposes integer/float min/max in device code; mirror them for the host build.
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline long min(long a, long b) { return a < b ? a : b; }
inline long max(long a, long b) { return a > b ? a : b; }
inline float min(float a, float b) { return a < b ? a : b; }
inline float max(float a, float b) { return a > b ? a : b; }
