// feed_probe.hip -- how fast can a 64x64-tile NT GEMM block be FED?  The access pattern of the tuned kernel's operand
// stream (300 blocks, each walking K in 64-wide tiles: 64 A rows + 64 B rows x 128 bytes per tile) with
//   mode 0: LDS-DMA (global_load_lds_dwordx4 into a 3-stage ring, counted vmcnt + barrier per tile)  -- what the kernel does
//   mode 1: plain global_load_dwordx4 into registers, U tiles in flight, no LDS
//   mode 2: plain loads + ds_write_b128 into a ring + barrier per tile (register-staged pipeline)
// No MFMA / LDS reads: this is the ceiling the feed path alone sets.   hipcc --offload-arch=gfx950 -O3 -o feed_probe feed_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ inline void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N> __device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int U>
__global__ __launch_bounds__(256) void feed(const bf16_t* A, const bf16_t* B, int M, int N, int K, unsigned* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-contiguous tile order as in the kernel
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int gx = gridDim.x, total = gx * gridDim.y, id = by * gx + bx;
        const int xcd = id & 7, slot = id >> 3, q = total >> 3, r = total & 7;
        const int nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        by = nid / gx; bx = nid - by * gx;
    }
    const int m0 = by * 64, n0 = bx * 64, nt = K / 64;
    const int rsub = lane >> 3, pc = lane & 7;
    const bf16_t* pa[2]; const bf16_t* pb[2];
    for (int i = 0; i < 2; i++) {
        const int r = (wave * 2 + i) * 8 + rsub;
        pa[i] = A + (size_t)min(m0 + r, M - 1) * K + pc * 8;
        pb[i] = B + (size_t)min(n0 + r, N - 1) * K + pc * 8;
    }
    unsigned acc = 0;
    if (MODE == 0) {
        constexpr int ST = 3, SB = 16384;
        auto issue = [&](int t) {
            char* st = smem + (t % ST) * SB;
            for (int i = 0; i < 2; i++) glds16(pa[i] + t * 64, st + (wave * 2 + i) * 1024);
            for (int i = 0; i < 2; i++) glds16(pb[i] + t * 64, st + 8192 + (wave * 2 + i) * 1024);
        };
        for (int s = 0; s < ST - 1; s++) if (s < nt) issue(s);
        for (int t = 0; t < nt; t++) {
            if (t + ST - 1 < nt) wait_vmcnt<(ST - 2) * 4>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (t + ST - 1 < nt) issue(t + ST - 1);
            acc += *reinterpret_cast<unsigned*>(smem + (t % ST) * SB + threadIdx.x * 4);
        }
    } else if (MODE == 1) {
        u32x4 v[U][4];
        auto ld = [&](int t, int s) {
            for (int i = 0; i < 2; i++) v[s][i] = *reinterpret_cast<const u32x4*>(pa[i] + t * 64);
            for (int i = 0; i < 2; i++) v[s][2 + i] = *reinterpret_cast<const u32x4*>(pb[i] + t * 64);
        };
#pragma unroll
        for (int s = 0; s < U; s++) if (s < nt) ld(s, s);
        for (int t0 = 0; t0 < nt; t0 += U) {
#pragma unroll
            for (int s = 0; s < U; s++) {
                if (t0 + s >= nt) break;
                for (int i = 0; i < 4; i++) acc += v[s][i][0] ^ v[s][i][3];
                if (t0 + s + U < nt) ld(t0 + s + U, s);
            }
        }
    } else {
        constexpr int SB = 16384;  // ring of 2 LDS stages, U register sets in flight
        u32x4 v[U][4];
        auto ld = [&](int t, int s) {
            for (int i = 0; i < 2; i++) v[s][i] = *reinterpret_cast<const u32x4*>(pa[i] + t * 64);
            for (int i = 0; i < 2; i++) v[s][2 + i] = *reinterpret_cast<const u32x4*>(pb[i] + t * 64);
        };
#pragma unroll
        for (int s = 0; s < U; s++) if (s < nt) ld(s, s);
        for (int t0 = 0; t0 < nt; t0 += U) {
#pragma unroll
            for (int s = 0; s < U; s++) {
                const int t = t0 + s;
                if (t >= nt) break;
                char* st = smem + (t & 1) * SB;
                for (int i = 0; i < 2; i++) {
                    const int r = (wave * 2 + i) * 8 + rsub;
                    *reinterpret_cast<u32x4*>(st + r * 128 + ((pc ^ ((r >> 1) & 7)) << 4)) = v[s][i];
                    *reinterpret_cast<u32x4*>(st + 8192 + r * 128 + ((pc ^ ((r >> 1) & 7)) << 4)) = v[s][2 + i];
                }
                if (t + U < nt) ld(t + U, s);
                __syncthreads();
                acc += *reinterpret_cast<unsigned*>(st + threadIdx.x * 4);
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); exit(1); } } while (0)

template <int MODE, int U>
float run(const bf16_t* A, const bf16_t* B, int M, int N, int K, unsigned* sink, size_t lds) {
    dim3 grid((N + 63) / 64, (M + 63) / 64);
    hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    for (int i = 0; i < 10; i++) feed<MODE, U><<<grid, 256, lds>>>(A, B, M, N, K, sink);
    CK(hipEventRecord(s));
    for (int i = 0; i < 100; i++) feed<MODE, U><<<grid, 256, lds>>>(A, B, M, N, K, sink);
    CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
    float ms; CK(hipEventElapsedTime(&ms, s, e));
    return ms * 10.f;  // us per launch
}

int main() {
    const int shapes[][3] = {{1600, 768, 768}, {1600, 768, 3072}, {1600, 3072, 768}, {1600, 2304, 768}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        bf16_t *A, *B; unsigned* sink;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&sink, 64));
        CK(hipMemset(A, 1, (size_t)M * K * 2)); CK(hipMemset(B, 1, (size_t)N * K * 2));
        const double mb = ((double)((M + 63) / 64) * ((N + 63) / 64)) * (K / 64) * 16384 / 1e6;
        printf("%dx%dx%d  (%.0f MB through the feed path)\n", M, N, K, mb);
        printf("  lds-dma 3-stage      %7.2f us\n", run<0, 1>(A, B, M, N, K, sink, 49152));
        printf("  regs U=2             %7.2f us\n", run<1, 2>(A, B, M, N, K, sink, 0));
        printf("  regs U=4             %7.2f us\n", run<1, 4>(A, B, M, N, K, sink, 0));
        printf("  regs U=6             %7.2f us\n", run<1, 6>(A, B, M, N, K, sink, 0));
        printf("  regs+ds_write U=2    %7.2f us\n", run<2, 2>(A, B, M, N, K, sink, 32768));
        printf("  regs+ds_write U=3    %7.2f us\n", run<2, 3>(A, B, M, N, K, sink, 32768));
        printf("  regs+ds_write U=4    %7.2f us\n", run<2, 4>(A, B, M, N, K, sink, 32768));
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(sink));
    }
    return 0;
}
