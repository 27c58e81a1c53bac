// Probe: global -> LDS fill rate per CU on gfx950 for (0) LDS-DMA global_load_lds_dwordx4, (1) global_load_dwordx4 ->
// VGPR -> ds_write_b128, (2) global_load_dwordx4 -> VGPR only; from L2-resident / MALL-resident / HBM-streamed data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, bool BAR>
__global__ __launch_bounds__(256) void fill(const uint4* __restrict__ src, uint32_t ntiles, int iters, uint32_t* sink) {
    __shared__ __attribute__((aligned(16))) uint4 lds[2][1024];  // 2 x 16 KB
    const int tid = threadIdx.x, wave = tid >> 6;
    uint32_t acc = 0;
    uint32_t t = (blockIdx.x * 131u) % ntiles;
    if (MODE == 0) {
        for (int it = 0; it < iters; it++) {
            const uint4* g = src + (size_t)t * 1024;
            uint4* l = lds[it & 1];
#pragma unroll
            for (int j = 0; j < 4; j++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + j * 256 + tid),
                                                 (__attribute__((address_space(3))) void*)(l + j * 256 + wave * 64), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // previous iteration's tile has landed
            if (BAR) __builtin_amdgcn_s_barrier();
            t += 17; if (t >= ntiles) t -= ntiles;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc = lds[0][tid].x ^ lds[1][tid].y;
    } else {
        uint4 r[4];
        {
            const uint4* g = src + (size_t)t * 1024;
#pragma unroll
            for (int j = 0; j < 4; j++) r[j] = g[j * 256 + tid];
        }
        for (int it = 0; it < iters; it++) {
            t += 17; if (t >= ntiles) t -= ntiles;
            const uint4* g = src + (size_t)t * 1024;
            uint4 n[4];
#pragma unroll
            for (int j = 0; j < 4; j++) n[j] = g[j * 256 + tid];
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) lds[it & 1][j * 256 + tid] = r[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) acc ^= r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
            }
            if (BAR) __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int j = 0; j < 4; j++) r[j] = n[j];
        }
        if (MODE == 1) { __syncthreads(); acc = lds[0][tid].x ^ lds[1][tid].y; }
#pragma unroll
        for (int j = 0; j < 4; j++) acc ^= r[j].x;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, bool BAR>
static void run(const uint4* src, size_t bytes, int bpc, uint32_t* sink, const char* label) {
    const uint32_t ntiles = (uint32_t)(bytes / 16384);
    const int grid = 256 * bpc;
    int iters = 2000;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    fill<MODE, BAR><<<grid, 256>>>(src, ntiles, 200, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    fill<MODE, BAR><<<grid, 256>>>(src, ntiles, iters, sink);
    CHECK(hipEventRecord(b)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double tot = (double)grid * iters * 16384.0;
    printf("%-6s mode=%d bar=%d blocks/CU=%d: %8.1f GB/s chip, %6.1f GB/s/CU, %5.1f B/clk/CU(@2.4GHz)\n", label, MODE, (int)BAR,
           bpc, tot / ms / 1e6, tot / ms / 1e6 / 256, tot / ms / 1e6 / 256 / 2.4);
}

int main() {
    const size_t big = (size_t)2 << 30;
    uint4* src; uint32_t* sink;
    CHECK(hipMalloc(&src, big)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(src, 1, big));
    struct { size_t bytes; const char* label; } sets[] = {{(size_t)2 << 20, "L2"}, {(size_t)96 << 20, "MALL"}, {big, "HBM"}};
    for (auto& s : sets)
        for (int bpc = 1; bpc <= 4; bpc++) {
            run<0, true>(src, s.bytes, bpc, sink, s.label);
            run<0, false>(src, s.bytes, bpc, sink, s.label);
            run<1, true>(src, s.bytes, bpc, sink, s.label);
            run<1, false>(src, s.bytes, bpc, sink, s.label);
            run<2, false>(src, s.bytes, bpc, sink, s.label);
        }
    return 0;
}
