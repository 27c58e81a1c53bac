// Probe the semantics of ds_read_b64_tr_b16 on gfx950: lane i supplies an address of 4 consecutive b16 values.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (uint16_t)i;  // element value = its index
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    // group g reads rows 4g..4g+3 (row pitch 64 elements), columns 0..15: lane i -> row 4g + i/4, cols (i%4)*4..+3
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;
    const uint32_t addr = base + (uint32_t)(((4 * g + (i >> 2)) * 64 + (i & 3) * 4) * 2);
    typedef short s4 __attribute__((ext_vector_type(4)));
    s4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; j++) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    k<<<1, 64>>>(d);
    printf("launch: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) { printf("lane %2d:", l); for (int j = 0; j < 4; j++) printf(" (r%d,c%d)", h[l*4+j] / 64, h[l*4+j] % 64); printf("\n"); }
    return 0;
}
