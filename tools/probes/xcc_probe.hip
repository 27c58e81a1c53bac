// Probe: which XCD does workgroup (bx, by) of a 2-D grid land on?  Reads HW_REG_XCC_ID (hwreg 20 on gfx942 / gfx950).
// Prints, for several grid shapes, how often xcc == (by * gx + bx) % 8 and the first 32 ids.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void probe(int* out, int spin) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (int)(v & 0xf);
    // keep the block alive for a while so that the whole grid is resident at once (like a GEMM block)
    long t0 = clock64();
    while (clock64() - t0 < spin) { }
}

int main() {
    int* d;
    CHECK(hipMalloc(&d, 1 << 20));
    int shapes[][3] = {{12, 25, 1}, {300, 1, 1}, {36, 25, 1}, {48, 13, 1}, {12, 25, 2}, {6, 13, 1}};
    for (auto& s : shapes) {
        const int n = s[0] * s[1] * s[2];
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipMemset(d, 0xff, n * 4));
            hipLaunchKernelGGL(probe, dim3(s[0], s[1], s[2]), dim3(256), 48 * 1024, 0, d, 20000);
            CHECK(hipDeviceSynchronize());
        }
        int* h = (int*)malloc(n * 4);
        CHECK(hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost));
        int ok = 0, cnt[16] = {0};
        for (int i = 0; i < n; i++) { ok += h[i] == i % 8; cnt[h[i] & 15]++; }
        printf("grid (%d,%d,%d): %d/%d blocks on xcc == id%%8; per-xcc counts:", s[0], s[1], s[2], ok, n);
        for (int x = 0; x < 8; x++) printf(" %d", cnt[x]);
        printf("\n  first ids:");
        for (int i = 0; i < 40 && i < n; i++) printf(" %d", h[i]);
        printf("\n");
        free(h);
    }
    return 0;
}
