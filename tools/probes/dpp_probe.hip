#include <hip/hip_runtime.h>
__global__ void k(float* o) {
    float x = (float)threadIdx.x;
    int up = __builtin_amdgcn_update_dpp(__float_as_int(-1.f), __float_as_int(x), 0x138, 0xf, 0xf, false);
    int dn = __builtin_amdgcn_update_dpp(__float_as_int(-2.f), __float_as_int(x), 0x130, 0xf, 0xf, false);
    o[threadIdx.x] = __int_as_float(up);
    o[64 + threadIdx.x] = __int_as_float(dn);
}
int main() {
    float* d; hipMalloc(&d, 128 * 4);
    k<<<1, 64>>>(d);
    float h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("up: %g %g %g ... %g %g\n", h[0], h[1], h[2], h[62], h[63]);
    printf("dn: %g %g %g ... %g %g\n", h[64], h[65], h[66], h[126], h[127]);
    printf("row edges up: %g %g %g  dn: %g %g %g\n", h[15], h[16], h[17], h[64+15], h[64+16], h[64+31]);
    return 0;
}
