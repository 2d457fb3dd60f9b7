// Probe of ds_read_b64_tr_b16 lane semantics on gfx950 (development aid for the backward-weights kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(float* out) {
    __shared__ __attribute__((aligned(16))) __fp16 t[16 * 64];
    for (int i = threadIdx.x; i < 16 * 64; i += 64) t[i] = (__fp16)(float)i;  // t[r][c] = r*64 + c  (exact in fp16 up to 2048)
    __syncthreads();
    const int lane = threadIdx.x;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    __fp16* addr = &t[q * 64 + g * 16 + 4 * p];
    fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)addr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (float)v[e];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", (int)h[l * 4 + e] / 64, (int)h[l * 4 + e] % 64);
        printf("\n");
    }
    return 0;
}
