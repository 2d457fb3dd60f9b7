#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const int* src, int nbytes, int* out) {
    __shared__ __attribute__((aligned(16))) int sm[256];
    for (int i = threadIdx.x; i < 256; i += 64) sm[i] = 0x7777;   // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x % 3 == 0) voff = 0xFFFFFF00u;   // out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)sm, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = sm[i];
}
int main() {
    int h[256]; for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
    int *d, *o; hipMalloc(&d, 1024); hipMalloc(&o, 1024);
    hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1024, o);
    int r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 8; ++l) printf("lane %d: %d %d %d %d\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3]);
    return 0;
}
