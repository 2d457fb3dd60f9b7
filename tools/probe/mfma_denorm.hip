// Probe: does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs on gfx950, and does the fp32->fp16 conversion
// produce them?  (Decides how the split-fp16 "hi + lo" operands of the fp32-tolerance conv mode may be scaled.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(const float* av, const float* bv, float* out, float* cv) {
    const int lane = threadIdx.x;
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (half_t)0.0f; b[j] = (half_t)0.0f; }
    // A[row r][k = 8h + j]: put av[r & 7] at k = 0 only; B[k=0][col] = bv[0]
    if (lane < 32) { a[0] = (half_t)av[lane & 7]; b[0] = (half_t)bv[0]; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    // C: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    if (lane == 0) { for (int r = 0; r < 4; ++r) out[r] = c[r]; }
    if (lane == 32) { for (int r = 0; r < 4; ++r) out[4 + r] = c[r]; }
    if (lane < 8) cv[lane] = (float)(half_t)av[lane];
}
int main() {
    float hav[8] = {1.0f, ldexpf(1.0f, -14), ldexpf(1.0f, -15), ldexpf(1.5f, -20), ldexpf(1.0f, -24), ldexpf(1.0f, -25), 3.0e-6f, 0.0f};
    float hbv[1] = {1024.0f};
    float *av, *bv, *out, *cv;
    hipMalloc(&av, 32); hipMalloc(&bv, 4); hipMalloc(&out, 32); hipMalloc(&cv, 32);
    hipMemcpy(av, hav, 32, hipMemcpyHostToDevice); hipMemcpy(bv, hbv, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, av, bv, out, cv);
    float ho[8], hc[8];
    hipMemcpy(ho, out, 32, hipMemcpyDeviceToHost); hipMemcpy(hc, cv, 32, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 8; ++r) {
        const double want = (double)(float)(half_t)hav[r] * 1024.0;
        printf("a=%.9g  cvt_f16=%.9g  mfma(a*1024)=%.9g  want=%.9g %s\n", hav[r], hc[r], ho[r], want, ho[r] == (float)want ? "ok" : "MISMATCH");
        bad += ho[r] != (float)want;
    }
    printf(bad ? "RESULT: fp16 subnormal inputs are NOT preserved by the MFMA (or the cvt)\n" : "RESULT: fp16 subnormals preserved\n");
    return 0;
}
