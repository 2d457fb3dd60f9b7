// Probe: under the chip's power limit, does the 16x16x32 fp16 MFMA shape deliver more FLOP/s than 32x32x16 for the SAME work per
// wave (64x64 output tile, all operands re-read from LDS by ds_read_b128, random data)?  (MI355X_MICROARCH.md, DVFS give-back (7))
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const _Float16* __restrict__ src, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32768 / 16; i += 256) reinterpret_cast<half8*>(lds)[i] = reinterpret_cast<const half8*>(src)[i + blockIdx.x % 7];
    __syncthreads();
    const char* base = lds + (tid >> 6) * 4096;
    if (SHAPE == 32) {
        f32x16 acc[2][2] = {};
        for (int it = 0; it < iters; ++it) {
            const int o = (it & 7) * 1024;
            half8 a[2], b[2], al[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const half8*>(base + ((o + i * 512 + lane * 16) & 4095) + 0);
                al[i] = *reinterpret_cast<const half8*>(base + ((o + 256 + i * 512 + lane * 16) & 4095) + 8192);
                b[i] = *reinterpret_cast<const half8*>(base + ((o + i * 512 + lane * 16) & 4095) + 16384);
                bl[i] = *reinterpret_cast<const half8*>(base + ((o + 256 + i * 512 + lane * 16) & 4095) + 24576 - 4096 * (tid >> 6) + 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], b[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
        }
        float s = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        out[blockIdx.x * 256 + tid] = s;
    } else {
        // same 64x64x16 work per iteration: 4x4 tiles of 16x16, K = 32 per MFMA -> per 16 k: 8 MFMAs of 16x16x32 per product term
        f32x4 acc[4][4] = {};
        for (int it = 0; it < iters; it += 2) {  // one pass covers K = 32 = two of the other variant's iterations
            const int o = (it & 7) * 1024;
            half8 a[4], b[4], al[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const half8*>(base + ((o + i * 256 + lane * 16) & 4095) + 0);
                al[i] = *reinterpret_cast<const half8*>(base + ((o + 128 + i * 256 + lane * 16) & 4095) + 8192);
                b[i] = *reinterpret_cast<const half8*>(base + ((o + i * 256 + lane * 16) & 4095) + 16384);
                bl[i] = *reinterpret_cast<const half8*>(base + ((o + 128 + i * 256 + lane * 16) & 4095) + 24576 - 4096 * (tid >> 6) + 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], b[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
        }
        float s = 0;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
        out[blockIdx.x * 256 + tid] = s;
    }
}

int main() {
    const int blocks = 2048, iters = 4096;
    std::vector<_Float16> h(32768 + 64);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    _Float16* src; float* out;
    (void)hipMalloc(&src, h.size() * 2); (void)hipMalloc(&out, blocks * 256 * 4);
    (void)hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {32, 16}) {
            for (int w = 0; w < 2; ++w) {
                if (w == 1) (void)hipEventRecord(e0);
                for (int r = 0; r < (w ? 8 : 2); ++r) {
                    if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
                    else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
                }
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            // MFMA flops per wave per iteration: 12 MFMAs x 32*32*16*2
            const double fl = 8.0 * blocks * 4.0 * iters * 12.0 * 32 * 32 * 16 * 2;
            printf("shape %2d: %.3f ms  %.1f TFLOP/s issued\n", shape, ms, fl / (ms * 1e-3) / 1e12);
        }
    return 0;
}
