// Shared host/device helpers for libeodiff (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/eodiff.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

void eod_set_error(const char* fmt, ...);

#define EOD_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            eod_set_error(__VA_ARGS__);   \
            return EOD_EINVAL;            \
        }                                 \
    } while (0)

#define EOD_CHECK_LAUNCH(what)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            eod_set_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
            return EOD_ELAUNCH;                                                  \
        }                                                                        \
    } while (0)

static inline bool eod_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int eod_esize(int dtype) { return dtype == EOD_F16 ? 2 : 4; }

template <typename T> struct dt;
template <> struct dt<float> {
    static constexpr int id = EOD_F32;
    static constexpr int epc = 4;  // elements per 16-byte chunk
};
template <> struct dt<half_t> {
    static constexpr int id = EOD_F16;
    static constexpr int epc = 8;
};

// a * b rounded to fp32 as its own operation.  hipcc contracts a multi-use fmul into EVERY consuming add (aggressive FMA fusion), and
// HIP's __fmul_rn is a plain `*`: where a product has to exist as a rounded fp32 value before it is used (the sinusoid argument
// t*f in front of sin/cos range reduction: an unrounded product moves a 999 rad argument by up to half an ulp = 3e-5 in the
// result), it is pinned in a register behind an opaque asm.
__device__ __forceinline__ float mul_rn(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}

// ---------------------------------------------------------------------------------------------------------------------------
// ACTIVATION BOUND TABLES of the split-fp16 ("fp32x3") products.  An fp32 activation enters the fp16 matrix pipe as
// hi + lo = fp16(s x) + fp16(s x - hi); s must keep |s x| inside the fp16 range for EVERY element of the tensor, whatever its
// magnitude (the reference multiplies in IEEE fp32: unet_openai.py:352,262-264,227,609,414-422).  s is a power of two per IMAGE
// (per image, not per batch: a sample's bits must not depend on what else is in the batch), derived ON THE DEVICE from a bound
// table ab[N][EOD_AB]: EOD_AB non-negative floats per image whose maximum B_n bounds max|x| over the image (as the splitting
// consumer sees it, i.e. behind a fused GroupNorm + SiLU when there is one).  Producers: eod_gn_finalize (raw and normalised
// bounds of the tensors it takes statistics of, from the per-channel partial sums of squares: sqrt(sum over a slot's pixels of x^2)
// >= max|x| over those pixels) and eod_act_bound (the same from conv-epilogue statistics, or a direct |x| maximum of a tensor).
// Consumer rule: s = 2^k with B s in [2^14, 2^15)  ->  |s x| < 2^15 (no fp16 overflow).  B overestimates the true maximum by at most
// sqrt(pixels per slot) (conv epilogues: 64 pixels -> 8x; eod_gn_partial: HW / 256 pixels), so max|s x| >= 2^9 in the worst case and
// `lo` resolves 2^-25 absolute on the s x scale = 2^-34 of the image's largest element: a tensor of ANY magnitude gets the 2^-22
// relative product accuracy of the fp32x3 note in igemm.hip (elements 2^12 and more below the image maximum keep an absolute error
// of 2^-34 of that maximum -- invisible in any norm of the result).  Sums of squares overflow above sqrt(FLT_MAX) = 1.8e19: there B
// is inf, s falls back to 2^-113 (finite results, no accuracy claim) -- the magnitude at which the GroupNorm variance of the
// reference itself is inf; the direct |x| maximum of eod_act_bound has no such limit.
// k is clamped to [EOD_AB_KMIN, EOD_AB_KMAX] so that s, 16/s and their products with the weight scale stay normal fp32 numbers.
// ---------------------------------------------------------------------------------------------------------------------------
#define EOD_AB 32
#define EOD_AB_KMIN (-113)  // B = inf / NaN / 2^127: s = 2^-113 keeps every finite fp32 value inside the fp16 range
#define EOD_AB_KMAX 60      // images whose largest element is below 2^-46 keep s = 2^60 (graceful: |s x| < 2^14)
#define EOD_AB_KMIN_ATTN (-48)  // attention operands (q, k, v): s >= 2^-48 keeps 1 / s^2 (the softmax exponent's factor) a normal number;
                                // |q|, |k|, |v| up to 2^63 -- beyond that q . k overflows fp32 itself
struct AbScale {
    float s, inv;  // operand scale, and 16 / s (what the epilogue multiplies by on top of the weights' 1 / (16 s_w))
};
__device__ __forceinline__ AbScale ab_scale_of(float B, int kmin = EOD_AB_KMIN) {
    const unsigned eb = (__float_as_uint(B) >> 23) & 0xffu;  // biased exponent of the (non-negative) bound; 255 = inf / NaN
    int k = 141 - (int)eb;                                   // 14 - floor(log2 B)
    k = k < kmin ? kmin : (k > EOD_AB_KMAX ? EOD_AB_KMAX : k);
    AbScale r;
    r.s = __uint_as_float((unsigned)(k + 127) << 23);
    r.inv = __uint_as_float((unsigned)(4 - k + 127) << 23);
    return r;
}
// maximum of image n's EOD_AB table entries, wave-uniform (every lane of the wave must call); NaN entries count as +inf
__device__ __forceinline__ float ab_wave_bound(const float* __restrict__ ab, long long n) {
    float b = 0.0f;
    if ((threadIdx.x & 63) < EOD_AB) {
        b = ab[n * EOD_AB + (threadIdx.x & 63)];
        b = (b == b) ? b : __uint_as_float(0x7f800000u);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o));
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(b)));
}

// SiLU (nn.SiLU, unet_openai.py:314,330,338): precise form for the fp32 parity mode, fast form
// (v_exp_f32 + v_rcp_f32) for the fp16 mode where the result is rounded to 11 bits anyway.
template <bool FAST> __device__ __forceinline__ float silu_f(float v) {
    if constexpr (FAST) {
        return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    } else {
        return v / (1.0f + expf(-v));
    }
}

// Slab decomposition shared by the streaming GroupNorm passes (gn_apply in norm.hip, the backward apply in train.hip): block =
// (cpp = C / EPC chunk columns, ry pixel rows), thread (tx, ty) owns ONE 16-byte channel chunk for its whole slab, so the
// per-channel tables live in registers (no per-chunk table loads, no 64-bit index divisions) and consecutive threads still cover
// consecutive chunks of consecutive pixels.  U pixels per thread are in flight per trip.
struct GnSlab {
    int cpp, ry, per, P, nz;  // nz channel blocks of cpp chunk columns (blockIdx.z) cover layers wider than 256 chunks per pixel
};
static inline GnSlab gn_slab(int N, int HW, int C, int epc, int unroll) {
    GnSlab g;
    const int cols = C / epc;
    g.nz = (cols + 255) / 256;
    g.cpp = (cols + g.nz - 1) / g.nz;
    g.ry = 256 / g.cpp;
    if (g.ry < 1) g.ry = 1;
    const int quantum = g.ry * unroll;
    long long want = (4096 + N - 1) / N;  // ~16 workgroups per CU over the batch
    long long per = (HW + want - 1) / want;
    per = (per + quantum - 1) / quantum * quantum;
    g.per = (int)per;
    g.P = (int)((HW + per - 1) / per);
    return g;
}
