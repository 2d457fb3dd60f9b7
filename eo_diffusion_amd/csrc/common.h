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
