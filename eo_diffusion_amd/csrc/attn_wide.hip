// Fused attention forward for WIDE heads (64 < d <= 512), natural qkv layout:
//     out[n, t, h*d + j] = sum_s softmax_s(q_t . k_s / sqrt(d)) v[s, j]          (QKVAttention(Legacy), unet_openai.py:465-515)
// The train.py:50 architecture runs ONE head over all 512 channels of its middle block (unet_openai.py:675-681: num_heads = 1,
// num_head_channels = -1); the <= 64-channel kernels (attn_bwd.hip / attn_x3.hip) keep a query's whole head in a lane's registers and
// cannot hold 512.  Here the HEAD DIM is split over the four waves of a workgroup instead of the queries:
//   * a workgroup owns 32 queries (lane & 31 = query) of one (image, head) and walks 32-key tiles;
//   * wave w owns the channel slice [w SLW, (w + 1) SLW), SLW = 32 NSL >= d / 4, in BOTH products:
//       S^T partial = K[:, slice] Q[:, slice]^T   (32 x 32 MFMA; K fragments straight from global memory: nobody else reads them)
//       -> the four partials meet in LDS ([4][16 regs][64 lanes] fp32, added in a fixed order by every wave: all four hold the same
//          bits of S afterwards), online softmax in fp32, redundantly per wave (32 exponentials per lane and tile)
//       O^T[slice] += V[:, slice]^T P^T           (P^T from the accumulator registers, V^T by transposed reads of the wave's PRIVATE
//          row-major V tile in LDS: no barrier around it)
//   * one barrier per key tile (the exchange buffer is double buffered).
// T x T never exists; the five launches of the materialised path (transposed v projection, max|x| pass, score GEMM, row softmax,
// P.V GEMM) become one.  fp32 storage (T = float): both products as three fp16 MFMAs on operands split into hi + lo and scaled by the
// image's power of two from the bound table of qkv (see attn_x3.hip); fp16 storage (T = half_t): plain fp16 MFMA, fp32 softmax.
// Layout: qkv [N][T][3C] as produced by the qkv projection (channel = q_off / k_off / v_off + head*head_stride + j), out [N][T][C],
// optional lse [N][heads][T].  Any T; d % 8 == 0, 64 < d <= 512.
#include "common.h"
#include <type_traits>

typedef __fp16 fp16x4w __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16x4w* aw_lds_h4;

constexpr int AW_ROWB = 128;            // V sub-tile row = 64 halves
constexpr int AW_VT = 32 * AW_ROWB;     // one sub-tile: 32 keys x 64 channels (fp16) = 4 KiB
constexpr int AW_KMIN = EOD_AB_KMIN_ATTN;

__device__ __forceinline__ int aw_swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int aw_off(int row, int c) { return row * AW_ROWB + ((c ^ aw_swz(row)) << 4); }

// fragment for a contraction over 16 tile ROWS in the order the accumulator registers 8*kb .. 8*kb+7 of a 32 x 32 C tile hold them:
// rows rb + 4 kg + {0..3} and rb + 8 + 4 kg + {0..3} (kg = lane >> 5), columns j0 + (lane & 31)   (see attn_bwd.hip)
__device__ __forceinline__ half8 aw_tr_frag(const char* tile, int rb, int j0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int kg = g >> 1;
    const int col = j0 + (g & 1) * 16 + 4 * p;
    const int r0 = rb + 4 * kg + q, r1 = r0 + 8;
    const fp16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((aw_lds_h4)(tile + aw_off(r0, col >> 3) + ((col >> 2) & 1) * 8));
    const fp16x4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((aw_lds_h4)(tile + aw_off(r1, col >> 3) + ((col >> 2) & 1) * 8));
    half8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[e] = (half_t)lo[e];
        f[4 + e] = (half_t)hi[e];
    }
    return f;
}

// (s x0, s x1) -> packed {hi0, hi1}, {lo0, lo1}: hi = fp16(s x), lo = fp16(s x - hi), four v_fma_mix instructions (see split4_scaled in igemm.hip)
__device__ __forceinline__ void aw_split2(float x0, float x1, float s, int& h, int& l) {
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l)
        : "v"(x0), "v"(x1), "v"(s));
}
// eight fp32 values (two 16-byte loads) -> the hi and lo fp16 fragments of s x
__device__ __forceinline__ void aw_split8(const f32x4& a, const f32x4& c, float s, half8& hi, half8& lo) {
    int h[4], l[4];
    aw_split2(a[0], a[1], s, h[0], l[0]);
    aw_split2(a[2], a[3], s, h[1], l[1]);
    aw_split2(c[0], c[1], s, h[2], l[2]);
    aw_split2(c[2], c[3], s, h[3], l[3]);
    hi = __builtin_bit_cast(half8, i32x4{h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(half8, i32x4{l[0], l[1], l[2], l[3]});
}

struct AttnWideP {
    const char* qkv;
    char* out;
    float* lse;
    int N, T, C, heads, d, q_off, k_off, v_off, hs;
    float scale_log2;  // log2(e) / sqrt(d)
    const float* ab;   // fp32 storage: bound table [N][EOD_AB] of qkv, or NULL (|x| < 4094 guaranteed by the caller)
};

template <typename T, int NSL>
__global__ __launch_bounds__(256, 1) void attn_fwd_wide_kernel(const AttnWideP p) {
    constexpr bool SPLIT = sizeof(T) == 4;
    constexpr int ES = sizeof(T);
    constexpr int SLW = 32 * NSL, KS = SLW / 16;  // channels per wave, 16-channel steps of the score product
    constexpr int NT64 = (NSL + 1) / 2;           // 64-channel sub-tiles of the wave's V slice
    constexpr int VWAVE = NT64 * AW_VT * (SPLIT ? 2 : 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    char* const sVh = smem + wave * VWAVE;
    char* const sVl = sVh + NT64 * AW_VT;  // (SPLIT only)
    f32x4* const sX = reinterpret_cast<f32x4*>(smem + 4 * VWAVE);  // [2 buffers][4 waves][4 register quads][64 lanes]
    const int b = blockIdx.y, n = b / p.heads, h = b - n * p.heads;
    const int q0 = blockIdx.x * 32;
    const int c0 = wave * SLW;  // this wave's first channel inside the head
    const long long ld = 3LL * p.C;
    const T* base = reinterpret_cast<const T*>(p.qkv) + (long long)n * p.T * ld + h * p.hs;

    // operand scale of this image (fp32 storage), 1 / s, and the exponent's factor log2(e) / sqrt(d) / s^2
    float s_op = 1.0f, rscale = 1.0f;
    if constexpr (SPLIT) {
        AbScale asc = {16.0f, 1.0f};
        if (p.ab) asc = ab_scale_of(ab_wave_bound(p.ab, n), AW_KMIN);
        s_op = asc.s;
        rscale = asc.inv * 0.0625f;
    }
    const float scale_log2 = p.scale_log2 * rscale * rscale;

    // ---- this lane's query row, channels c0 + 16 ks + 8 lh .. + 7 ----
    half8 qh[KS], ql[SPLIT ? KS : 1];
    {
        const int q = q0 + lr;
        const T* qp = base + (long long)q * ld + p.q_off + c0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int j = 16 * ks + 8 * lh;
            const bool ok = q < p.T && c0 + j < p.d;
            if constexpr (SPLIT) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    a = *reinterpret_cast<const f32x4*>(qp + j);
                    c = *reinterpret_cast<const f32x4*>(qp + j + 4);
                }
                aw_split8(a, c, s_op, qh[ks], ql[ks]);
            } else {
                i32x4 a = {0, 0, 0, 0};
                if (ok) a = *reinterpret_cast<const i32x4*>(qp + j);
                qh[ks] = __builtin_bit_cast(half8, a);
            }
        }
    }

    // ---- K fragments of a tile: key = 32 kt + lr, the same channel groups as Q; straight into registers ----
    constexpr int KR = SPLIT ? 2 : 1;
    i32x4 kreg[KS][KR];
    auto load_k = [&](int kt) {
        const int key = kt * 32 + lr;
        const T* kp = base + (long long)key * ld + p.k_off + c0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int j = 16 * ks + 8 * lh;
            const bool ok = key < p.T && c0 + j < p.d;
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                kreg[ks][r] = i32x4{0, 0, 0, 0};
                if (ok) kreg[ks][r] = *reinterpret_cast<const i32x4*>(kp + j + r * 4);
            }
        }
    };
    // ---- V slice of a tile: 32 keys x SLW channels, row-major; slot s = lane + 64 i -> (row, 16-byte column) ----
    constexpr int VCOLS = SLW * ES / 16;        // 16-byte columns per row of the slice
    constexpr int VL = 32 * VCOLS / 64;         // loads per lane
    i32x4 vreg[VL];
    auto load_v = [&](int kt) {
#pragma unroll
        for (int i = 0; i < VL; ++i) {
            const int s = lane + 64 * i, row = s / VCOLS, col = s - row * VCOLS;
            const int key = kt * 32 + row;
            vreg[i] = i32x4{0, 0, 0, 0};
            if (key < p.T && c0 + col * (16 / ES) < p.d)
                vreg[i] = *reinterpret_cast<const i32x4*>(base + (long long)key * ld + p.v_off + c0 + col * (16 / ES));
        }
    };
    auto store_v = [&]() {
#pragma unroll
        for (int i = 0; i < VL; ++i) {
            const int s = lane + 64 * i, row = s / VCOLS, col = s - row * VCOLS;
            if constexpr (SPLIT) {  // col = float4 column: sub-tile col / 16, inside it 16-byte chunk (col % 16) / 2, half (col & 1)
                const int off = (col >> 4) * AW_VT + aw_off(row, (col & 15) >> 1) + (col & 1) * 8;
                const f32x4 v = __builtin_bit_cast(f32x4, vreg[i]);
                int h0, l0, h1, l1;
                aw_split2(v[0], v[1], s_op, h0, l0);
                aw_split2(v[2], v[3], s_op, h1, l1);
                typedef int i32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<i32x2*>(sVh + off) = i32x2{h0, h1};
                *reinterpret_cast<i32x2*>(sVl + off) = i32x2{l0, l1};
            } else {                // col = 8-channel chunk: sub-tile col / 8, chunk col % 8
                *reinterpret_cast<i32x4*>(sVh + (col >> 3) * AW_VT + aw_off(row, col & 7)) = vreg[i];
            }
        }
    };

    f32x16 o[NSL];  // O^T tiles: registers = output channel, lane & 31 = query
#pragma unroll
    for (int t = 0; t < NSL; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const int nkt = (p.T + 31) / 32;
    load_k(0);
    for (int kt = 0; kt < nkt; ++kt) {
        load_v(kt);  // in flight under the score product and the exchange
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (SPLIT) {
                half8 kh, kl;
                aw_split8(__builtin_bit_cast(f32x4, kreg[ks][0]), __builtin_bit_cast(f32x4, kreg[ks][1]), s_op, kh, kl);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
            } else {
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, kreg[ks][0]), qh[ks], s, 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) load_k(kt + 1);  // in flight under the softmax and the second product
        // ---- the four channel slices' partial scores meet in LDS; every wave adds them in the same order ----
        {
            f32x4* xw = sX + ((kt & 1) * 16 + wave * 4) * 64 + lane;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) xw[r4 * 64] = f32x4{s[4 * r4], s[4 * r4 + 1], s[4 * r4 + 2], s[4 * r4 + 3]};
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): my partial is in LDS
            __builtin_amdgcn_s_barrier();        // every partial of this tile is in LDS.  (Two buffers: a wave that writes tile kt + 1's partial
                                                 //  has passed THIS barrier, i.e. every wave has finished reading tile kt - 1's buffer, the one it reuses)
            const f32x4* xr = sX + (kt & 1) * 16 * 64 + lane;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 a = xr[(0 * 4 + r4) * 64];
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const f32x4 c = xr[(w * 4 + r4) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] += c[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) s[4 * r4 + e] = a[e];
            }
        }
        // ---- online softmax over this tile's 32 keys (16 in this lane's registers, 16 in lane ^ 32), fp32 ----
        const bool ragged = kt * 32 + 32 > p.T;
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (ragged) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                s[r] = key < p.T ? s[r] : -INFINITY;
            }
            mloc = fmaxf(mloc, s[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc * scale_log2);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float lsum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2, -m_new));
            s[r] = e;
            lsum += e;
        }
        l_run = l_run * alpha + lsum;
#pragma unroll
        for (int t = 0; t < NSL; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        // ---- O^T[slice] += V[:, slice]^T P^T ----
        store_v();  // (this wave's private tile: LDS operations of one wave execute in order, no barrier)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            half8 ph, pl;
            if constexpr (SPLIT) {
                int phi[4], pli[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) aw_split2(s[8 * kb + 2 * e], s[8 * kb + 2 * e + 1], 1.0f, phi[e], pli[e]);
                ph = __builtin_bit_cast(half8, i32x4{phi[0], phi[1], phi[2], phi[3]});
                pl = __builtin_bit_cast(half8, i32x4{pli[0], pli[1], pli[2], pli[3]});
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) ph[e] = (half_t)s[8 * kb + e];
            }
#pragma unroll
            for (int t = 0; t < NSL; ++t) {
                const half8 vh = aw_tr_frag(sVh + (t >> 1) * AW_VT, 16 * kb, (t & 1) * 32, lane);
                if constexpr (SPLIT) {
                    const half8 vl = aw_tr_frag(sVl + (t >> 1) * AW_VT, 16 * kb, (t & 1) * 32, lane);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[t], 0, 0, 0);
                }
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[t], 0, 0, 0);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = rscale / l_tot;
    const int q = q0 + lr;
    if (q < p.T) {
        if (p.lse && lh == 0 && wave == 0) p.lse[((long long)n * p.heads + h) * p.T + q] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
        T* op = reinterpret_cast<T*>(p.out) + ((long long)n * p.T + q) * p.C + h * p.d + c0;
#pragma unroll
        for (int t = 0; t < NSL; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int j0 = t * 32 + 8 * g4 + 4 * lh;  // registers 4 g4 .. 4 g4 + 3 = 4 consecutive output channels
                if (c0 + j0 + 3 < p.d) {
                    if constexpr (SPLIT) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = o[t][4 * g4 + e] * inv;
                        *reinterpret_cast<f32x4*>(op + j0) = v;
                    } else {
                        half4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (half_t)(o[t][4 * g4 + e] * inv);
                        *reinterpret_cast<half4*>(op + j0) = v;
                    }
                }
            }
    }
}

template <typename T> static int launch_wide(const AttnWideP& p, hipStream_t st) {
    constexpr bool SPLIT = sizeof(T) == 4;
    const int nsl = ((p.d + 3) / 4 + 31) / 32;  // 32-channel blocks per wave: the smallest slice with 4 slices >= d
    const dim3 grid((p.T + 31) / 32, p.N * p.heads);
    const size_t lds = (size_t)4 * ((nsl + 1) / 2) * AW_VT * (SPLIT ? 2 : 1) + 2 * 16 * 64 * sizeof(f32x4);
#define EOD_AW(NSL_)                                                                                                                   \
    do {                                                                                                                           \
        static bool attr = false;                                                                                                  \
        if (!attr) {                                                                                                               \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_wide_kernel<T, NSL_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                                   \
            attr = true;                                                                                                           \
        }                                                                                                                          \
        hipLaunchKernelGGL((attn_fwd_wide_kernel<T, NSL_>), grid, dim3(256), lds, st, p);                                          \
    } while (0)
    if (nsl == 1) EOD_AW(1);
    else if (nsl == 2) EOD_AW(2);
    else if (nsl == 3) EOD_AW(3);
    else EOD_AW(4);
#undef EOD_AW
    EOD_CHECK_LAUNCH("attention_fwd_nat (wide heads)");
    return EOD_OK;
}

// called by eod_attention_fwd_nat (attn_bwd.hip) for 64 < d <= 512
int eod_attention_fwd_wide(const void* qkv, void* out, float* lse, int dtype, int N, int T, int C, int heads, int d, int q_off, int k_off,
                           int v_off, int head_stride, const float* qkv_bound, hipStream_t st) {
    const int es = eod_esize(dtype), epc = 16 / es;
    EOD_REQUIRE(d % 8 == 0 && d > 64 && d <= 512, "attention_fwd_nat (wide heads): head dim %d", d);
    EOD_REQUIRE(q_off % epc == 0 && k_off % epc == 0 && v_off % epc == 0 && head_stride % epc == 0 && C % epc == 0 && eod_aligned16(qkv) &&
                    eod_aligned16(out),
                "attention_fwd_nat (wide heads): alignment of the head slices");
    EOD_REQUIRE(N * heads <= 65535, "attention_fwd_nat (wide heads): %d x %d heads", N, heads);
    AttnWideP p;
    p.qkv = (const char*)qkv; p.out = (char*)out; p.lse = lse;
    p.N = N; p.T = T; p.C = C; p.heads = heads; p.d = d; p.q_off = q_off; p.k_off = k_off; p.v_off = v_off; p.hs = head_stride;
    p.scale_log2 = 1.4426950408889634f / sqrtf((float)d);
    p.ab = qkv_bound;
    return dtype == EOD_F16 ? launch_wide<half_t>(p, st) : launch_wide<float>(p, st);
}
