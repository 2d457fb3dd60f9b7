// Fused attention forward for fp32 STORAGE at fp32-grade accuracy on the fp16 matrix pipe ("fp32x3" mode):
//     out[n, t, h*d + j] = sum_s softmax_s(q_t . k_s / sqrt(d)) v[s, j]          (QKVAttention(Legacy), unet_openai.py:465-515)
// The reference materialises the T x T weights per head in fp32 (:476-480, 508-514); here they never leave the registers, and both
// contractions run as three fp16 MFMAs per product on operands split into hi + lo halves (see the fp32x3 note in igemm.hip):
//     S^T = K Q^T      : q, k (x s each) split when they are staged; S = Kl.Qh + Kh.Ql + Kh.Qh
//     O^T += V^T P^T   : P in [0, 1] split from the fp32 accumulator registers, v (x s) split when staged
// s = the power-of-two operand scale of the IMAGE, derived from the bound table of the qkv tensor (common.h: |s x| < 2^15 for every
// element, whatever the magnitude of q / k / v; no table: s = 16, the caller guarantees |x| < 4094).
// Structure = attn_fwd_nat_kernel (attn_bwd.hip): a workgroup owns 128 queries (one wave = 32, lane & 31 = query) and walks 64-key
// tiles; online softmax per lane in fp32; P^T is fed to the second product straight from the accumulators (permuted key order),
// V^T comes from transposed LDS reads of the row-major V tile.  Differences forced by the 4-byte storage:
//   * K / V tiles are REGISTER-staged (global_load_dwordx4 -> split -> ds_write_b64 into an fp16 hi tile and an fp16 lo tile): the
//     conversion needs the values in registers anyway, and the loads of tile kt+1 are in flight while tile kt is computed;
//   * Q never touches LDS: a lane's A... B-operand fragments are rows of its own query, loaded and split once;
//   * one LDS buffer per operand half (4 x 8 KiB), two barriers per key tile; two workgroups per CU cover each other's barriers.
// PS = true: qkv arrives PRE-SPLIT (the qkv conv's epilogue wrote [8 x fp16 hi | 8 x fp16 lo] per 8 channels, scaled per image from an
// a-priori table: eod_conv_desc.y_presplit_bound).  Then nothing is converted here: a lane's Q fragments are 16-byte loads of the
// finished hi / lo chunks, and the K / V tiles are staged by LDS-DMA (buffer_load ... lds, 64 lanes x 16 B = 8 key rows x 128 B per
// instruction, the bank swizzle applied on the SOURCE side; channel groups beyond the head dim come from an out-of-range offset =
// zeros), double buffered: the DMA of tile kt + 1 is in flight under the MFMAs of tile kt and the loop has ONE barrier per tile.
// Layout: qkv [N][T][3C] fp32 as produced by the qkv projection (channel = q_off / k_off / v_off + head*head_stride + j),
// out [N][T][C] fp32, optional lse [N][heads][T].  Any T, d % 8 == 0, d <= 64.
#include "common.h"
#include <type_traits>

typedef __fp16 fp16x4c __attribute__((__vector_size__(4 * sizeof(__fp16))));

constexpr int AX_ROWB = 128;       // LDS row = 64 halves (head dim padded with zeros)
constexpr int AX_KMIN = EOD_AB_KMIN_ATTN;  // smallest operand scale 2^-48 (common.h)

__device__ __forceinline__ int ax_swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int ax_off(int row, int c) { return row * AX_ROWB + ((c ^ ax_swz(row)) << 4); }

__device__ __forceinline__ half8 ax_row_frag(const char* tile, int row0, int ks, int lane) {
    const int row = row0 + (lane & 31);
    return *reinterpret_cast<const half8*>(tile + ax_off(row, 2 * ks + (lane >> 5)));
}

// fragment for a contraction over 16 tile ROWS in the permuted order {rb + 4kg + 0..3, rb + 8 + 4kg + 0..3} (kg = lane >> 5),
// columns j0 + (lane & 31): what the accumulator registers 8*kb .. 8*kb+7 of a 32x32 C tile hold (see attn_bwd.hip)
__device__ __forceinline__ half8 ax_tr_frag(const char* tile, int rb, int j0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int kg = g >> 1;
    const int col = j0 + (g & 1) * 16 + 4 * p;
    const int r0 = rb + 4 * kg + q, r1 = r0 + 8;
    const fp16x4c lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
        (__attribute__((address_space(3))) fp16x4c*)(tile + ax_off(r0, col >> 3) + ((col >> 2) & 1) * 8));
    const fp16x4c hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
        (__attribute__((address_space(3))) fp16x4c*)(tile + ax_off(r1, col >> 3) + ((col >> 2) & 1) * 8));
    half8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[e] = (half_t)lo[e];
        f[4 + e] = (half_t)hi[e];
    }
    return f;
}

__device__ __forceinline__ void ax_split(float x, half_t& hi, half_t& lo) {
    hi = (half_t)x;
    lo = (half_t)(x - (float)hi);
}

struct AttnX3P {
    const float* qkv;
    float* out;
    float* lse;
    int N, T, C, heads, d, q_off, k_off, v_off, hs;
    float scale_log2;  // log2(e) / sqrt(d)
    const float* ab;   // bound table [N][EOD_AB] of qkv, or NULL
    int out_ps;        // write `out` PRE-SPLIT ([8 x fp16 hi | 8 x fp16 lo] per 8 channels of s_n * out, s_n from `ab`: rows of out are
                       // convex combinations of v rows, so the table of qkv bounds them) for the proj_out conv (eod_conv_desc.x_presplit)
};

typedef __attribute__((address_space(3))) void ax_lds_void;

template <int DS, int DT, bool PS = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_nat_x3_kernel(const AttnX3P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 64 * AX_ROWB, BUF = 4 * TILE;  // one buffer = K hi | K lo | V hi | V lo tiles of [64][128 B]; PS: two buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, n = b / p.heads, h = b - n * p.heads;
    const int q0 = blockIdx.x * 128;
    const long long ld = 3LL * p.C;
    const float* base = p.qkv + (long long)n * p.T * ld;
    // operand scale of this image (wave-uniform), 1 / s, and the exponent's factor log2(e) / sqrt(d) / s^2 (exact: powers of two)
    AbScale asc = {16.0f, 1.0f};
    float out_s = 16.0f;  // scale of the pre-split output: what its consumer derives from the same table (default exponent range)
    if (p.ab) {
        const float B = ab_wave_bound(p.ab, n);
        asc = ab_scale_of(B, AX_KMIN);
        out_s = ab_scale_of(B).s;
    }
    const float AX_SCALE = asc.s, rscale = asc.inv * 0.0625f;
    const float scale_log2 = p.scale_log2 * rscale * rscale;

    // ---- this lane's query row, split once: fragment ks = channels 16 ks + 8 lh .. + 7 ----
    half8 qh[DS], ql[DS];
    {
        const int q = q0 + wave * 32 + lr;
        const float* qp = base + (long long)q * ld + p.q_off + h * p.hs;
#pragma unroll
        for (int ks = 0; ks < DS; ++ks) {
            const int j = 16 * ks + 8 * lh;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
            if (q < p.T && j < p.d) {
                a = *reinterpret_cast<const f32x4*>(qp + j);
                c = *reinterpret_cast<const f32x4*>(qp + j + 4);
            }
            if constexpr (PS) {  // the 8-channel group is already [8 x hi | 8 x lo]
                qh[ks] = __builtin_bit_cast(half8, a);
                ql[ks] = __builtin_bit_cast(half8, c);
                continue;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                half_t hi, lo;
                ax_split(a[e] * AX_SCALE, hi, lo);
                qh[ks][e] = hi; ql[ks][e] = lo;
                ax_split(c[e] * AX_SCALE, hi, lo);
                qh[ks][4 + e] = hi; ql[ks][4 + e] = lo;
            }
        }
    }

    // ---- K / V staging: slot s = tid + 256 i -> (row = s / 16, float4 column c4 = s % 16); masked slots are zeros ----
    f32x4 rk[4], rv[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s = tid + 256 * i, row = s >> 4, c4 = s & 15;
            const int key = kt * 64 + row;
            rk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (key < p.T && c4 * 4 < p.d) {
                const float* kp = base + (long long)key * ld + h * p.hs + c4 * 4;
                rk[i] = *reinterpret_cast<const f32x4*>(kp + p.k_off);
                rv[i] = *reinterpret_cast<const f32x4*>(kp + p.v_off);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s = tid + 256 * i, row = s >> 4, c4 = s & 15;
            const int off = ax_off(row, c4 >> 1) + (c4 & 1) * 8;
            half4 kh, kl, vh, vl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                half_t hi, lo;
                ax_split(rk[i][e] * AX_SCALE, hi, lo);
                kh[e] = hi; kl[e] = lo;
                ax_split(rv[i][e] * AX_SCALE, hi, lo);
                vh[e] = hi; vl[e] = lo;
            }
            *reinterpret_cast<half4*>(smem + off) = kh;
            *reinterpret_cast<half4*>(smem + TILE + off) = kl;
            *reinterpret_cast<half4*>(smem + 2 * TILE + off) = vh;
            *reinterpret_cast<half4*>(smem + 3 * TILE + off) = vl;
        }
    };

    // PS: LDS-DMA staging.  One instruction = 8 key rows x 8 chunk slots; slot j of row r holds channel group j ^ swz(r) (ax_off),
    // so the lane fetches THAT group's hi (or lo) chunk: source = key row + (group * 8 channels + 0 / 4) floats.  Wave w stages rows
    // 16 w .. 16 w + 15 of each of the four tiles: 8 instructions per key tile and wave.
    __amdgpu_buffer_rsrc_t rs_kv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
    auto dma_tile = [&](int kt, int buf) {
        if constexpr (PS) {
            char* b0 = smem + buf * BUF;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wave * 16 + i * 8 + (lane >> 3), key = kt * 64 + row;
                const int grp = (lane & 7) ^ ax_swz(row);
                const bool okr = key < p.T && grp * 8 < p.d;
                const unsigned rowoff = (unsigned)(((long long)key * ld + h * p.hs + grp * 8) * 4);
                const unsigned kofs = okr ? rowoff + (unsigned)p.k_off * 4u : 0x80000000u, vofs = okr ? rowoff + (unsigned)p.v_off * 4u : 0x80000000u;
                char* dst = b0 + (wave * 16 + i * 8) * AX_ROWB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kv, (ax_lds_void*)(dst), 16, kofs, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kv, (ax_lds_void*)(dst + TILE), 16, okr ? kofs + 16u : 0x80000000u, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kv, (ax_lds_void*)(dst + 2 * TILE), 16, vofs, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kv, (ax_lds_void*)(dst + 3 * TILE), 16, okr ? vofs + 16u : 0x80000000u, 0, 0, 0);
            }
        }
    };
    f32x16 o[DT];  // O^T tiles: registers = output channel j, lane & 31 = query
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const int nkt = (p.T + 63) / 64;
    if constexpr (PS) dma_tile(0, 0); else load_tile(0);
    // one key tile; RAGGED (compile-time) = the last tile of a sequence that is not a multiple of 64 keys: only that instance carries
    // the per-element key mask (see attn_fwd_nat_kernel in attn_bwd.hip)
    auto tile = [&](const int kt, auto ragged_c) {
        constexpr bool RAGGED = decltype(ragged_c)::value;
        char* const sKh = smem + (PS ? (kt & 1) * BUF : 0);
        char* const sKl = sKh + TILE;
        char* const sVh = sKh + 2 * TILE;
        char* const sVl = sKh + 3 * TILE;
        if constexpr (PS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of tile kt have landed ...
            __builtin_amdgcn_s_barrier();                     // ... and everybody's; every wave is also done reading tile kt - 1
            if (kt + 1 < nkt) dma_tile(kt + 1, (kt + 1) & 1);  // into the buffer tile kt - 1 just vacated, in flight under the MFMAs below
        } else {
        __syncthreads();  // every wave is done reading the previous tile
        store_tile();     // (waits for this tile's global loads)
        __syncthreads();  // tile kt is visible
        if (kt + 1 < nkt) load_tile(kt + 1);  // in flight under the MFMAs below
        }
        f32x16 s[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[mt][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < DS; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const half8 kh = ax_row_frag(sKh, mt * 32, ks, lane), kl = ax_row_frag(sKl, mt * 32, ks, lane);
                s[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s[mt], 0, 0, 0);
                s[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s[mt], 0, 0, 0);
                s[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s[mt], 0, 0, 0);
            }
        // ---- online softmax over this tile's 64 keys (32 in this lane's registers, 32 in lane ^ 32), fp32 ----
        float mloc = -INFINITY;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (RAGGED) {
                    const int key = kt * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    s[mt][r] = key < p.T ? s[mt][r] : -INFINITY;
                }
                mloc = fmaxf(mloc, s[mt][r]);  // raw scores: the (positive) scale is applied inside the exponent's fma below -- one rounding
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc * scale_log2);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // raw v_exp_f32 (1 ulp; arguments <= 0, underflow to 0 is the intent)
        m_run = m_new;
        float lsum = 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[mt][r], scale_log2, -m_new));
                s[mt][r] = e;
                lsum += e;
            }
        l_run = l_run * alpha + lsum;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
        // ---- O^T += V^T P^T, P split from the accumulator registers ----
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                half8 ph, pl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    half_t hi, lo;
                    ax_split(s[mt][8 * kb + e], hi, lo);
                    ph[e] = hi; pl[e] = lo;
                }
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const half8 vh = ax_tr_frag(sVh, mt * 32 + 16 * kb, t * 32, lane), vl = ax_tr_frag(sVl, mt * 32 + 16 * kb, t * 32, lane);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[t], 0, 0, 0);
                }
            }
    };
    const int nfull = (p.T & 63) ? nkt - 1 : nkt;
    for (int kt = 0; kt < nfull; ++kt) tile(kt, std::false_type{});
    if (nfull < nkt) tile(nkt - 1, std::true_type{});
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = rscale / l_tot;
    const int q = q0 + wave * 32 + lr;
    if (q < p.T) {
        if (p.lse && lh == 0) p.lse[((long long)n * p.heads + h) * p.T + q] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
        float* op = p.out + ((long long)n * p.T + q) * p.C + h * p.d;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int j0 = t * 32 + 8 * g4 + 4 * lh;  // registers 4*g4 .. 4*g4+3 = 4 consecutive output channels
                if (j0 + 3 < p.d) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = o[t][4 * g4 + e] * inv;
                    if (p.out_ps) {
                        // lanes l (lh = 0) and l + 32 (lh = 1) hold the two halves of one 8-channel group of the same query: the first
                        // stores [hi 0-3 | hi 4-7] at the group's first 16 bytes, the second [lo 0-3 | lo 4-7] behind it
                        half4 h4, l4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            half_t hi, lo;
                            ax_split(v[e] * out_s, hi, lo);
                            h4[e] = hi; l4[e] = lo;
                        }
                        typedef int i32x2 __attribute__((ext_vector_type(2)));
                        const i32x2 hb = __builtin_bit_cast(i32x2, h4), lb = __builtin_bit_cast(i32x2, l4);
                        const int s0 = lh ? hb[0] : lb[0], s1 = lh ? hb[1] : lb[1];
                        const int r0 = __shfl_xor(s0, 32), r1 = __shfl_xor(s1, 32);
                        const i32x4 w = lh ? i32x4{r0, r1, lb[0], lb[1]} : i32x4{hb[0], hb[1], r0, r1};
                        *reinterpret_cast<i32x4*>(op + j0) = w;
                    } else {
                        *reinterpret_cast<f32x4*>(op + j0) = v;
                    }
                }
            }
    }
}

int eod_attention_fwd_nat_x3(const float* qkv, float* out, float* lse, int N, int T, int C, int heads, int d, int q_off, int k_off, int v_off,
                             int head_stride, const float* qkv_bound, int out_presplit, int in_presplit, hipStream_t st) {
    EOD_REQUIRE(q_off % 4 == 0 && k_off % 4 == 0 && v_off % 4 == 0 && head_stride % 4 == 0 && eod_aligned16(qkv) && eod_aligned16(out) && C % 4 == 0,
                "attention_fwd_nat (fp32): alignment of the head slices");
    AttnX3P p;
    p.qkv = qkv; p.out = out; p.lse = lse;
    p.N = N; p.T = T; p.C = C; p.heads = heads; p.d = d; p.q_off = q_off; p.k_off = k_off; p.v_off = v_off; p.hs = head_stride;
    p.scale_log2 = 1.4426950408889634f / sqrtf((float)d);
    p.ab = qkv_bound;
    p.out_ps = out_presplit;
    const dim3 grid((T + 127) / 128, N * heads);
    const int ds = (d + 15) / 16;
    if (in_presplit) {
        EOD_REQUIRE(qkv_bound && q_off % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && head_stride % 8 == 0 && C % 8 == 0,
                    "attention_fwd_nat (fp32, pre-split qkv): needs the table the producer scaled by, and whole 8-channel groups");
        EOD_REQUIRE((long long)T * 3 * C * 4 < 0x7fffffffLL, "attention_fwd_nat (fp32, pre-split qkv): one image of qkv exceeds the 2 GiB window");
        const size_t lds2 = (size_t)2 * 256 * AX_ROWB;  // two buffers of four tiles (64 KiB: two workgroups per CU)
#define EOD_AX_PS(DS_, DT_)                                                                                                          \
        do {                                                                                                                     \
            static bool attr = false;                                                                                            \
            if (!attr) {                                                                                                         \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_nat_x3_kernel<DS_, DT_, true>),                 \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);                               \
                attr = true;                                                                                                     \
            }                                                                                                                    \
            hipLaunchKernelGGL((attn_fwd_nat_x3_kernel<DS_, DT_, true>), grid, dim3(256), lds2, st, p);                          \
        } while (0)
        if (ds == 1) EOD_AX_PS(1, 1);
        else if (ds == 2) EOD_AX_PS(2, 1);
        else if (ds == 3) EOD_AX_PS(3, 2);
        else EOD_AX_PS(4, 2);
#undef EOD_AX_PS
        EOD_CHECK_LAUNCH("attention_fwd_nat (fp32, pre-split qkv)");
        return EOD_OK;
    }
    const size_t lds = (size_t)256 * AX_ROWB;
    if (ds == 1) hipLaunchKernelGGL((attn_fwd_nat_x3_kernel<1, 1>), grid, dim3(256), lds, st, p);
    else if (ds == 2) hipLaunchKernelGGL((attn_fwd_nat_x3_kernel<2, 1>), grid, dim3(256), lds, st, p);
    else if (ds == 3) hipLaunchKernelGGL((attn_fwd_nat_x3_kernel<3, 2>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((attn_fwd_nat_x3_kernel<4, 2>), grid, dim3(256), lds, st, p);
    EOD_CHECK_LAUNCH("attention_fwd_nat (fp32)");
    return EOD_OK;
}
