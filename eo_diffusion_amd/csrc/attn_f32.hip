// Fused attention forward in EXACT fp32 (the "fp32" precision mode): fp32 storage, IEEE fp32 products on v_mfma_f32_32x32x2_f32,
// fp32 online softmax -- the T x T weights of QKVAttention(Legacy) (unet_openai.py:476-480, 508-514) never exist in HBM in this mode
// either (the reference materialises them: 4.3 GB at 256 x 256, batch 8, 8 heads).
//     out[n, t, h*d + j] = sum_s softmax_s(q_t . k_s / sqrt(d)) v[s, j]
// Structure = attn_fwd_nat_x3_kernel (attn_x3.hip): a workgroup owns 128 queries (one wave = 32, lane & 31 = query) and walks
// 64-key tiles; S^T = K Q^T with the keys in the accumulator registers, online softmax per lane, O^T += V^T P^T with P^T fed to the
// second product straight from those registers.  What the 32x32x2 shape changes:
//   * an MFMA contracts over TWO channels (k = lane >> 5).  Channel pairing {c, c + d/2}: lane half 0 walks channels 0 .. d/2 - 1,
//     half 1 walks d/2 .. d - 1, so a lane's operands for four consecutive K-steps are ONE 16-byte LDS read (K) / four registers (Q);
//   * K tile rows are padded to d + 4 floats: the 16 lanes of a ds_read_b128 group read 16 consecutive keys at one column, and
//     (d + 4) * 4 bytes is an odd multiple of 16 for every d % 8 == 0, i.e. 16 distinct 16-byte slots of the 256-byte bank row;
//   * the second product needs no transposed reads: accumulator register r of the score tile holds, in lane half lh, the key
//     8 (r >> 2) + (r & 3) + 4 lh -- exactly the two k values of one 32x32x2 step -- and its A operand V^T[channel][key] is one
//     ds_read_b32 along a row of the row-major V tile (32 consecutive channels of one key: conflict-free).
// Layout: qkv [N][T][3C] fp32 (channel = q_off / k_off / v_off + head*head_stride + j), out [N][T][C] fp32, optional lse [N][heads][T].
// Any T, d % 8 == 0, d <= 64.  MFMA-bound by construction (128 x 64-cycle MFMAs per key tile and wave at d = 64 against ~250 VALU).
#include "common.h"
#include <type_traits>

struct AttnF32P {
    const float* qkv;
    float* out;
    float* lse;
    int N, T, C, heads, d, q_off, k_off, v_off, hs;
    float scale_log2;  // log2(e) / sqrt(d)
};

// DQ = d / 8 (16-byte groups per lane half), DT = ceil(d / 32) output-channel tiles
template <int DQ, int DT>
__global__ __launch_bounds__(256, 2) void attn_fwd_nat_f32_kernel(const AttnF32P p) {
    constexpr int D = 8 * DQ, KS = D + 4;          // head dim, K-tile row stride (floats)
    constexpr int KTILE = 64 * KS, VTILE = 64 * D;  // floats
    extern __shared__ __attribute__((aligned(16))) float smf[];
    float* sK = smf;          // [64][KS]
    float* sV = smf + KTILE;  // [64][D]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, n = b / p.heads, h = b - n * p.heads;
    const int q0 = blockIdx.x * 128;
    const long long ld = 3LL * p.C;
    const float* base = p.qkv + (long long)n * p.T * ld;

    // ---- this lane's query row: channels lh * d/2 + 0 .. d/2 - 1 (its k value of every K-step) ----
    f32x4 qf[DQ];
    {
        const int q = q0 + wave * 32 + lr;
        const float* qp = base + (long long)q * ld + p.q_off + h * p.hs + lh * (D / 2);
#pragma unroll
        for (int g = 0; g < DQ; ++g) {
            qf[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (q < p.T) qf[g] = *reinterpret_cast<const f32x4*>(qp + 4 * g);
        }
    }

    // ---- K / V staging through registers: slot s = tid + 256 i -> (row = s / (D/4), float4 column = s % (D/4)) ----
    constexpr int C4 = D / 4, SLOTS = 64 * C4, NLD = (SLOTS + 255) / 256;
    f32x4 rk[NLD], rv[NLD];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int s = tid + 256 * i, row = s / C4, c4 = s - row * C4;
            const int key = kt * 64 + row;
            rk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (s < SLOTS && key < p.T) {
                const float* kp = base + (long long)key * ld + h * p.hs + c4 * 4;
                rk[i] = *reinterpret_cast<const f32x4*>(kp + p.k_off);
                rv[i] = *reinterpret_cast<const f32x4*>(kp + p.v_off);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int s = tid + 256 * i, row = s / C4, c4 = s - row * C4;
            if (s < SLOTS) {
                *reinterpret_cast<f32x4*>(sK + row * KS + c4 * 4) = rk[i];
                *reinterpret_cast<f32x4*>(sV + row * D + c4 * 4) = rv[i];
            }
        }
    };

    f32x16 o[DT];  // O^T tiles: registers = output channel, lane & 31 = query
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const int nkt = (p.T + 63) / 64;
    load_tile(0);
    auto tile = [&](const int kt, auto ragged_c) {
        constexpr bool RAGGED = decltype(ragged_c)::value;
        __syncthreads();  // every wave is done reading the previous tile
        store_tile();     // (waits for this tile's global loads)
        __syncthreads();  // tile kt is visible
        if (kt + 1 < nkt) load_tile(kt + 1);  // in flight under the MFMAs below
        // ---- S^T = K Q^T: rows = keys (mt * 32 + lane & 31), k = channel pair {c, c + d/2} ----
        f32x16 s[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[mt][r] = 0.0f;
#pragma unroll
        for (int g = 0; g < DQ; ++g)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(sK + (mt * 32 + lr) * KS + lh * (D / 2) + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[g][e], s[mt], 0, 0, 0);
            }
        // ---- online softmax over this tile's 64 keys (32 in this lane's registers, 32 in lane ^ 32) ----
        float mloc = -INFINITY;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (RAGGED) {
                    const int key = kt * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    s[mt][r] = key < p.T ? s[mt][r] : -INFINITY;
                }
                mloc = fmaxf(mloc, s[mt][r]);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc * p.scale_log2);
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float lsum = 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = exp2f(fmaf(s[mt][r], p.scale_log2, -m_new));
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
        // ---- O^T += V^T P^T: register r of s[mt] is the B operand of the step over keys {kappa, kappa + 4} ----
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const int j = t * 32 + lr;
                    const float vf = (D % 32 == 0 || j < D) ? sV[key * D + j] : 0.0f;
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[mt][r], o[t], 0, 0, 0);
                }
            }
    };
    const int nfull = (p.T & 63) ? nkt - 1 : nkt;
    for (int kt = 0; kt < nfull; ++kt) tile(kt, std::false_type{});
    if (nfull < nkt) tile(nkt - 1, std::true_type{});
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int q = q0 + wave * 32 + lr;
    if (q < p.T) {
        if (p.lse && lh == 0) p.lse[((long long)n * p.heads + h) * p.T + q] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
        float* op = p.out + ((long long)n * p.T + q) * p.C + h * p.d;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int j0 = t * 32 + 8 * g4 + 4 * lh;  // registers 4*g4 .. 4*g4+3 = 4 consecutive output channels
                if (j0 + 3 < D) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = o[t][4 * g4 + e] * inv;
                    *reinterpret_cast<f32x4*>(op + j0) = v;
                }
            }
    }
}

template <int DQ> static void launch_attn_f32(const AttnF32P& p, dim3 grid, hipStream_t st) {
    constexpr int D = 8 * DQ, DT = (D + 31) / 32;
    const size_t lds = (size_t)(64 * (D + 4) + 64 * D) * sizeof(float);
    hipLaunchKernelGGL((attn_fwd_nat_f32_kernel<DQ, DT>), grid, dim3(256), lds, st, p);
}

int eod_attention_fwd_nat_f32(const float* qkv, float* out, float* lse, int N, int T, int C, int heads, int d, int q_off, int k_off, int v_off,
                              int head_stride, hipStream_t st) {
    EOD_REQUIRE(q_off % 4 == 0 && k_off % 4 == 0 && v_off % 4 == 0 && head_stride % 4 == 0 && eod_aligned16(qkv) && eod_aligned16(out) && C % 4 == 0,
                "attention_fwd_nat (exact fp32): alignment of the head slices");
    EOD_REQUIRE(d % 8 == 0 && d >= 8 && d <= 64, "attention_fwd_nat (exact fp32): the head dim must be a multiple of 8 and <= 64");
    AttnF32P p;
    p.qkv = qkv; p.out = out; p.lse = lse;
    p.N = N; p.T = T; p.C = C; p.heads = heads; p.d = d; p.q_off = q_off; p.k_off = k_off; p.v_off = v_off; p.hs = head_stride;
    p.scale_log2 = 1.4426950408889634f / sqrtf((float)d);
    const dim3 grid((T + 127) / 128, N * heads);
    switch (d / 8) {
        case 1: launch_attn_f32<1>(p, grid, st); break;
        case 2: launch_attn_f32<2>(p, grid, st); break;
        case 3: launch_attn_f32<3>(p, grid, st); break;
        case 4: launch_attn_f32<4>(p, grid, st); break;
        case 5: launch_attn_f32<5>(p, grid, st); break;
        case 6: launch_attn_f32<6>(p, grid, st); break;
        case 7: launch_attn_f32<7>(p, grid, st); break;
        default: launch_attn_f32<8>(p, grid, st); break;
    }
    EOD_CHECK_LAUNCH("attention_fwd_nat (exact fp32)");
    return EOD_OK;
}
