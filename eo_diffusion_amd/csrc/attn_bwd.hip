// Flash-style backward of softmax(q k^T / sqrt(d)) v  (QKVAttention(Legacy), unet_openai.py:465-515), fp16, head dim <= 64.
//
// Nothing T x T is read from or written to HBM: every tile of P is rebuilt from q, k and the forward's per-row log-sum-exp
//     P[t][s] = exp(q_t.k_s / sqrt(d) - lse[t]),   dP[t][s] = dO_t.v_s,   dS = P * (dP - D[t]),   D[t] = dO_t.O_t
// Two kernels (no atomics, fixed summation order):
//   * attn_bwd_kv_kernel: a workgroup owns 128 keys (one wave = 32 keys) and walks all query tiles:
//         S = Q K^T, dP = dO V^T (C layout: lane = key, registers = queries);  dV += P^T dO,  dK += dS^T Q / sqrt(d)
//   * attn_bwd_q_kernel: a workgroup owns 128 queries (one wave = 32 queries) and walks all key tiles:
//         S^T = K Q^T, dP^T = V dO^T (lane = query, registers = keys);  dQ += dS K / sqrt(d)
// In both, the SECOND product contracts over the axis that lives in the accumulator REGISTERS of the first one.  The C layout of a
// 32x32 MFMA gives a lane 16 values of that axis in the order {0-3, 8-11, 16-19, 24-27} (+4 for the upper half-wave); taken eight
// at a time they are exactly the A-operand fragment of a K=16 MFMA for a PERMUTED order of the contraction index -- and the B
// operand (rows of dO / Q / K, staged row-major in LDS as they lie in HBM) is fetched in the same permuted order with two
// ds_read_b64_tr_b16 (4 consecutive rows each).  So P / dS never leave the registers.
// Layout of the operands: qkv [N][T][3C] as produced by the qkv conv (channel = q_off/k_off/v_off + head*head_stride + j),
// dO / O [N][T][C], lse / D [N][heads][T] fp32, dqkv [N][T][3C].  Any T (rows / keys beyond T are zero-filled and masked through
// lse = +inf / P = 0), d % 8 == 0, d <= 64.
#include "common.h"
#include <type_traits>

typedef __fp16 fp16x4b __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) void lds_void_b;

// P and dS enter the second pair of MFMAs as fp16.  Their natural magnitudes are ~1/T and ~(1/T) x |dP - D|: at T = 4096 that is
// 2e-4 and 1e-6 and below -- the fp16 SUBNORMAL range (min normal 6.1e-5), where every halving of the value costs a mantissa bit
// (measured at 512 x 512 x 13, T = 4096 / 16384: 6.6 % error on the qkv weight gradients against the exact-fp32 mode).  Both are
// therefore multiplied by an exact power of two before the conversion and the accumulators divided by it once at the end:
// P x 2^8 (P <= 1 -> at most 256), dS x 2^min(12, floor(log2 T)): the 2^12 that T >= 4096 needs overflows fp16 once P |dP - D| > 16, which
// a SHORT sequence reaches at an ordinary loss scale (T = 4: P ~ 1/4, and the mean-reduced loss of a tiny prediction tensor makes dP
// large; found by tests/test_gpu_fuzz_archs.py, training case 15: NaN gradients at the default loss scale of 1024) -- P ~ 1/T is what
// the scale compensates, so it follows T (AttnBwdP::ds_log2; unchanged for T >= 4096).  An inf that still occurs is caught by the
// guarded optimizer step like any other fp16 overflow.
#define AB_P_SCALE 256.0f
// The softmax work per score element is what bounds these kernels at small head dims (d = 32: ~8 VALU slots per element against 2 x 32
// MACs), so the scales cost nothing: they ride in the exponent.  P x 2^8 = exp2(s * alpha*log2(e) - (lse*log2(e) - 8)) is ONE fma + one
// v_exp_f32 with the per-row constant prepared where lse is staged; dS x 2^k = (P x 2^8) * (2^(k-8) dP - 2^(k-8) D) is one fma + one multiply.
#define AB_LOG2E 1.4426950408889634f
#define AB_P_LOG2 8.0f                                  // log2(AB_P_SCALE)

struct AttnBwdP {
    const char* qkv;
    const char* dO;
    const float* lse;
    const float* D;
    char* dqkv;
    int N, T, C, heads, d, q_off, k_off, v_off, hs;
    float alpha;
    float ds_log2;   // log2 of the scale dS is carried on: min(12, floor(log2 T))
};

constexpr int AB_ROWB = 128;  // LDS row = 64 halves (head dim padded with zeros)

__device__ __forceinline__ int ab_swz(int row) { return (row >> 1) & 7; }
// 16-byte chunk `c` of tile row `row`
__device__ __forceinline__ int ab_off(int row, int c) { return row * AB_ROWB + ((c ^ ab_swz(row)) << 4); }

// stage `rows` consecutive sequence positions (starting at t0) of one head's d channels into an LDS tile (rows x 128 B):
// one DMA instruction = 8 rows x 8 chunks; group g covers rows 8g..8g+7
__device__ __forceinline__ void ab_stage(const __amdgpu_buffer_rsrc_t rs, long long row_stride_b, long long base_b, int t0, int ngroups, int d,
                                         char* tile, int wave, int lane, int nwaves, int T = 0x7fffffff) {
    const int r8 = lane >> 3, slot = lane & 7;
    for (int g = wave; g < ngroups; g += nwaves) {
        const int row = g * 8 + r8;
        const int c = slot ^ ab_swz(row);
        const unsigned v = (c * 8 < d && t0 + row < T) ? (unsigned)(base_b + (long long)(t0 + row) * row_stride_b + c * 16) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_b*)(tile + g * 1024), 16, v, 0, 0, 0);
    }
}

// A / B fragment of a row-major tile for a K=16 slice `ks` of the head dim: 16 bytes of row (lane & 31), chunk 2*ks + (lane >> 5)
__device__ __forceinline__ half8 ab_row_frag(const char* tile, int row0, int ks, int lane) {
    const int row = row0 + (lane & 31);
    return *reinterpret_cast<const half8*>(tile + ab_off(row, 2 * ks + (lane >> 5)));
}

// B fragment for a contraction over 16 tile ROWS in the permuted order of the header comment: rows rb + 4*kg + {0..3} and
// rb + 8 + 4*kg + {0..3} (kg = lane >> 5), columns j0 + (lane & 31)
__device__ __forceinline__ half8 ab_tr_frag(const char* tile, int rb, int j0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int kg = g >> 1;
    const int col = j0 + (g & 1) * 16 + 4 * p;  // first of this lane's 4 address columns
    const int r0 = rb + 4 * kg + q, r1 = r0 + 8;
    const fp16x4b lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
        (__attribute__((address_space(3))) fp16x4b*)(tile + ab_off(r0, col >> 3) + ((col >> 2) & 1) * 8));
    const fp16x4b hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
        (__attribute__((address_space(3))) fp16x4b*)(tile + ab_off(r1, col >> 3) + ((col >> 2) & 1) * 8));
    half8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[e] = (half_t)lo[e];
        f[4 + e] = (half_t)hi[e];
    }
    return f;
}

// the 8 accumulator registers 8*kb .. 8*kb+7 of a C tile as an fp16 A fragment (see header)
__device__ __forceinline__ half8 ab_acc_frag(const f32x16& c, int kb) {
    half8 f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (half_t)c[8 * kb + e];
    return f;
}

// =============================================================================================
// dK, dV: grid (T/128, N*heads), 256 threads.  DS = ceil(d/16) K-slices of the head dim, DT = ceil(d/32) output column tiles.
// =============================================================================================
template <int DS, int DT>
__global__ __launch_bounds__(256, 2) void attn_bwd_kv_kernel(const AttnBwdP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                   // [128][128 B]
    char* sV = smem + 128 * AB_ROWB;   // [128][128 B]
    char* sQ = smem + 256 * AB_ROWB;   // [2][64][128 B]
    char* sO = sQ + 2 * 64 * AB_ROWB;  // [2][64][128 B]  (dO)
    float* sL = reinterpret_cast<float*>(sO + 2 * 64 * AB_ROWB);  // [2][64] lse, then [2][64] D
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float ds_over_p = exp2f(p.ds_log2 - AB_P_LOG2), ds_inv = exp2f(-p.ds_log2);  // exact powers of two
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, n = b / p.heads, h = b - n * p.heads;
    const int s0 = blockIdx.x * 128;
    const long long rs3 = (long long)3 * p.C * 2, rsC = (long long)p.C * 2;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.qkv) + (long long)n * p.T * rs3, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.dO) + (long long)n * p.T * rsC, 0, 0x7fffffff, 0x00020000);
    const long long qb = (long long)(p.q_off + h * p.hs) * 2, kb_ = (long long)(p.k_off + h * p.hs) * 2, vb = (long long)(p.v_off + h * p.hs) * 2;
    const long long ob = (long long)h * p.d * 2;
    const float* lse = p.lse + ((long long)n * p.heads + h) * p.T;
    const float* Dv = p.D + ((long long)n * p.heads + h) * p.T;

    ab_stage(rq, rs3, kb_, s0, 16, p.d, sK, wave, lane, 4, p.T);
    ab_stage(rq, rs3, vb, s0, 16, p.d, sV, wave, lane, 4, p.T);
    auto stage_q = [&](int qt, int buf) {
        ab_stage(rq, rs3, qb, qt * 64, 8, p.d, sQ + buf * 64 * AB_ROWB, wave, lane, 4, p.T);
        ab_stage(ro, rsC, ob, qt * 64, 8, p.d, sO + buf * 64 * AB_ROWB, wave, lane, 4, p.T);
        if (tid < 64) {  // rows beyond T: lse = +inf makes their P exactly 0
            const bool in = qt * 64 + tid < p.T;
            sL[buf * 64 + tid] = in ? lse[qt * 64 + tid] * AB_LOG2E - AB_P_LOG2 : INFINITY;   // exponent offset of P x 2^8 (log2 domain)
            sL[128 + buf * 64 + tid] = in ? Dv[qt * 64 + tid] * ds_over_p : 0.0f;            // 2^(k-8) D
        }
    };
    stage_q(0, 0);

    f32x16 dv[DT], dk[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[t][r] = dk[t][r] = 0.0f;

    const int nqt = (p.T + 63) / 64;
    // Keys beyond T need no mask here: a key is a LANE of both products' outputs (dV / dK rows), its column of P / dS feeds only its
    // own two rows, and rows >= T are never stored (K / V rows beyond T are zero-filled, so those values are finite).
    const float a2 = p.alpha * AB_LOG2E;
    for (int qt = 0; qt < nqt; ++qt) {
        const int buf = qt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (qt + 1 < nqt) stage_q(qt + 1, buf ^ 1);
        const char* tQ = sQ + buf * 64 * AB_ROWB;
        const char* tO = sO + buf * 64 * AB_ROWB;
        const float* tL = sL + buf * 64;
        const float* tD = sL + 128 + buf * 64;
        // ---- S = Q K^T, dP = dO V^T for 64 queries x this wave's 32 keys ----
        f32x16 s[2], dp[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[mt][r] = dp[mt][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < DS; ++ks) {
            const half8 fk = ab_row_frag(sK, wave * 32, ks, lane);
            const half8 fv = ab_row_frag(sV, wave * 32, ks, lane);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const half8 fq = ab_row_frag(tQ, mt * 32, ks, lane);
                const half8 fo = ab_row_frag(tO, mt * 32, ks, lane);
                s[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fq, fk, s[mt], 0, 0, 0);
                dp[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fo, fv, dp[mt], 0, 0, 0);
            }
        }
        // ---- P = exp(alpha S - lse[q]),  dS = P (dP - D[q]);  rows (registers) are queries: q = mt*32 + (r&3) + 8*(r>>2) + 4*lh ----
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int qrow = mt * 32 + 8 * g4 + 4 * lh;
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(tL + qrow);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(tD + qrow);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pe = __builtin_amdgcn_exp2f(fmaf(s[mt][4 * g4 + e], a2, -l4[e]));  // P x 2^8 (0 for rows beyond T: l4 = +inf)
                    s[mt][4 * g4 + e] = pe;
                    dp[mt][4 * g4 + e] = pe * fmaf(dp[mt][4 * g4 + e], ds_over_p, -d4[e]);  // dS x 2^k
                }
            }
        // ---- dV += P^T dO,  dK += dS^T Q   (contraction over the 64 queries, permuted order) ----
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const half8 fp = ab_acc_frag(s[mt], kb);
                const half8 fs = ab_acc_frag(dp[mt], kb);
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const half8 bo = ab_tr_frag(tO, mt * 32 + 16 * kb, t * 32, lane);
                    const half8 bq = ab_tr_frag(tQ, mt * 32 + 16 * kb, t * 32, lane);
                    dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fp, bo, dv[t], 0, 0, 0);
                    dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fs, bq, dk[t], 0, 0, 0);
                }
            }
    }
    // ---- store: C layout lane = channel j, registers = keys ----
    half_t* out = reinterpret_cast<half_t*>(p.dqkv) + (long long)n * p.T * 3 * p.C;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        const int j = t * 32 + lr;
        if (j < p.d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int srow = s0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (srow >= p.T) continue;
                half_t* row = out + (long long)srow * 3 * p.C;
                row[p.v_off + h * p.hs + j] = (half_t)(dv[t][r] * (1.0f / AB_P_SCALE));
                row[p.k_off + h * p.hs + j] = (half_t)(dk[t][r] * (p.alpha * ds_inv));
            }
        }
    }
}

// =============================================================================================
// dQ: grid (T/128, N*heads), 256 threads; a wave owns 32 queries, lane & 31 = query in every first-product C tile
// =============================================================================================
template <int DS, int DT>
__global__ __launch_bounds__(256, 2) void attn_bwd_q_kernel(const AttnBwdP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQ = smem;                   // [128][128 B]
    char* sO = smem + 128 * AB_ROWB;   // [128][128 B]  (dO)
    char* sK = smem + 256 * AB_ROWB;   // [2][64][128 B]
    char* sV = sK + 2 * 64 * AB_ROWB;  // [2][64][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, n = b / p.heads, h = b - n * p.heads;
    const int q0 = blockIdx.x * 128;
    const long long rs3 = (long long)3 * p.C * 2, rsC = (long long)p.C * 2;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.qkv) + (long long)n * p.T * rs3, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.dO) + (long long)n * p.T * rsC, 0, 0x7fffffff, 0x00020000);
    const long long qb = (long long)(p.q_off + h * p.hs) * 2, kb_ = (long long)(p.k_off + h * p.hs) * 2, vb = (long long)(p.v_off + h * p.hs) * 2;
    const long long ob = (long long)h * p.d * 2;
    const int myq = q0 + wave * 32 + lr;
    // exponent offset of P x 2^12 in the log2 domain (+inf: P = 0 for rows beyond T)
    const float my_l2 = myq < p.T ? p.lse[((long long)n * p.heads + h) * p.T + myq] * AB_LOG2E - p.ds_log2 : INFINITY;
    const float my_D = myq < p.T ? p.D[((long long)n * p.heads + h) * p.T + myq] : 0.0f;
    const float a2 = p.alpha * AB_LOG2E;

    ab_stage(rq, rs3, qb, q0, 16, p.d, sQ, wave, lane, 4, p.T);
    ab_stage(ro, rsC, ob, q0, 16, p.d, sO, wave, lane, 4, p.T);
    auto stage_k = [&](int kt, int buf) {
        ab_stage(rq, rs3, kb_, kt * 64, 8, p.d, sK + buf * 64 * AB_ROWB, wave, lane, 4, p.T);
        ab_stage(rq, rs3, vb, kt * 64, 8, p.d, sV + buf * 64 * AB_ROWB, wave, lane, 4, p.T);
    };
    stage_k(0, 0);
    f32x16 dq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[t][r] = 0.0f;

    const int nkt = (p.T + 63) / 64;
    // one key tile; RAGGED (compile-time) = the last tile of a sequence that is not a multiple of 64 keys: only that instance masks keys
    // beyond T.  A zero-filled key row gives s = 0 and dP = 0, so dS = -exp(-lse) 2^12 D there: finite in fp32 and multiplied by a
    // zero K row, but it enters the MFMA as fp16 -- once every logit of a query row is very negative (lse << 0) it overflows to inf,
    // and inf x 0 is NaN in that query's dQ.  Full tiles have no such rows and keep the unmasked form.
    auto tile = [&](const int kt, auto ragged_c) {
        constexpr bool RAGGED = decltype(ragged_c)::value;
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) stage_k(kt + 1, buf ^ 1);
        const char* tK = sK + buf * 64 * AB_ROWB;
        const char* tV = sV + buf * 64 * AB_ROWB;
        // ---- S^T = K Q^T, dP^T = V dO^T: 64 keys (registers) x this wave's 32 queries (lanes) ----
        f32x16 s[2], dp[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[mt][r] = dp[mt][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < DS; ++ks) {
            const half8 fq = ab_row_frag(sQ, wave * 32, ks, lane);
            const half8 fo = ab_row_frag(sO, wave * 32, ks, lane);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const half8 fk = ab_row_frag(tK, mt * 32, ks, lane);
                const half8 fv = ab_row_frag(tV, mt * 32, ks, lane);
                s[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fk, fq, s[mt], 0, 0, 0);
                dp[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fv, fo, dp[mt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pe = __builtin_amdgcn_exp2f(fmaf(s[mt][r], a2, -my_l2));  // P x 2^12
                if constexpr (RAGGED) {
                    const int key = kt * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    pe = key < p.T ? pe : 0.0f;
                }
                dp[mt][r] = pe * (dp[mt][r] - my_D);  // dS^T x 2^12
            }
        // ---- dQ += dS K  (contraction over the 64 keys held in registers, permuted order; B = K rows via transposed reads) ----
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const half8 fs = ab_acc_frag(dp[mt], kb);
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const half8 bk = ab_tr_frag(tK, mt * 32 + 16 * kb, t * 32, lane);
                    dq[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fs, bk, dq[t], 0, 0, 0);
                }
            }
    };
    const int nfull = (p.T & 63) ? nkt - 1 : nkt;
    for (int kt = 0; kt < nfull; ++kt) tile(kt, std::false_type{});
    if (nfull < nkt) tile(nkt - 1, std::true_type{});
    half_t* out = reinterpret_cast<half_t*>(p.dqkv) + (long long)n * p.T * 3 * p.C;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        const int j = t * 32 + lr;
        if (j < p.d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qrow = q0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (qrow >= p.T) continue;
                out[(long long)qrow * 3 * p.C + p.q_off + h * p.hs + j] = (half_t)(dq[t][r] * (p.alpha * exp2f(-p.ds_log2)));
            }
        }
    }
}

// =============================================================================================
// Forward on the NATURAL qkv layout (no packed q|k / transposed v operands): same structure as attn_bwd_q_kernel.  A workgroup owns
// 128 queries (one wave = 32, lane & 31 = query), walks the key tiles:  S^T = K Q^T (registers = keys), online softmax per lane,
// O^T += V^T P^T with V^T fetched by transposed LDS reads from the row-major V tile and P^T fed from the accumulator registers
// (C[m = channel][n = query]: the rescale factor of the online softmax is a per-lane scalar).  Any T (keys beyond T are masked,
// rows beyond T are zero-filled and not stored); optional log-sum-exp output for the training path.
// =============================================================================================
struct AttnFwdP {
    const char* qkv;
    char* out;
    float* lse;
    int N, T, C, heads, d, q_off, k_off, v_off, hs;
    float scale_log2;
};

template <int DS, int DT>
__global__ __launch_bounds__(256, 2) void attn_fwd_nat_kernel(const AttnFwdP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQ = smem;                   // [128][128 B]
    char* sK = smem + 128 * AB_ROWB;   // [2][64][128 B]
    char* sV = sK + 2 * 64 * AB_ROWB;  // [2][64][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, n = b / p.heads, h = b - n * p.heads;
    const int q0 = blockIdx.x * 128;
    const long long rs3 = (long long)3 * p.C * 2;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.qkv) + (long long)n * p.T * rs3, 0, 0x7fffffff, 0x00020000);
    const long long qb = (long long)(p.q_off + h * p.hs) * 2, kb_ = (long long)(p.k_off + h * p.hs) * 2, vb = (long long)(p.v_off + h * p.hs) * 2;
    ab_stage(rq, rs3, qb, q0, 16, p.d, sQ, wave, lane, 4, p.T);
    auto stage_k = [&](int kt, int buf) {
        ab_stage(rq, rs3, kb_, kt * 64, 8, p.d, sK + buf * 64 * AB_ROWB, wave, lane, 4, p.T);
        ab_stage(rq, rs3, vb, kt * 64, 8, p.d, sV + buf * 64 * AB_ROWB, wave, lane, 4, p.T);
    };
    stage_k(0, 0);
    f32x16 o[DT];  // O^T tiles: registers = output channel j, lane & 31 = query
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;  // running max (log2 domain) / this lane's share of the running sum
    const int nkt = (p.T + 63) / 64;
    // one key tile; RAGGED (compile-time) = the last tile of a sequence that is not a multiple of 64: only that instance carries the
    // per-element key mask (as a run-time `if` inside one body the compiler turned it into a compare + select per element in EVERY
    // tile: 10 VALU instructions per score element instead of 4.5, profiles/r02_c_attention_pmc.txt)
    auto tile = [&](const int kt, auto ragged_c) {
        constexpr bool RAGGED = decltype(ragged_c)::value;
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) stage_k(kt + 1, buf ^ 1);
        const char* tK = sK + buf * 64 * AB_ROWB;
        const char* tV = sV + buf * 64 * AB_ROWB;
        f32x16 s[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[mt][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < DS; ++ks) {
            const half8 fq = ab_row_frag(sQ, wave * 32, ks, lane);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) s[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ab_row_frag(tK, mt * 32, ks, lane), fq, s[mt], 0, 0, 0);
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
                mloc = fmaxf(mloc, s[mt][r]);  // raw q.k: the 1/sqrt(d) * log2(e) factor is positive, so the maximum is taken before it ...
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc * p.scale_log2);   // finite: every tile has at least one valid key
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // raw v_exp_f32: arguments are <= 0; 0 on the first tile
        m_run = m_new;
        float lsum = 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[mt][r], p.scale_log2, -m_new));  // ... and applied inside the exponent's fma
                s[mt][r] = e;
                lsum += e;
            }
        l_run = l_run * alpha + lsum;
        if (__any(alpha != 1.0f)) {  // wave-uniform skip: once the running maxima have settled nothing needs rescaling
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
        // ---- O^T += V^T P^T: A = V^T fragment (transposed read of the row-major V tile, permuted key order), B = P^T from registers ----
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const half8 fp = ab_acc_frag(s[mt], kb);
#pragma unroll
                for (int t = 0; t < DT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ab_tr_frag(tV, mt * 32 + 16 * kb, t * 32, lane), fp, o[t], 0, 0, 0);
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
        half_t* op = reinterpret_cast<half_t*>(p.out) + ((long long)n * p.T + q) * p.C + h * p.d;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int j0 = t * 32 + 8 * g4 + 4 * lh;  // registers 4*g4 .. 4*g4+3 = 4 consecutive output channels
                if (j0 + 3 < p.d) {
                    half4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (half_t)(o[t][4 * g4 + e] * inv);
                    *reinterpret_cast<half4*>(op + j0) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (j0 + e < p.d) op[j0 + e] = (half_t)(o[t][4 * g4 + e] * inv);
                }
            }
    }
}

int eod_attention_fwd_nat_x3(const float* qkv, float* out, float* lse, int N, int T, int C, int heads, int d, int q_off, int k_off, int v_off,
                             int head_stride, const float* qkv_bound, int out_presplit, int in_presplit, hipStream_t st);  // csrc/attn_x3.hip
int eod_attention_fwd_nat_f32(const float* qkv, float* out, float* lse, int N, int T, int C, int heads, int d, int q_off, int k_off, int v_off,
                              int head_stride, hipStream_t st);  // csrc/attn_f32.hip

int eod_attention_fwd_wide(const void* qkv, void* out, float* lse, int dtype, int N, int T, int C, int heads, int d, int q_off, int k_off,
                           int v_off, int head_stride, const float* qkv_bound, hipStream_t st);  // csrc/attn_wide.hip

extern "C" int eod_attention_fwd_nat(const void* qkv, void* out, float* lse, int dtype, int N, int T, int C, int heads, int d, int q_off,
                                     int k_off, int v_off, int head_stride, const float* qkv_bound, int flags, void* stream) {
    const int out_presplit = (flags & EOD_ATTN_OUT_PRESPLIT) != 0, in_presplit = (flags & EOD_ATTN_IN_PRESPLIT) != 0;
    EOD_REQUIRE(qkv && out && N > 0 && T > 0 && heads > 0 && d > 0 && C == heads * d, "attention_fwd_nat: bad args");
    EOD_REQUIRE(dtype == EOD_F16 || dtype == EOD_F32, "attention_fwd_nat: bad dtype %d", dtype);
    EOD_REQUIRE(d % 8 == 0 && d <= 512, "attention_fwd_nat: the head dim must be a multiple of 8 and <= 512");
    if (d > 64) {  // wide heads: the head dim split over the waves of a workgroup (csrc/attn_wide.hip); fp16 and split-fp16 products
        EOD_REQUIRE(!(dtype == EOD_F32 && (flags & EOD_ATTN_EXACT_F32)), "attention_fwd_nat: the exact fp32 product exists for head dims <= 64");
        EOD_REQUIRE(!out_presplit && !in_presplit, "attention_fwd_nat: pre-split tensors exist for head dims <= 64");
        EOD_REQUIRE((long long)T * 3 * C * eod_esize(dtype) < 0x7fffffffLL * 4LL, "attention_fwd_nat: sequence too long");
        return eod_attention_fwd_wide(qkv, out, lse, dtype, N, T, C, heads, d, q_off, k_off, v_off, head_stride, qkv_bound, (hipStream_t)stream);
    }
    if (dtype == EOD_F32 && (flags & EOD_ATTN_EXACT_F32)) {  // exact fp32 mode: IEEE fp32 products on v_mfma_f32_32x32x2_f32 (csrc/attn_f32.hip)
        EOD_REQUIRE(!out_presplit && !in_presplit, "attention_fwd_nat: pre-split tensors belong to the split-fp16 product, not to the exact fp32 one");
        return eod_attention_fwd_nat_f32((const float*)qkv, (float*)out, lse, N, T, C, heads, d, q_off, k_off, v_off, head_stride, (hipStream_t)stream);
    }
    if (dtype == EOD_F32) {  // fp32 storage: fp32-grade products as three fp16 MFMAs on split operands (csrc/attn_x3.hip)
        EOD_REQUIRE((long long)T * 3 * C * 4 < 0x7fffffffLL * 4LL, "attention_fwd_nat: sequence too long");
        return eod_attention_fwd_nat_x3((const float*)qkv, (float*)out, lse, N, T, C, heads, d, q_off, k_off, v_off, head_stride, qkv_bound,
                                        out_presplit, in_presplit, (hipStream_t)stream);
    }
    EOD_REQUIRE(!out_presplit && !in_presplit, "attention_fwd_nat: pre-split tensors exist for fp32 storage only");
    EOD_REQUIRE(q_off % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && head_stride % 8 == 0 && eod_aligned16(qkv) && ((uintptr_t)out & 7) == 0,
                "attention_fwd_nat: alignment of the head slices");
    EOD_REQUIRE((long long)T * 3 * C * 2 < 0x7fffffffLL, "attention_fwd_nat: one image of qkv exceeds the 2 GiB window");
    AttnFwdP p;
    p.qkv = (const char*)qkv; p.out = (char*)out; p.lse = lse;
    p.N = N; p.T = T; p.C = C; p.heads = heads; p.d = d; p.q_off = q_off; p.k_off = k_off; p.v_off = v_off; p.hs = head_stride;
    p.scale_log2 = 1.4426950408889634f / sqrtf((float)d);
    const dim3 grid((T + 127) / 128, N * heads);
    const size_t lds = (size_t)(128 + 256) * AB_ROWB;
    hipStream_t st = (hipStream_t)stream;
    const int ds = (d + 15) / 16;
    if (ds == 1) hipLaunchKernelGGL((attn_fwd_nat_kernel<1, 1>), grid, dim3(256), lds, st, p);
    else if (ds == 2) hipLaunchKernelGGL((attn_fwd_nat_kernel<2, 1>), grid, dim3(256), lds, st, p);
    else if (ds == 3) hipLaunchKernelGGL((attn_fwd_nat_kernel<3, 2>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((attn_fwd_nat_kernel<4, 2>), grid, dim3(256), lds, st, p);
    EOD_CHECK_LAUNCH("attention_fwd_nat");
    return EOD_OK;
}

extern "C" int eod_attention_bwd(const void* qkv, const void* dO, const float* lse, const float* D, void* dqkv, int dtype, int N, int T, int C,
                                 int heads, int d, int q_off, int k_off, int v_off, int head_stride, void* stream) {
    EOD_REQUIRE(qkv && dO && lse && D && dqkv && N > 0 && T > 0 && heads > 0 && d > 0 && C == heads * d, "attention_bwd: bad args");
    EOD_REQUIRE(dtype == EOD_F16, "attention_bwd: fp16 only");
    EOD_REQUIRE(d % 8 == 0 && d <= 64, "attention_bwd: needs a head dim that is a multiple of 8 and <= 64");
    EOD_REQUIRE(q_off % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && head_stride % 8 == 0 && eod_aligned16(qkv) && eod_aligned16(dO),
                "attention_bwd: 16-byte alignment of the head slices");
    EOD_REQUIRE((long long)T * 3 * C * 2 < 0x7fffffffLL, "attention_bwd: one image of qkv exceeds the 2 GiB window");
    AttnBwdP p;
    p.qkv = (const char*)qkv; p.dO = (const char*)dO; p.lse = lse; p.D = D; p.dqkv = (char*)dqkv;
    p.N = N; p.T = T; p.C = C; p.heads = heads; p.d = d; p.q_off = q_off; p.k_off = k_off; p.v_off = v_off; p.hs = head_stride;
    p.alpha = 1.0f / sqrtf((float)d);
    {
        int l2 = 0;
        while ((2 << l2) <= T && l2 < 12) ++l2;  // min(12, floor(log2 T))
        p.ds_log2 = (float)l2;
    }
    const dim3 grid((T + 127) / 128, N * heads);
    const size_t lds_kv = (size_t)(256 + 256) * AB_ROWB + 4 * 64 * sizeof(float);
    const size_t lds_q = (size_t)(256 + 256) * AB_ROWB;
    hipStream_t st = (hipStream_t)stream;
    const int ds = (d + 15) / 16, dt = (d + 31) / 32;
#define EOD_AB_LAUNCH(DS_, DT_)                                                                                                   \
    do {                                                                                                                          \
        static bool attr = false;                                                                                                 \
        if (!attr) {                                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kv_kernel<DS_, DT_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv); \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_q_kernel<DS_, DT_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);   \
            attr = true;                                                                                                          \
        }                                                                                                                         \
        hipLaunchKernelGGL((attn_bwd_kv_kernel<DS_, DT_>), grid, dim3(256), lds_kv, st, p);                                       \
        hipLaunchKernelGGL((attn_bwd_q_kernel<DS_, DT_>), grid, dim3(256), lds_q, st, p);                                         \
    } while (0)
    if (ds == 1) EOD_AB_LAUNCH(1, 1);
    else if (ds == 2) EOD_AB_LAUNCH(2, 1);
    else if (ds == 3) EOD_AB_LAUNCH(3, 2);
    else EOD_AB_LAUNCH(4, 2);
#undef EOD_AB_LAUNCH
    EOD_CHECK_LAUNCH("attention_bwd");
    return EOD_OK;
}
