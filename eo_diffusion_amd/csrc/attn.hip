// Fused ("flash"-style) QKV attention forward for the guided-diffusion AttentionBlock (unet_openai.py:456-515):
//     a[n, t, h*d + j] = sum_s softmax_s( (q_t . k_s) / sqrt(d) ) * v[s, j]
// The T x T weight matrix (4.3 GB in fp32 at 256x256 / batch 8 in the reference, :476-480) is never materialised:
// one workgroup = 128 queries of one (image, head), 4 waves x 32 queries, streaming 64-key tiles through LDS with an
// online (running max / running sum) softmax in fp32.
//
// MFMA formulation (v_mfma_f32_32x32x16_f16), chosen so that NO cross-lane data movement is needed between the two
// contractions and every softmax reduction is in-lane:
//   1) S^T = K . Q^T   (A = K tile rows from LDS, B = Q rows held in registers).  In the C layout the LANE is the query
//      and the 16 registers are keys, so max / sum over keys are register reductions (+ one exchange with lane^32,
//      which holds the other 16 keys of the same query).
//   2) O^T = V^T . P^T (A = V^T rows [d][keys] from LDS, B = P).  A lane's 8 consecutive S^T registers are exactly a
//      valid B fragment for a PERMUTED key order (keys 16s + 4h + {0..3} and 16s + 8 + 4h + {0..3} for lane half h), so P
//      goes from the accumulator to the B operand with only an fp32->fp16 convert; V^T is read from LDS in the same
//      permuted order (two ds_read_b64 instead of one ds_read_b128).  O^T again has the query on the lane, so the
//      online-softmax rescale is one multiply per register.
// Inputs are what the attention block's projections already produce (backbones/unet_openai.py AttentionBlock._emit):
//   qk  [N*T][2*Cq]  q heads | k heads, each head padded to dpad (multiple of 8) channels, d contiguous
//   vT  [N][C][ldt]  V transposed (keys contiguous), produced by the operand-swapped projection GEMM
// fp16 storage, head dim <= 64.  Other shapes (fp32 parity mode, d = 128 / 512) use the materialised GEMM path.
#include "common.h"

struct AttnP {
    const half_t* qk;
    const half_t* vT;
    half_t* out;
    float* lse;
    long long ld_qk, ldt;
    int N, T, C, heads, d, dpad, k_off;
    float scale_log2;  // log2(e) / sqrt(d)
};

constexpr int AT_BQ = 128, AT_KT = 64;
constexpr int AT_KROW = 128;        // K tile row stride (bytes): dpad <= 64 halves, 16-byte chunks XOR-swizzled
constexpr int AT_VROW = 136;        // V^T tile row stride (bytes): 64 keys + 8 B pad -> conflict-free ds_read_b64

template <int DS, int DT>  // DS = dpad/16 QK^T k-substeps (3 or 4), DT = 32-row tiles of O^T (ceil(d/32))
__global__ __launch_bounds__(256, 2) void attn_flash_kernel(const AttnP p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * (AT_KT * AT_KROW + 64 * AT_VROW)];
    constexpr int STAGE = AT_KT * AT_KROW + 64 * AT_VROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int nh = blockIdx.y, n = nh / p.heads, h = nh - n * p.heads;
    const int q0 = blockIdx.x * AT_BQ + wave * 32;

    const half_t* qbase = p.qk + (long long)n * p.T * p.ld_qk + h * p.dpad;
    const half_t* kbase = qbase + p.k_off;
    const half_t* vbase = p.vT + ((long long)n * p.C + (long long)h * p.d) * p.ldt;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane (query lr, half lh) holds Q[q][16s + 8lh .. +8] ----
    i32x4 qf[DS];
    {
        const int q = q0 + lr;
        const bool ok = q < p.T;
        const half_t* qp = qbase + (long long)(ok ? q : 0) * p.ld_qk + 8 * lh;
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            const i32x4 v = *reinterpret_cast<const i32x4*>(qp + 16 * s);
            qf[s] = (ok && (2 * s + lh) * 8 < p.dpad) ? v : i32x4{0, 0, 0, 0};  // zero beyond the (padded) head dim
        }
    }

    // ---- staging: K tile = 64 rows x (DS*2) 16-byte chunks, V^T tile = (DT*32) rows x 8 chunks.  Split into
    // stage_load (global -> registers, issued BEFORE the MFMAs of the current tile) and stage_store (registers -> LDS,
    // AFTER them): the global-load latency hides under the tile's compute (guide T14) ----
    constexpr int KCH = DS * 2;
    constexpr int NKC = (AT_KT * KCH + 255) / 256, NVC = (DT * 32 * 8 + 255) / 256;
    i32x4 rk[NKC], rv[NVC];
    auto stage_load = [&](int kt) {
        const int key0 = kt * AT_KT;
#pragma unroll
        for (int i = 0; i < NKC; ++i) {
            const int c = tid + i * 256;
            const int row = c / KCH, ch = c - row * KCH;
            const int key = key0 + row;
            const bool ok = c < AT_KT * KCH && key < p.T && ch * 8 < p.dpad;
            const i32x4 v = *reinterpret_cast<const i32x4*>(ok ? kbase + (long long)key * p.ld_qk + ch * 8 : kbase);
            rk[i] = ok ? v : i32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
            const int c = tid + i * 256;
            const int row = c >> 3, ch = c & 7;  // row = output channel j of this head, ch = 8-key chunk
            // vT rows are zero beyond T (torch.zeros at plan build); chunks beyond ldt are skipped
            const bool ok = c < DT * 32 * 8 && row < p.d && key0 + ch * 8 < p.ldt;
            const i32x4 v = *reinterpret_cast<const i32x4*>(ok ? vbase + (long long)row * p.ldt + key0 + ch * 8 : vbase);
            rv[i] = ok ? v : i32x4{0, 0, 0, 0};
        }
    };
    auto stage_store = [&](int buf) {
        char* sK = smem + buf * STAGE;
        char* sV = sK + AT_KT * AT_KROW;
#pragma unroll
        for (int i = 0; i < NKC; ++i) {
            const int c = tid + i * 256;
            const int row = c / KCH, ch = c - row * KCH;
            if (c < AT_KT * KCH) *reinterpret_cast<i32x4*>(sK + row * AT_KROW + ((ch ^ ((row >> 1) & 7)) << 4)) = rk[i];
        }
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
            const int c = tid + i * 256;
            const int row = c >> 3, ch = c & 7;
            if (c < DT * 32 * 8) {
                long long* dst = reinterpret_cast<long long*>(sV + row * AT_VROW + ch * 16);
                const long long* sv = reinterpret_cast<const long long*>(&rv[i]);
                dst[0] = sv[0];
                dst[1] = sv[1];
            }
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;  // running max (log2 domain) / this lane's share of the running sum

    const int nkt = (p.T + AT_KT - 1) / AT_KT;
    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) stage_load(kt + 1);
        const char* sK = smem + buf * STAGE;
        const char* sV = sK + AT_KT * AT_KROW;

        // ---- S^T tiles: 2 x (32 keys x 32 queries) ----
        f32x16 st[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[mt][r] = 0.0f;
            const int row = mt * 32 + lr;
            const char* kr = sK + row * AT_KROW;
            const int sw = (row >> 1) & 7;
#pragma unroll
            for (int s = 0; s < DS; ++s) {
                const i32x4 kf = *reinterpret_cast<const i32x4*>(kr + (((2 * s + lh) ^ sw) << 4));
                st[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, kf), __builtin_bit_cast(half8, qf[s]), st[mt], 0, 0, 0);
            }
        }
        // ---- online softmax (lane = query; registers = keys (i&3) + 8(i>>2) + 4lh of each 32-key tile) ----
        const int key0 = kt * AT_KT;
        float mloc = -INFINITY;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float t = st[mt][r] * p.scale_log2;
                t = key < p.T ? t : -INFINITY;
                st[mt][r] = t;
                mloc = fmaxf(mloc, t);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));       // the other 32 keys of this query live in lane^32
        const float m_new = fmaxf(m_run, mloc);         // finite: every tile has at least one valid key
        const float alpha = exp2f(m_run - m_new);       // 0 on the first tile (m_run = -inf)
        m_run = m_new;
        float lsum = 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = exp2f(st[mt][r] - m_new);
                st[mt][r] = e;
                lsum += e;
            }
        l_run = l_run * alpha + lsum;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;

        // ---- O^T += V^T . P^T : 4 k-substeps of 16 keys; B fragment = 8 consecutive S^T registers ----
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int mt = s4 >> 1, rb = (s4 & 1) * 8;
            half8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (half_t)st[mt][rb + j];
            // keys of this fragment in tile order: 16*s4 + 4lh + {0..3} and 16*s4 + 8 + 4lh + {0..3}
            const int kb = (16 * s4 + 4 * lh) * 2;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const char* vr = sV + (t * 32 + lr) * AT_VROW + kb;
                long long v2[2];
                v2[0] = *reinterpret_cast<const long long*>(vr);
                v2[1] = *reinterpret_cast<const long long*>(vr + 16);
                half8 vf;
                __builtin_memcpy(&vf, v2, 16);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[t], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) stage_store(buf ^ 1);  // the other buffer was last read in iteration kt-1 (barrier below)
        __syncthreads();  // next stage written / this stage free
    }
    // ---- normalise and store: O^T[j][q] -> out[n, q, h*d + j] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int q = q0 + lr;
    if (p.lse && lh == 0 && q < p.T)  // natural-log log-sum-exp of the (scaled) scores of this query: (m + log2 l) * ln 2
        p.lse[((long long)n * p.heads + h) * p.T + q] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
    if (q < p.T) {
        half_t* op = p.out + ((long long)n * p.T + q) * p.C + h * p.d;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int j0 = t * 32 + 8 * g4 + 4 * lh;  // 4 consecutive output channels per register group
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

extern "C" int eod_attention_fwd(const eod_attn_desc* d, void* stream) {
    EOD_REQUIRE(d && d->qk && d->vT && d->out, "attention: null pointer");
    EOD_REQUIRE(d->dtype == EOD_F16, "attention: the fused kernel is fp16-only (use the GEMM path in fp32 mode)");
    EOD_REQUIRE(d->N > 0 && d->T > 0 && d->heads > 0 && d->d > 0 && d->C == d->heads * d->d, "attention: bad dims");
    EOD_REQUIRE(d->dpad % 8 == 0 && d->dpad >= d->d && d->dpad <= 64, "attention: head dim %d (padded %d) unsupported by the fused kernel", d->d, d->dpad);
    EOD_REQUIRE(d->d % 4 == 0 && d->C % 4 == 0, "attention: head dim must be a multiple of 4");
    EOD_REQUIRE(d->ld_qk % 8 == 0 && d->ldt % 8 == 0 && d->ldt >= d->T && d->k_off % 8 == 0, "attention: leading dimensions must be multiples of 8");
    EOD_REQUIRE(eod_aligned16(d->qk) && eod_aligned16(d->vT) && ((uintptr_t)d->out & 7) == 0, "attention: alignment");
    AttnP p;
    p.qk = (const half_t*)d->qk;
    p.vT = (const half_t*)d->vT;
    p.out = (half_t*)d->out;
    p.lse = d->lse;
    p.ld_qk = d->ld_qk;
    p.ldt = d->ldt;
    p.N = d->N; p.T = d->T; p.C = d->C; p.heads = d->heads; p.d = d->d; p.dpad = d->dpad; p.k_off = d->k_off;
    p.scale_log2 = 1.4426950408889634f / sqrtf((float)d->d);
    dim3 grid((d->T + AT_BQ - 1) / AT_BQ, d->N * d->heads);
    hipStream_t st = (hipStream_t)stream;
    const int ds = (d->dpad + 15) / 16, dt = (d->d + 31) / 32;
    if (ds <= 1 && dt == 1)
        hipLaunchKernelGGL((attn_flash_kernel<1, 1>), grid, dim3(256), 0, st, p);
    else if (ds == 2 && dt == 1)
        hipLaunchKernelGGL((attn_flash_kernel<2, 1>), grid, dim3(256), 0, st, p);
    else if (ds == 3 && dt == 2)
        hipLaunchKernelGGL((attn_flash_kernel<3, 2>), grid, dim3(256), 0, st, p);
    else if (ds == 4 && dt == 2)
        hipLaunchKernelGGL((attn_flash_kernel<4, 2>), grid, dim3(256), 0, st, p);
    else {
        eod_set_error("attention: no fused variant for head dim %d (padded %d)", d->d, d->dpad);
        return EOD_ENOSYS;
    }
    EOD_CHECK_LAUNCH("attention_fwd");
    return EOD_OK;
}
