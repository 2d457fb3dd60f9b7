// Fused DDPM / DDIM sampler updates.  HBM-bound elementwise kernels, one pass
// over x_t, 16-byte accesses where the shape allows.
//
// THIS FILE IS COMPILED WITH -ffp-contract=off: every multiply/add/div/sqrt below is a separately
// rounded IEEE fp32 operation in exactly the order of the reference's torch expression, so the
// results are bit-identical to the torch CPU path (tests/test_gpu_sampler.py checks bits).
// hipcc's default correctly-rounded fp32 divide and sqrt are relied upon (no fast-math here).
//
//   q_sample      diffusion/model.py:94-98
//   repaint_mix   diffusion/model.py:58-60
//   ddpm_step     diffusion/model.py:101-122 (clip=0), :126-150 (clip=1)
//   ddim_step     diffusion/ddim.py:192-206
#include "common.h"

// Caller-supplied timesteps index the schedule tables: an index outside [0, T) would read behind them (the reference's gather
// raises there).  Such a sample is computed with index 0 and its whole output is poisoned with NaN -- loud, but memory-safe.
__device__ __forceinline__ long long checked_t(long long tn, int T, bool& bad) {
    bad = tn < 0 || tn >= (long long)T;
    return bad ? 0 : tn;
}
#define EOD_POISON(bad, v) ((bad) ? __builtin_nanf("") : (v))

__device__ __forceinline__ long long tmin_of(const long long* t, int N) {
    long long m = t[0];
    for (int i = 1; i < N; ++i) m = t[i] < m ? t[i] : m;
    return m;
}

// grid (blocks, N): blockIdx.y = sample, so per-sample coefficients are computed once per thread.
__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const long long* __restrict__ t,
                                const float* __restrict__ sa, const float* __restrict__ sb, float* __restrict__ out, long long chw, int T) {
    const int n = blockIdx.y;
    bool bad;
    const long long tn = checked_t(t[n], T, bad);
    const float a = sa[tn], b = sb[tn];
    const long long base = (long long)n * chw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chw; i += (long long)gridDim.x * blockDim.x) {
        const float p = a * x0[base + i];
        const float q = b * noise[base + i];
        out[base + i] = EOD_POISON(bad, p + q);
    }
}

__global__ void repaint_mix_kernel(const float* __restrict__ x_t, const float* __restrict__ gt, const float* __restrict__ mask,
                                   const float* __restrict__ noise, const long long* __restrict__ t, const float* __restrict__ sa,
                                   const float* __restrict__ sb, float* __restrict__ out, int C, long long hw, int T) {
    const int n = blockIdx.y;
    bool bad;
    const long long tn = checked_t(t[n], T, bad);
    const float a = sa[tn], b = sb[tn];
    const long long chw = (long long)C * hw, base = (long long)n * chw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chw; i += (long long)gridDim.x * blockDim.x) {
        const float m = mask[(long long)n * hw + (i % hw)];
        const float p = a * gt[base + i];
        const float q = b * noise[base + i];
        const float gn = p + q;              // gt_noised = _forward_diffusion(gt, t, noise)
        const float l = m * gn;              // mask*gt_noised
        const float om = 1.0f - m;           // (1-mask)
        const float r = om * x_t[base + i];  // (1-mask)*x_t
        out[base + i] = EOD_POISON(bad, l + r);
    }
}

template <bool CLIP>
__global__ void ddpm_step_kernel(const float* __restrict__ x_t, const float* __restrict__ pred, const float* __restrict__ noise,
                                 const long long* __restrict__ t, const float* __restrict__ betas, const float* __restrict__ alphas,
                                 const float* __restrict__ acp, const float* __restrict__ s1m, float* __restrict__ out, int N,
                                 long long chw, int T) {
    const int n = blockIdx.y;
    bool bad;
    const long long tn = checked_t(t[n], T, bad);
    const bool all_pos = tmin_of(t, N) > 0;  // the reference branches on the BATCH minimum (model.py:113,140)
    const float alpha_t = alphas[tn], acp_t = acp[tn], beta_t = betas[tn];
    float c_x0 = 0.f, c_pred = 0.f, m_x0 = 0.f, m_xt = 0.f, std = 0.0f, k_mean = 0.f, k_pred = 0.f;
    if (CLIP) {
        c_x0 = sqrtf(1.0f / acp_t);           // torch.sqrt(1. / alpha_t_cumprod)
        c_pred = sqrtf(1.0f / acp_t - 1.0f);  // torch.sqrt(1. / alpha_t_cumprod - 1.)
        if (all_pos) {
            const float acp_prev = acp[tn - 1];
            m_x0 = beta_t * sqrtf(acp_prev) / (1.0f - acp_t);
            m_xt = (1.0f - acp_prev) * sqrtf(alpha_t) / (1.0f - acp_t);
            std = sqrtf(beta_t * (1.0f - acp_prev) / (1.0f - acp_t));
        } else {
            m_x0 = beta_t / (1.0f - acp_t);
        }
    } else {
        k_mean = 1.0f / sqrtf(alpha_t);          // (1./torch.sqrt(alpha_t))
        k_pred = (1.0f - alpha_t) / s1m[tn];     // ((1.0-alpha_t)/sqrt_one_minus_alpha_cumprod_t)
        if (all_pos) {
            const float acp_prev = acp[tn - 1];
            std = sqrtf(beta_t * (1.0f - acp_prev) / (1.0f - acp_t));
        }
    }
    const long long base = (long long)n * chw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chw; i += (long long)gridDim.x * blockDim.x) {
        const float x = x_t[base + i], e = pred[base + i], z = noise[base + i];
        float mean;
        if (CLIP) {
            const float u = c_x0 * x;
            const float v = c_pred * e;
            float x0 = u - v;
            x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
            if (all_pos) {
                const float p = m_x0 * x0;
                const float q = m_xt * x;
                mean = p + q;
            } else {
                mean = m_x0 * x0;
            }
        } else {
            const float v = k_pred * e;
            const float d = x - v;
            mean = k_mean * d;
        }
        const float sz = std * z;
        out[base + i] = EOD_POISON(bad, mean + sz);
    }
}

__global__ void ddim_step_kernel(const float* __restrict__ x, const float* __restrict__ e_t, const float* __restrict__ noise,
                                 float a_t, float a_prev, float sigma_t, float s1m_at, float temperature,
                                 float* __restrict__ x_prev, float* __restrict__ pred_x0, long long numel) {
    const float sq_at = sqrtf(a_t);
    const float sig2 = sigma_t * sigma_t;                // sigma_t**2
    const float dcoef = sqrtf((1.0f - a_prev) - sig2);   // (1. - a_prev - sigma_t**2).sqrt()
    const float sq_ap = sqrtf(a_prev);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (long long)gridDim.x * blockDim.x) {
        const float xv = x[i], e = e_t[i];
        const float se = s1m_at * e;
        const float p0 = (xv - se) / sq_at;              // pred_x0
        const float dir = dcoef * e;                     // dir_xt
        float nz = 0.0f;
        if (noise) {
            const float sn = sigma_t * noise[i];
            nz = sn * temperature;                       // sigma_t * noise * temperature
        } else {
            nz = (sigma_t * 0.0f) * temperature;
        }
        const float a = sq_ap * p0;
        const float b = a + dir;
        x_prev[i] = b + nz;
        if (pred_x0) pred_x0[i] = p0;
    }
}

// classifier-free guidance (ddim.py:177-181): e = e_uncond + scale * (e_cond - e_uncond)
__global__ void cfg_combine_kernel(const float* __restrict__ eu, const float* __restrict__ ec, float scale, float* __restrict__ out,
                                   long long numel) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (long long)gridDim.x * blockDim.x) {
        const float d = ec[i] - eu[i];
        const float m = scale * d;
        out[i] = eu[i] + m;
    }
}

// table-driven DDPM step of the LDM-derived sampler (ddpm.py:221-255): predict_start_from_noise, clamp, q_posterior,
// noise scaled by exp(0.5 * posterior_log_variance_clipped) and masked out at t == 0 (per sample, not per batch).
template <bool CLIP>
__global__ void ldm_p_sample_kernel(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ noise,
                                    const long long* __restrict__ t, const float* __restrict__ sr, const float* __restrict__ srm1,
                                    const float* __restrict__ c1, const float* __restrict__ c2, const float* __restrict__ lv,
                                    float* __restrict__ out, long long chw, int T) {
    const int n = blockIdx.y;
    bool bad;
    const long long tn = checked_t(t[n], T, bad);
    const float a = sr[tn], b = srm1[tn], k1 = c1[tn], k2 = c2[tn];
    const float nonzero = 1.0f - (tn == 0 ? 1.0f : 0.0f);
    const float sd = expf(0.5f * lv[tn]);
    const float ns = nonzero * sd;
    const long long base = (long long)n * chw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chw; i += (long long)gridDim.x * blockDim.x) {
        const float xv = x[base + i];
        const float u = a * xv;
        const float v = b * eps[base + i];
        float x0 = u - v;
        if (CLIP) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        const float p = k1 * x0;
        const float q = k2 * xv;
        const float mean = p + q;
        const float z = ns * noise[base + i];
        out[base + i] = EOD_POISON(bad, mean + z);
    }
}

static inline unsigned blocks_for(long long n, int cap) {
    long long b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

extern "C" int eod_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_acp, const float* sqrt_1m_acp,
                            float* out, int N, int64_t chw, int T, void* stream) {
    EOD_REQUIRE(x0 && noise && t && sqrt_acp && sqrt_1m_acp && out && N > 0 && chw > 0 && T > 0, "q_sample: bad args");
    hipLaunchKernelGGL(q_sample_kernel, dim3(blocks_for(chw, 512), N), dim3(256), 0, (hipStream_t)stream, x0, noise, (const long long*)t, sqrt_acp, sqrt_1m_acp, out, (long long)chw, T);
    EOD_CHECK_LAUNCH("q_sample");
    return EOD_OK;
}

extern "C" int eod_repaint_mix(const float* x_t, const float* gt, const float* mask, const float* noise, const int64_t* t,
                               const float* sqrt_acp, const float* sqrt_1m_acp, float* out, int N, int C, int64_t hw, int T,
                               void* stream) {
    EOD_REQUIRE(x_t && gt && mask && noise && t && sqrt_acp && sqrt_1m_acp && out && N > 0 && C > 0 && hw > 0 && T > 0, "repaint_mix: bad args");
    hipLaunchKernelGGL(repaint_mix_kernel, dim3(blocks_for((long long)C * hw, 512), N), dim3(256), 0, (hipStream_t)stream, x_t, gt, mask, noise, (const long long*)t, sqrt_acp, sqrt_1m_acp, out, C, (long long)hw, T);
    EOD_CHECK_LAUNCH("repaint_mix");
    return EOD_OK;
}

extern "C" int eod_ddpm_step(const float* x_t, const float* pred, const float* noise, const int64_t* t, const float* betas,
                             const float* alphas, const float* acp, const float* sqrt_1m_acp, float* out, int N, int64_t chw, int T,
                             int clip, void* stream) {
    EOD_REQUIRE(x_t && pred && noise && t && betas && alphas && acp && sqrt_1m_acp && out && N > 0 && chw > 0 && T > 0, "ddpm_step: bad args");
    dim3 grid(blocks_for(chw, 512), N);
    if (clip)
        hipLaunchKernelGGL(ddpm_step_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x_t, pred, noise, (const long long*)t, betas, alphas, acp, sqrt_1m_acp, out, N, (long long)chw, T);
    else
        hipLaunchKernelGGL(ddpm_step_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x_t, pred, noise, (const long long*)t, betas, alphas, acp, sqrt_1m_acp, out, N, (long long)chw, T);
    EOD_CHECK_LAUNCH("ddpm_step");
    return EOD_OK;
}

extern "C" int eod_ddim_step(const float* x, const float* e_t, const float* noise, float a_t, float a_prev, float sigma_t,
                             float sqrt_1m_at, float temperature, float* x_prev, float* pred_x0, int64_t numel, void* stream) {
    EOD_REQUIRE(x && e_t && x_prev && numel > 0, "ddim_step: bad args");
    hipLaunchKernelGGL(ddim_step_kernel, dim3(blocks_for(numel, 4096)), dim3(256), 0, (hipStream_t)stream, x, e_t, noise, a_t, a_prev, sigma_t, sqrt_1m_at, temperature, x_prev, pred_x0, (long long)numel);
    EOD_CHECK_LAUNCH("ddim_step");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// k15: Philox4x32-10 + Box-Muller.  counter = (element_index/4, global sample index, step, stream_id),
// key = seed.  Each thread produces 4 normals (one 16-byte store).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__device__ __forceinline__ float u01(unsigned u) { return ((float)(u >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__global__ void randn_philox_kernel(float* __restrict__ out, long long chw, unsigned long long seed, long long sample0, int step,
                                    int stream_id) {
    const int n = blockIdx.y;
    const long long quads = (chw + 3) / 4;
    float* o = out + (long long)n * chw;
    for (long long qd = (long long)blockIdx.x * blockDim.x + threadIdx.x; qd < quads; qd += (long long)gridDim.x * blockDim.x) {
        unsigned c0 = (unsigned)qd, c1 = (unsigned)(sample0 + n), c2 = (unsigned)step, c3 = (unsigned)stream_id ^ (unsigned)(qd >> 32);
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c0, c1, c2, c3, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const float r0 = sqrtf(-2.0f * logf(u01(c0))), r1 = sqrtf(-2.0f * logf(u01(c2)));
        const float a0 = 6.28318530717958647692f * u01(c1), a1 = 6.28318530717958647692f * u01(c3);
        const float z[4] = {r0 * cosf(a0), r0 * sinf(a0), r1 * cosf(a1), r1 * sinf(a1)};
        const long long e = qd * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (e + j < chw) o[e + j] = z[j];
    }
}

extern "C" int eod_randn_philox(float* out, int N, int64_t chw, uint64_t seed, int64_t sample0, int32_t step, int32_t stream_id,
                                void* stream) {
    EOD_REQUIRE(out && N > 0 && chw > 0, "randn_philox: bad args");
    hipLaunchKernelGGL(randn_philox_kernel, dim3(blocks_for((chw + 3) / 4, 512), N), dim3(256), 0, (hipStream_t)stream, out, (long long)chw, (unsigned long long)seed, (long long)sample0, step, stream_id);
    EOD_CHECK_LAUNCH("randn_philox");
    return EOD_OK;
}

extern "C" int eod_cfg_combine(const float* e_uncond, const float* e_cond, float scale, float* out, int64_t numel, void* stream) {
    EOD_REQUIRE(e_uncond && e_cond && out && numel > 0, "cfg_combine: bad args");
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(blocks_for(numel, 4096)), dim3(256), 0, (hipStream_t)stream, e_uncond, e_cond, scale, out, (long long)numel);
    EOD_CHECK_LAUNCH("cfg_combine");
    return EOD_OK;
}

extern "C" int eod_ldm_p_sample(const float* x, const float* eps, const float* noise, const int64_t* t, const float* sqrt_recip_acp,
                                const float* sqrt_recipm1_acp, const float* post_coef1, const float* post_coef2,
                                const float* post_logvar, float* out, int N, int64_t chw, int T, int clip, void* stream) {
    EOD_REQUIRE(x && eps && noise && t && sqrt_recip_acp && sqrt_recipm1_acp && post_coef1 && post_coef2 && post_logvar && out && N > 0 && chw > 0 && T > 0,
                "ldm_p_sample: bad args");
    dim3 grid(blocks_for(chw, 512), N);
    if (clip)
        hipLaunchKernelGGL(ldm_p_sample_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, eps, noise, (const long long*)t, sqrt_recip_acp, sqrt_recipm1_acp, post_coef1, post_coef2, post_logvar, out, (long long)chw, T);
    else
        hipLaunchKernelGGL(ldm_p_sample_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, eps, noise, (const long long*)t, sqrt_recip_acp, sqrt_recipm1_acp, post_coef1, post_coef2, post_logvar, out, (long long)chw, T);
    EOD_CHECK_LAUNCH("ldm_p_sample");
    return EOD_OK;
}


// ---------------------------------------------------------------------------------------------
// Harness-side elementwise ops of inference.py, same bit-exact convention as above:
//   repaint_cond   inference.py:100-109   cond = cat(image, 1 - mask)   (cond_type == "sum": mask 1 = keep after the inversion)
//   postprocess    inference.py:128       samples.clip(0, 1)  |  (samples + 1) / 2
//   masked_preview inference.py:134       image * (mask + 0.7).clip(0, 1)
// ---------------------------------------------------------------------------------------------
__global__ void repaint_cond_kernel(const float* __restrict__ image, const float* __restrict__ mask, float* __restrict__ cond, int C,
                                    long long hw, int invert) {
    const int n = blockIdx.y;
    const long long per_in = (long long)C * hw, per_out = (long long)(C + 1) * hw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_out; i += (long long)gridDim.x * blockDim.x) {
        float v;
        if (i < per_in) {
            v = image[(long long)n * per_in + i];
        } else {
            const float m = mask[(long long)n * hw + (i - per_in)];
            v = invert ? 1.0f - m : m;
        }
        cond[(long long)n * per_out + i] = v;
    }
}

__global__ void postprocess_kernel(const float* __restrict__ x, float* __restrict__ y, long long numel, int mode) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        float o;
        if (mode == 0) {
            o = fminf(fmaxf(v, 0.0f), 1.0f);
            if (v != v) o = v;  // torch.clip propagates NaN
        } else {
            const float a = v + 1.0f;
            o = a / 2.0f;
        }
        y[i] = o;
    }
}

__global__ void masked_preview_kernel(const float* __restrict__ image, const float* __restrict__ mask, float* __restrict__ out, int C,
                                      long long hw, float lift) {
    const int n = blockIdx.y;
    const long long per = (long long)C * hw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x) {
        const float a = mask[(long long)n * hw + (i % hw)] + lift;
        const float g = fminf(fmaxf(a, 0.0f), 1.0f);
        out[(long long)n * per + i] = image[(long long)n * per + i] * g;
    }
}

extern "C" int eod_repaint_cond(const float* image, const float* mask, float* cond, int N, int C, int64_t hw, int invert, void* stream) {
    EOD_REQUIRE(image && mask && cond && N > 0 && C > 0 && hw > 0, "repaint_cond: bad args");
    hipLaunchKernelGGL(repaint_cond_kernel, dim3(blocks_for((long long)(C + 1) * hw, 512), N), dim3(256), 0, (hipStream_t)stream, image, mask, cond, C, (long long)hw, invert);
    EOD_CHECK_LAUNCH("repaint_cond");
    return EOD_OK;
}

extern "C" int eod_postprocess(const float* x, float* y, int64_t numel, int mode, void* stream) {
    EOD_REQUIRE(x && y && numel > 0 && (mode == 0 || mode == 1), "postprocess: bad args");
    hipLaunchKernelGGL(postprocess_kernel, dim3(blocks_for(numel, 4096)), dim3(256), 0, (hipStream_t)stream, x, y, (long long)numel, mode);
    EOD_CHECK_LAUNCH("postprocess");
    return EOD_OK;
}

extern "C" int eod_masked_preview(const float* image, const float* mask, float* out, int N, int C, int64_t hw, float lift, void* stream) {
    EOD_REQUIRE(image && mask && out && N > 0 && C > 0 && hw > 0, "masked_preview: bad args");
    hipLaunchKernelGGL(masked_preview_kernel, dim3(blocks_for((long long)C * hw, 512), N), dim3(256), 0, (hipStream_t)stream, image, mask, out, C, (long long)hw, lift);
    EOD_CHECK_LAUNCH("masked_preview");
    return EOD_OK;
}
