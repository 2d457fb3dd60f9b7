"""Oracle: DDPM / DDIM sampler algebra on the CPU (test infrastructure -- see oracle/__init__.py).

Each function replays the exact fp32 torch op sequence of the reference (with IEEE correctly-rounded
sqrt, see _sqrt) so that the fused HIP sampler kernels (built with -ffp-contract=off) can be compared
BIT-EXACTLY:

  q_sample                   diffusion/model.py:94-98   (== ddpm.py:279-282, util.py:113-116)
  ddpm_step_clip             diffusion/model.py:126-150 (given eps_hat instead of calling the UNet)
  ddpm_step_noclip           diffusion/model.py:101-122
  repaint_mix                diffusion/model.py:58-60
  ddpm_sampling              diffusion/model.py:46-75   (loop, RePaint mix, shared eps)
  ddim_step                  diffusion/ddim.py:186-206
  ddim_sampling              diffusion/ddim.py:114-164  (mask mix :145-148 with the harness-side
                             q_sample noise, SURVEY.md section 8c adaptation 3)
  ldm_p_sample               diffusion/ddpm.py:221-255  (table-driven DDPM step, spec only)
"""
import numpy as np
import torch


def _sqrt(x):
    """IEEE correctly-rounded fp32 sqrt.  torch's AVX512 CPU sqrt is NOT correctly rounded and its
    result depends on the CPU model (measured: 7/1000 entries of sqrt(alphas_cumprod) off by 1 ulp on the
    build container, 216/1000 on the EPYC 9575F GPU host; numpy's float32 sqrt and the GPU are exact), so
    the reference's own bits are machine-dependent here.  The oracle pins the IEEE result."""
    if not torch.is_tensor(x):
        x = torch.tensor(x, dtype=torch.float32)
    return torch.from_numpy(np.sqrt(x.detach().cpu().numpy().astype(np.float32, copy=False))).reshape(x.shape)


def _g(table, t, n):
    return table.gather(-1, t).reshape(n, 1, 1, 1)


def q_sample(tb, x0, t, noise):
    assert x0.shape == noise.shape
    n = x0.shape[0]
    return _g(tb["sqrt_alphas_cumprod"], t, n) * x0 + _g(tb["sqrt_one_minus_alphas_cumprod"], t, n) * noise


def repaint_mix(tb, x_t, gt, mask, t, noise):
    gt_noised = q_sample(tb, gt, t, noise)
    return mask * gt_noised + (1 - mask) * x_t


def ddpm_step_clip(tb, x_t, t, noise, pred):
    n = x_t.shape[0]
    alpha_t = _g(tb["alphas"], t, n)
    acp = _g(tb["alphas_cumprod"], t, n)
    beta_t = _g(tb["betas"], t, n)
    x0 = _sqrt(1.0 / acp) * x_t - _sqrt(1.0 / acp - 1.0) * pred
    x0 = x0.clamp(-1.0, 1.0)
    if t.min() > 0:
        acp_prev = _g(tb["alphas_cumprod"], t - 1, n)
        mean = (beta_t * _sqrt(acp_prev) / (1.0 - acp)) * x0 + (
            (1.0 - acp_prev) * _sqrt(alpha_t) / (1.0 - acp)
        ) * x_t
        std = _sqrt(beta_t * (1.0 - acp_prev) / (1.0 - acp))
    else:
        mean = (beta_t / (1.0 - acp)) * x0
        std = 0.0
    return mean + std * noise


def ddpm_step_noclip(tb, x_t, t, noise, pred):
    n = x_t.shape[0]
    alpha_t = _g(tb["alphas"], t, n)
    acp = _g(tb["alphas_cumprod"], t, n)
    beta_t = _g(tb["betas"], t, n)
    s1m = _g(tb["sqrt_one_minus_alphas_cumprod"], t, n)
    mean = (1.0 / _sqrt(alpha_t)) * (x_t - ((1.0 - alpha_t) / s1m) * pred)
    if t.min() > 0:
        acp_prev = _g(tb["alphas_cumprod"], t - 1, n)
        std = _sqrt(beta_t * (1.0 - acp_prev) / (1.0 - acp))
    else:
        std = 0.0
    return mean + std * noise


def ddpm_sampling(tb, eps_fn, x_T, noises, timesteps, clip=True, gt=None, mask=None, record=None):
    """model.py:46-75 with the RNG made explicit: `noises[k]` is the tensor the reference draws
    at loop iteration k (t = T-1-k) and uses for BOTH the RePaint q_sample and the reverse step."""
    x_t = x_T
    n = x_t.shape[0]
    for k, i in enumerate(range(timesteps - 1, -1, -1)):
        noise = noises[k]
        t = torch.full((n,), i, dtype=torch.int64)
        if gt is not None:
            x_t = repaint_mix(tb, x_t, gt, mask, t, noise)
        pred = eps_fn(x_t, t)
        x_t = (ddpm_step_clip if clip else ddpm_step_noclip)(tb, x_t, t, noise, pred)
        if record is not None:
            record.append(x_t)
    return x_t


def ddim_step(x, e_t, a_t, a_prev, sigma_t, sqrt_1m_at, noise, temperature=1.0):
    """ddim.py:192-206.  The four scalars arrive as python floats / 0-d values and are rounded to
    fp32 by torch.full exactly as in the reference."""
    b = x.shape[0]
    a_t = torch.full((b, 1, 1, 1), float(a_t))
    a_prev = torch.full((b, 1, 1, 1), float(a_prev))
    sigma_t = torch.full((b, 1, 1, 1), float(sigma_t))
    sqrt_1m_at = torch.full((b, 1, 1, 1), float(sqrt_1m_at))
    pred_x0 = (x - sqrt_1m_at * e_t) / _sqrt(a_t)
    dir_xt = _sqrt(1.0 - a_prev - sigma_t ** 2) * e_t
    nz = sigma_t * noise * temperature
    x_prev = _sqrt(a_prev) * pred_x0 + dir_xt + nz
    return x_prev, pred_x0


def ddim_sampling(tb, dd, steps, eps_fn, x_T, step_noises, x0=None, mask=None, mix_noises=None, temperature=1.0):
    """ddim.py:114-164.  `steps` = int64 ddim timesteps (ascending), dd = oracle.schedule.ddim_tables.
    step_noises[i] is the tensor drawn at ddim.py:203; mix_noises[i] the harness-supplied q_sample
    noise of the masked branch (ddim.py:147 omits it -- SURVEY.md 8c adaptation 3)."""
    img = x_T
    b = img.shape[0]
    total = len(steps)
    pred_x0 = img
    for i, step in enumerate(np.flip(steps)):
        index = total - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        if mask is not None:
            assert x0 is not None
            img_orig = q_sample(tb, x0, ts, mix_noises[i])
            img = img_orig * mask + (1.0 - mask) * img
        e_t = eps_fn(img, ts)
        img, pred_x0 = ddim_step(
            img, e_t, dd["a"][index], dd["a_prev"][index], dd["sigma"][index], dd["sqrt_1m_a"][index],
            step_noises[i], temperature,
        )
    return img, pred_x0


def ldm_p_sample(lt, x, t, eps_hat, noise, clip_denoised=True):
    """ddpm.py:221-255: predict_start_from_noise, clamp, q_posterior, nonzero-mask noise."""
    b = x.shape[0]
    ex = lambda a: a.gather(-1, t).reshape(b, 1, 1, 1)
    x_recon = ex(lt["sqrt_recip_alphas_cumprod"]) * x - ex(lt["sqrt_recipm1_alphas_cumprod"]) * eps_hat
    if clip_denoised:
        x_recon = x_recon.clamp(-1.0, 1.0)
    mean = ex(lt["posterior_mean_coef1"]) * x_recon + ex(lt["posterior_mean_coef2"]) * x
    logvar = ex(lt["posterior_log_variance_clipped"])
    nonzero = (1 - (t == 0).float()).reshape(b, 1, 1, 1)
    return mean + nonzero * (0.5 * logvar).exp() * noise
