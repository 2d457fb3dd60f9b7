"""Shared test helpers (CPU side)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gload(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def gt(name):
    """fixture as torch tensors"""
    return {k: torch.from_numpy(np.asarray(v)) for k, v in gload(name).items()}


def unet_cfgs():
    with open(os.path.join(GOLDEN, "unet_cfgs.json")) as f:
        return json.load(f)


def key_contracts():
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        return json.load(f)


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def bits_equal(a, b):
    a = a.contiguous()
    b = b.contiguous()
    return a.shape == b.shape and a.dtype == b.dtype and bool((a.view(torch.int32) == b.view(torch.int32)).all())


def close_ulp(a, b, rel=4e-7):
    """|a-b| <= rel * max(|b|, tiny) elementwise scaled by the tensor's magnitude: a few fp32 ulps.
    Used where the reference's own bits are machine-dependent (torch's AVX512 CPU sqrt is not correctly
    rounded, see oracle/sampler_ref.py:_sqrt)."""
    a = a.double()
    b = b.double()
    scale = b.abs().max().clamp_min(1e-30)
    return bool(((a - b).abs() <= rel * torch.maximum(b.abs(), 1e-3 * scale) * 4).all())


def eo_tables_as_ldm(tb):
    """The coefficient tables of ddpm.py's p_sample (register_schedule names) filled with what model.py:126-150 computes per step
    from ITS fp32 buffers, in model.py's own fp32 op order: with these tables `DDPM.p_sample` and
    `EODiffusion._reverse_diffusion_with_clip` are the same function up to sqrt(var) vs exp(0.5 log var)."""
    betas, alphas, acp = (tb[k].float() for k in ("betas", "alphas", "alphas_cumprod"))
    acp_prev = torch.cat([torch.ones(1), acp[:-1]])  # t = 0: model.py's branch mean = beta_0 / (1 - acp_0) * x0, no noise
    var = betas * (1.0 - acp_prev) / (1.0 - acp)
    return {
        "betas": betas, "alphas_cumprod": acp, "alphas_cumprod_prev": acp_prev,
        "sqrt_alphas_cumprod": tb["sqrt_alphas_cumprod"].float(),
        "sqrt_one_minus_alphas_cumprod": tb["sqrt_one_minus_alphas_cumprod"].float(),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - acp),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / acp), "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / acp - 1.0),
        "posterior_variance": var, "posterior_log_variance_clipped": torch.log(torch.clamp(var, min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(acp_prev) / (1.0 - acp),
        "posterior_mean_coef2": (1.0 - acp_prev) * torch.sqrt(alphas) / (1.0 - acp),
    }
