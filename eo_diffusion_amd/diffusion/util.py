"""Schedule helpers with the reference's names (diffusion/util.py:38-116, 281-284).

All of these are init-time, host-side table builders (numpy float64 / torch fp32 exactly as the
reference types them); the per-step arithmetic that consumes the tables lives in libeodiff.so.
"""
import numpy as np
import torch


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """float64 betas as a numpy array (util.py:38-60)."""
    f64 = torch.float64
    if schedule == "linear":
        b = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=f64) ** 2
    elif schedule == "cosine":
        grid = torch.arange(n_timestep + 1, dtype=f64) / n_timestep + cosine_s
        abar = torch.cos(grid / (1 + cosine_s) * np.pi / 2).pow(2)
        abar = abar / abar[0]
        b = torch.clamp(1 - abar[1:] / abar[:-1], min=0, max=0.999)
    elif schedule == "sqrt_linear":
        b = torch.linspace(linear_start, linear_end, n_timestep, dtype=f64)
    elif schedule == "sqrt":
        b = torch.linspace(linear_start, linear_end, n_timestep, dtype=f64) ** 0.5
    else:
        raise ValueError(f"schedule '{schedule}' unknown.")
    return b.numpy()


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """Integer DDIM sub-sequence, shifted by +1 (util.py:63-77).  Bit-exact integer contract."""
    if ddim_discr_method == "uniform":
        stride = num_ddpm_timesteps // num_ddim_timesteps
        base = np.arange(0, num_ddpm_timesteps, stride)
    elif ddim_discr_method == "quad":
        base = (np.linspace(0, np.sqrt(num_ddpm_timesteps * 0.8), num_ddim_timesteps) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps_out = np.asarray(base) + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """(sigmas, alphas, alphas_prev) with the reference's dtypes (util.py:80-91): alphas is an fp32
    tensor slice, alphas_prev a float64 ndarray, sigmas their mixed-type product."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, this results in the following sigma_t schedule "
              f"for ddim sampler {sigmas}")
    return sigmas, alphas, alphas_prev


def extract_into_tensor(a, t, x_shape):
    """a[t] broadcast to x_shape's rank (util.py:113-116); index plumbing only."""
    b = t.shape[0]
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


def noise_like(shape, device, repeat=False):
    """util.py:281-284"""
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)


# ------------------------------------------------------------------------------------------------
# The rest of the reference's diffusion/util.py: latent-diffusion helpers nothing on the EODiffusion path calls (util.py:20-36, 94-279).
# They are here so that `from diffusion.util import X` keeps working for every X the reference defines: the layer helpers are the ones
# of backbones/unet_openai.py (same classes: parameter containers of the HIP path), the others small host-side functions.
# ------------------------------------------------------------------------------------------------
from ..backbones.unet_openai import (CheckpointFunction, GroupNorm32, avg_pool_nd, checkpoint, conv_nd, linear,  # noqa: E402,F401
                                     normalization, zero_module)
from ..backbones.unet_openai import timestep_embedding as _timestep_embedding  # noqa: E402

SiLU = torch.nn.SiLU  # (util.py:226-228 spells x * sigmoid(x) as a module)


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """util.py:168-188: the sinusoidal table of the UNet (HIP kernel), or with repeat_only the timestep copied into every column"""
    if repeat_only:
        return timesteps[:, None].repeat(1, dim)
    return _timestep_embedding(timesteps, dim, max_period)


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    """util.py:94-110: beta_i = min(1 - alpha_bar((i+1)/N) / alpha_bar(i/N), max_beta) as a float64 numpy array"""
    n = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)])


def scale_module(module, scale):
    """util.py:200-206: multiply every parameter in place, return the module"""
    for p in module.parameters():
        p.detach().mul_(scale)
    return module


def mean_flat(tensor):
    """util.py:209-213: mean over every dimension but the first"""
    return tensor.mean(dim=list(range(1, len(tensor.shape))))


def get_obj_from_str(string, reload=False):
    """util.py:30-35: "pkg.mod.Name" -> the object"""
    import importlib
    module, name = string.rsplit(".", 1)
    mod = importlib.import_module(module)
    if reload:
        mod = importlib.reload(mod)
    return getattr(mod, name)


def instantiate_from_config(config):
    """util.py:20-27: {"target": "pkg.mod.Class", "params": {...}} -> Class(**params); the two LDM marker strings give None"""
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


class HybridConditioner(torch.nn.Module):
    """util.py:268-278: {"c_concat": [encoder(c_concat)], "c_crossattn": [encoder(c_crossattn)]} from two configured encoders"""

    def __init__(self, c_concat_config, c_crossattn_config):
        super().__init__()
        self.concat_conditioner = instantiate_from_config(c_concat_config)
        self.crossattn_conditioner = instantiate_from_config(c_crossattn_config)

    def forward(self, c_concat, c_crossattn):
        return {"c_concat": [self.concat_conditioner(c_concat)], "c_crossattn": [self.crossattn_conditioner(c_crossattn)]}
