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
