"""EODiffusion: Gaussian-diffusion wrapper (cosine schedule, q_sample, DDPM reverse steps, RePaint
`cond_type="sum"` sampling loop) with the reference's API (diffusion/model.py:12-150), running its
per-step arithmetic in fused HIP kernels (libeodiff.so: eod_q_sample, eod_repaint_mix, eod_ddpm_step,
eod_randn_philox) and the denoiser through UNetModel's native launch program.

Differences from the reference that are deliberate (and documented in DESIGN.md):
  * no `t.min() > 0` host synchronisation per step (model.py:113,140): the batch-wide branch is
    evaluated on the device inside eod_ddpm_step;
  * no per-step H2D copy of `t` (model.py:56);
  * PNG dumps happen only when `save=True` (the reference's `A and B or C and D and save` precedence
    slip at model.py:62 writes some regardless);
  * optional keyword-only extras on `sampling`: injected x_T / per-step noises (parity tests) and a
    counter-based Philox noise source keyed by the GLOBAL sample index (multi-GPU sharding).
"""
import math
import os

import torch
import torch.nn as nn

from .. import _lib
from ..backbones.unet_openai import *  # noqa: F401,F403  (the reference re-exports these, model.py:5)
from ..engine import current_stream_ptr, require_gpu

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(it, **kw):
        return it


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


class EODiffusion(nn.Module):
    def __init__(self, model, image_size, in_channels, time_embedding_dim=256, timesteps=1000, cond_type=None,
                 device="cpu"):
        super().__init__()
        self.timesteps = timesteps
        self.in_channels = in_channels
        self.image_size = image_size
        self.cond_type = cond_type
        self.device = device
        betas = self._cosine_variance_schedule(timesteps)
        alphas = 1.0 - betas
        alphas_cumprod = torch.cumprod(alphas, dim=-1)
        # state_dict contract: exactly these five fp32 buffers (model.py:28-32); checkpoints overwrite them,
        # so every sampler below reads the BUFFERS and never re-derives the schedule.
        self.register_buffer("betas", betas)
        self.register_buffer("alphas", alphas)
        self.register_buffer("alphas_cumprod", alphas_cumprod)
        self.register_buffer("sqrt_alphas_cumprod", torch.sqrt(alphas_cumprod))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", torch.sqrt(1.0 - alphas_cumprod))
        self.model = model

    # ------------------------------------------------------------------ schedule (init-time, host)
    def _cosine_variance_schedule(self, timesteps, epsilon=0.008):
        """fp32 replay of model.py:87-92.  The rounding of `1 - f[t+1]/f[t]` in fp32 is part of the
        checkpoint contract, so the op order is kept exactly."""
        s = torch.linspace(0, timesteps, steps=timesteps + 1, dtype=torch.float32)
        f = torch.cos(((s / timesteps + epsilon) / (1.0 + epsilon)) * math.pi * 0.5) ** 2
        return torch.clip(1.0 - f[1:] / f[:timesteps], 0.0, 0.999)

    # ------------------------------------------------------------------ helpers
    def _tables_on(self, dev):
        if self.betas.device != dev:
            raise _lib.EodError(f"EODiffusion buffers are on {self.betas.device}, data on {dev}: call .to(device) first")

    def _t64(self, t, dev):
        """timesteps as int64 on the device.  A host-side tensor is range-checked here for free (the reference's gather raises
        on an index outside [0, T)); for device tensors the kernels poison that sample's output with NaN instead of reading
        behind the schedule tables (no host synchronisation on the hot path)."""
        if t.device.type == "cpu" and t.numel():
            lo, hi = int(t.min()), int(t.max())
            if lo < 0 or hi >= self.timesteps:
                raise IndexError(f"timestep index out of range: got [{lo}, {hi}], schedule has {self.timesteps} steps")
        return t.to(device=dev, dtype=torch.int64).contiguous()

    # ------------------------------------------------------------------ training forward (model.py:38-44)
    def forward(self, x, noise, cond=None, y=None):
        t = torch.randint(0, self.timesteps, (x.shape[0],)).to(x.device)
        x_t = self._forward_diffusion(x, t, noise)
        return self.model(x_t, t, cond=cond, y=y)

    # ------------------------------------------------------------------ q(x_t | x_0)  (model.py:94-98)
    def _forward_diffusion(self, x_0, t, noise):
        assert x_0.shape == noise.shape
        require_gpu(x_0, "EODiffusion._forward_diffusion")
        self._tables_on(x_0.device)
        x0, nz, t = _f32c(x_0), _f32c(noise), self._t64(t, x_0.device)
        out = torch.empty_like(x0)
        n = x0.shape[0]
        _lib.check(_lib.lib().eod_q_sample(x0.data_ptr(), nz.data_ptr(), t.data_ptr(),
                                           self.sqrt_alphas_cumprod.data_ptr(),
                                           self.sqrt_one_minus_alphas_cumprod.data_ptr(), out.data_ptr(), n,
                                           x0.numel() // n, self.timesteps, current_stream_ptr(x0.device)), "eod_q_sample")
        return out

    def _repaint_mix(self, x_t, gt, mask, t, noise):
        """x_t <- mask*q_sample(gt,t,noise) + (1-mask)*x_t  (model.py:58-60), one fused pass."""
        n, c, h, w = x_t.shape
        x, g, m, z = _f32c(x_t), _f32c(gt), _f32c(mask), _f32c(noise)
        assert g.shape == x.shape and m.shape in ((n, 1, h, w), (n, c, h, w)), (g.shape, m.shape)
        out = torch.empty_like(x)
        if m.shape[1] != 1:  # a mask per channel: every (sample, channel) plane is a one-channel sample of the same kernel
            t = t.repeat_interleave(c)
            n, c = n * c, 1
        _lib.check(_lib.lib().eod_repaint_mix(x.data_ptr(), g.data_ptr(), m.data_ptr(), z.data_ptr(), t.data_ptr(),
                                              self.sqrt_alphas_cumprod.data_ptr(),
                                              self.sqrt_one_minus_alphas_cumprod.data_ptr(), out.data_ptr(), n, c,
                                              h * w, self.timesteps, current_stream_ptr(x.device)), "eod_repaint_mix")
        return out

    @staticmethod
    def _broadcast_mask(mask, like):
        """`mask` as the reference's `img_orig * mask + (1. - mask) * img` (ddim.py:147-148) would broadcast it against `like` [N,C,H,W]:
        anything broadcastable -- [H,W], [1,1,H,W], [N,1,H,W], [N,C,H,W] ... -- becomes [N,1,H,W] (one plane per sample) or, when it
        differs between channels, [N,C,H,W]"""
        n, c, h, w = like.shape
        m = torch.as_tensor(mask, device=like.device).float()
        try:
            shape = torch.broadcast_shapes(tuple(m.shape), (n, c, h, w))
        except RuntimeError:
            shape = None
        if shape != (n, c, h, w):
            raise _lib.EodError(f"mask of shape {tuple(m.shape)} does not broadcast against {(n, c, h, w)}")
        while m.dim() < 4:
            m = m[None]
        return m.expand(n, m.shape[1], h, w).contiguous()

    def _ddpm_update(self, x_t, pred, noise, t, clip):
        x, e, z = _f32c(x_t), _f32c(pred), _f32c(noise)
        out = torch.empty_like(x)
        n = x.shape[0]
        _lib.check(_lib.lib().eod_ddpm_step(x.data_ptr(), e.data_ptr(), z.data_ptr(), t.data_ptr(),
                                            self.betas.data_ptr(), self.alphas.data_ptr(),
                                            self.alphas_cumprod.data_ptr(),
                                            self.sqrt_one_minus_alphas_cumprod.data_ptr(), out.data_ptr(), n,
                                            x.numel() // n, self.timesteps, int(clip), current_stream_ptr(x.device)),
                   "eod_ddpm_step")
        return out

    # ------------------------------------------------------------------ reverse steps (model.py:101-150)
    @torch.no_grad()
    def _reverse_diffusion(self, x_t, t, noise, cond=None, y=None):
        require_gpu(x_t, "EODiffusion._reverse_diffusion")
        self._tables_on(x_t.device)
        t = self._t64(t, x_t.device)
        pred = self.model(x_t, t, cond=cond, y=y)
        return self._ddpm_update(x_t, pred, noise, t, clip=False)

    @torch.no_grad()
    def _reverse_diffusion_with_clip(self, x_t, t, noise, cond=None, y=None):
        require_gpu(x_t, "EODiffusion._reverse_diffusion_with_clip")
        self._tables_on(x_t.device)
        t = self._t64(t, x_t.device)
        pred = self.model(x_t, t, cond=cond, y=y)
        return self._ddpm_update(x_t, pred, noise, t, clip=True)

    # ------------------------------------------------------------------ noise sources
    def _philox(self, shape, dev, seed, sample0, step, stream_id):
        out = torch.empty(shape, dtype=torch.float32, device=dev)
        n = shape[0]
        _lib.check(_lib.lib().eod_randn_philox(out.data_ptr(), n, out.numel() // n, seed, sample0, step, stream_id,
                                               current_stream_ptr(dev)), "eod_randn_philox")
        return out

    # ------------------------------------------------------------------ sampling loop (model.py:46-75)
    @torch.no_grad()
    def sampling(self, n_samples, clipped_reverse_diffusion=True, device="cpu", cond=None, y=None, idx=0, save=False,
                 *, x_T=None, noises=None, rng="torch", seed=0, sample_offset=0, progress=True):
        """Reverse chain t = T-1 ... 0.  RNG order of the reference is kept: x_T is drawn on the CPU
        generator (model.py:48), one `randn_like` per step on the device generator (:55) used for BOTH the
        RePaint q_sample of gt (:59) and the reverse step (:69).
        Extras: x_T / noises ([T,n,C,H,W] or a callable k -> tensor) inject the draws; rng="philox" uses the
        counter-based generator keyed by (seed, sample_offset + n, t) so results do not depend on sharding."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.EodError("EODiffusion.sampling: device must be a HIP GPU ('cuda[:i]'); there is no CPU path")
        self._tables_on(dev)
        shape = (n_samples, self.in_channels, self.image_size, self.image_size)
        if x_T is not None:
            x_t = _f32c(x_T.to(dev))
        elif rng == "philox":
            x_t = self._philox(shape, dev, seed, sample_offset, self.timesteps, 0)
        else:
            x_t = torch.randn(shape).to(dev)
        gt = mask = None
        if cond is not None and self.cond_type == "sum":
            cond = cond.to(dev)
            gt, mask = cond[:n_samples, :3].contiguous(), cond[:n_samples, 3][:, None].contiguous()
            cond = None
        steps = range(self.timesteps - 1, -1, -1)
        it = tqdm(steps, desc="Sampling") if progress else steps
        for k, i in enumerate(it):
            if noises is not None:
                noise = noises(k) if callable(noises) else noises[k]
                noise = _f32c(noise.to(dev))
            elif rng == "philox":
                noise = self._philox(shape, dev, seed, sample_offset, i, 1)
            else:
                noise = torch.randn_like(x_t)
            t = torch.full((n_samples,), i, dtype=torch.int64, device=dev)
            if self.cond_type == "sum" and gt is not None:
                x_t = self._repaint_mix(x_t, gt, mask, t, noise)
            if save and (i % 25 == 0 and i <= 200 or i % 100 == 0 and i <= self.timesteps):
                _save_grid((x_t + 1.0) / 2.0, f"results/prova/s{idx}_{i}_pred.png", int(math.sqrt(n_samples)))
            pred = self.model(x_t, t, cond=cond, y=y)
            x_t = self._ddpm_update(x_t, pred, noise, t, clip=clipped_reverse_diffusion)
        return x_t

    def forward_only(self, img, device="cpu"):
        """Noising-only visualisation helper (model.py:77-84), without the reference's breakpoint()."""
        out = []
        for i in range(self.timesteps - 1, -1, -1):
            noise = torch.randn_like(img)
            t = torch.full((img.shape[0],), i, dtype=torch.int64, device=img.device)
            out.append(self._forward_diffusion(img, t, noise))
        return out


def _save_grid(x, path, nrow):
    """PNG side effect of model.py:62-66 (host-side, off the hot path; PIL instead of torchvision)."""
    try:
        from PIL import Image
    except Exception:
        return
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    x = x.detach().clamp(0, 1).cpu()
    n, c, h, w = x.shape
    nrow = max(1, nrow)
    rows = (n + nrow - 1) // nrow
    grid = torch.zeros(3, rows * h, nrow * w)
    for k in range(n):
        r, q = divmod(k, nrow)
        tile = x[k, :3] if c >= 3 else x[k, :1].expand(3, h, w)
        grid[:, r * h:(r + 1) * h, q * w:(q + 1) * w] = tile
    Image.fromarray((grid.permute(1, 2, 0).numpy() * 255).astype("uint8")).save(path)
