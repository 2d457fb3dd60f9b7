"""Harness-side operations of the reference's inference.py around the sampling call.

    cond = assemble_repaint_cond(image, mask)           inference.py:100-109   mask = 1 - mask; cond = cat((image, mask), 1)
    samples = postprocess_samples(samples, image)       inference.py:128       clip(0, 1) for [0,1] data, (x + 1) / 2 for [-1,1] data
    preview = masked_preview(image, mask)               inference.py:134       image * (mask + 0.7).clip(0, 1)
    psnr(samples, gt), ssim(samples, gt)                inference.py:136-138   torchmetrics' functional PSNR / SSIM (data_range 1)
    make_label(shape, 10, 10, 40, 40)                   script_utils/utils.py:17-37 (random rectangle of --random_label)

The elementwise ones are fused HIP kernels (one pass each, bit-exact vs the torch expressions; csrc/sampler.hip).  PSNR reuses the
MSE kernel of the training path; SSIM is a host-side numpy evaluation (it runs once per saved batch, off the hot path), and
make_label is host-side numpy exactly like the reference's."""
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr
from .engine import current_stream_ptr, require_gpu


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def assemble_repaint_cond(image, mask, invert=True):
    """cond [N, C+1, H, W] = cat(image, 1 - mask) (invert=True: the dataset's segmentation marks the region to REPAINT, the
    sampler wants 1 = keep, inference.py:102).  image [N,C,H,W], mask [N,1,H,W] or [N,H,W]."""
    require_gpu(image, "assemble_repaint_cond")
    x = _f32c(image)
    n, c, h, w = x.shape
    m = _f32c(mask.to(x.device)).reshape(n, 1, h, w)
    out = torch.empty((n, c + 1, h, w), dtype=torch.float32, device=x.device)
    check(_lib.lib().eod_repaint_cond(ptr(x), ptr(m), ptr(out), n, c, h * w, int(bool(invert)), current_stream_ptr(x.device)), "eod_repaint_cond")
    return out


def postprocess_samples(samples, image=None, *, data_nonneg=None):
    """inference.py:128: `samples.clip(0,1) if image.min() >= 0 else (samples + 1.) / 2.`.  Pass data_nonneg to skip the host
    synchronisation of `image.min()` (the data range is a property of the dataset, not of the batch)."""
    require_gpu(samples, "postprocess_samples")
    if data_nonneg is None:
        if image is None:
            raise ValueError("postprocess_samples needs `image` or `data_nonneg`")
        data_nonneg = bool(image.min() >= 0)
    x = _f32c(samples)
    out = torch.empty_like(x)
    check(_lib.lib().eod_postprocess(ptr(x), ptr(out), x.numel(), 0 if data_nonneg else 1, current_stream_ptr(x.device)), "eod_postprocess")
    return out


def masked_preview(image, mask, lift=0.7):
    """inference.py:134: the conditioning picture that is saved next to the sample: image * (mask + 0.7).clip(0, 1)"""
    require_gpu(image, "masked_preview")
    x = _f32c(image)
    n, c, h, w = x.shape
    m = _f32c(mask.to(x.device)).reshape(n, 1, h, w)
    out = torch.empty_like(x)
    check(_lib.lib().eod_masked_preview(ptr(x), ptr(m), ptr(out), n, c, h * w, float(lift), current_stream_ptr(x.device)), "eod_masked_preview")
    return out


def psnr(preds, target, data_range=1.0):
    """torchmetrics.functional.peak_signal_noise_ratio(preds, target, data_range) with its defaults (base 10, mean over all
    elements): 10 log10(data_range^2 / mse).  The squared-error reduction runs on the GPU (eod_mse_loss)."""
    from .optim import mse_loss
    require_gpu(preds, "psnr")
    loss, _ = mse_loss(_f32c(preds), _f32c(target.to(preds.device)), want_grad=False)
    mse = float(loss)
    return float("inf") if mse == 0.0 else 10.0 * math.log10(data_range * data_range / mse)


def _gauss_kernel(size=11, sigma=1.5):
    d = np.arange((1 - size) / 2.0, (1 + size) / 2.0, 1.0)
    g = np.exp(-((d / sigma) ** 2) / 2.0)
    return g / g.sum()


def ssim(preds, target, data_range=1.0, kernel_size=11, sigma=1.5, k1=0.01, k2=0.03):
    """torchmetrics.functional.structural_similarity_index_measure with its defaults (gaussian 11x11 window, sigma 1.5, reflect
    padding of (k-1)/2, the padded border cropped again, mean over pixels, channels and batch).  Host-side (numpy, float64 window
    sums): evaluated once per saved batch."""
    p = preds.detach().float().cpu().numpy().astype(np.float64)
    t = target.detach().float().cpu().numpy().astype(np.float64)
    assert p.shape == t.shape and p.ndim == 4
    pad = (kernel_size - 1) // 2
    g = _gauss_kernel(kernel_size, sigma)

    def blur(a):
        a = np.pad(a, ((0, 0), (0, 0), (pad, pad), (pad, pad)), mode="reflect")
        a = np.apply_along_axis(lambda v: np.convolve(v, g, mode="valid"), 2, a)
        return np.apply_along_axis(lambda v: np.convolve(v, g, mode="valid"), 3, a)

    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    mu_p, mu_t = blur(p), blur(t)
    s_pp, s_tt, s_pt = blur(p * p) - mu_p * mu_p, blur(t * t) - mu_t * mu_t, blur(p * t) - mu_p * mu_t
    full = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2))
    full = full[..., pad:-pad, pad:-pad] if pad and full.shape[-1] > 2 * pad and full.shape[-2] > 2 * pad else full
    return float(full.reshape(full.shape[0], -1).mean(-1).mean())


def make_label(shape, mnw, mnh, mxw, mxh, rng=None):
    """script_utils/utils.py:17-37: a [w, h] array of zeros with one random rectangle of ones, its side lengths drawn between
    mn% and mx% of the image (np.random.randint, same draw order: ws, hs, x, y).  `rng` (a numpy RandomState-like with
    `.randint`) defaults to numpy's global state, as in the reference."""
    rng = rng or np.random
    label = np.zeros(shape)
    w, h = shape
    mnw, mxw, mnh, mxh = int(w * mnw / 100), int(w * mxw / 100), int(h * mnh / 100), int(h * mxh / 100)
    ws = rng.randint(mnw, mxw, 1)[0]
    hs = rng.randint(mnh, mxh, 1)[0]
    x = rng.randint(ws, w - ws, 1)[0]
    y = rng.randint(hs, h - hs, 1)[0]
    label[x:x + ws, y:y + hs] = 1.0
    return label
