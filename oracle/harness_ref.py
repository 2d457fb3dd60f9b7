"""Oracle (test infrastructure -- see oracle/__init__.py): the harness-side tensor expressions of inference.py, restated with
stock torch CPU ops, and the published PSNR / SSIM definitions the reference takes from torchmetrics.

  repaint_cond     inference.py:100-109   mask = 1 - mask ; cond = torch.cat((image, cond), dim=1)
  postprocess      inference.py:128       samples.clip(0,1) if image.min() >= 0 else (samples + 1.) / 2.
  masked_preview   inference.py:134       image * ((mask + 0.7).clip(0, 1))
  psnr / ssim      inference.py:136-138   torchmetrics.functional.{peak_signal_noise_ratio, structural_similarity_index_measure}

PARITY of psnr / ssim: torchmetrics is a third-party dependency that is neither vendored in /root/reference nor installed here
(eo_diffusion.yml pins torchmetrics==0.11.4); the two functions below restate its published algorithm (gaussian 11x11 window with
sigma 1.5, k1 0.01, k2 0.03, reflect padding, border crop; PSNR = 10 log10(range^2 / MSE)) with torch conv2d.  No golden values
exist for them in the reference: parity of these two metrics is UNPINNED beyond the published formula."""
import torch
import torch.nn.functional as F


def repaint_cond(image, mask, invert=True):
    m = 1 - mask if invert else mask
    return torch.cat((image, m), dim=1)


def postprocess(samples, image):
    return samples.clip(0, 1) if image.min() >= 0 else (samples + 1.) / 2.


def masked_preview(image, mask):
    return image * ((mask + 0.7).clip(0, 1))


def psnr(preds, target, data_range=1.0):
    mse = torch.mean((preds.double() - target.double()) ** 2)
    return float(10.0 * torch.log10(data_range ** 2 / mse))


def ssim(preds, target, data_range=1.0, kernel_size=11, sigma=1.5, k1=0.01, k2=0.03):
    p, t = preds.double(), target.double()
    c = p.shape[1]
    dist = torch.arange((1 - kernel_size) / 2, (1 + kernel_size) / 2, 1, dtype=torch.float64)
    g = torch.exp(-((dist / sigma) ** 2) / 2)
    g = (g / g.sum()).unsqueeze(0)
    kernel = (g.t() @ g).expand(c, 1, kernel_size, kernel_size)
    pad = (kernel_size - 1) // 2
    p, t = F.pad(p, (pad, pad, pad, pad), mode="reflect"), F.pad(t, (pad, pad, pad, pad), mode="reflect")
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    stack = torch.cat((p, t, p * p, t * t, p * t))
    out = F.conv2d(stack, kernel, groups=c)
    mu_p, mu_t, e_pp, e_tt, e_pt = out.split(preds.shape[0])
    s_pp, s_tt, s_pt = e_pp - mu_p ** 2, e_tt - mu_t ** 2, e_pt - mu_p * mu_t
    full = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p ** 2 + mu_t ** 2 + c1) * (s_pp + s_tt + c2))
    full = full[..., pad:-pad, pad:-pad]
    return float(full.reshape(full.shape[0], -1).mean(-1).mean())
