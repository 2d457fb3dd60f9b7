"""TEST INFRASTRUCTURE (oracle): CPU restatement of the optimizer side of the reference's training step
(train.py:75,86,116-122 and script_utils/utils.py:56-67).  Only tests/ may import this.

The arithmetic lives in a third-party dependency, PyTorch (pinned 1.13.0 in eo_diffusion.yml:114): this file restates
  * nn.MSELoss(reduction='mean') and its gradient,
  * torch.optim.AdamW's single-tensor algorithm (torch/optim/adamw.py `_single_tensor_adamw`, 1.13):
        param.mul_(1 - lr * weight_decay)
        exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1)
        exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        denom = (exp_avg_sq.sqrt() / sqrt(1 - beta2**step)).add_(eps)
        param.addcdiv_(exp_avg, denom, value=-(lr / (1 - beta1**step)))
  * the EMA lambda of utils.py:63-65: decay * avg + (1 - decay) * param
in float32 numpy, every operation rounded separately (scalars are rounded to float32 first, as ATen does for float
tensors).  Pinned by tests/test_oracle_golden.py::test_train_ref_* against torch.optim.AdamW / F.mse_loss of the torch
build in the container (2.10, CPU) within a few ulp (that build fuses some multiply-adds)."""
import math

import numpy as np

f32 = np.float32


def mse_loss(pred, target):
    """returns (loss, dpred) with dpred = 2*(pred-target)/n"""
    d = pred.astype(f32) - target.astype(f32)
    n = d.size
    loss = f32(np.sum((d * d).astype(np.float64)) / n)  # summation order is implementation-defined: compare with a tolerance
    return loss, (d * f32(2.0 / n)).astype(f32)


def adamw_step(p, g, m, v, *, lr, beta1, beta2, eps, weight_decay, step):
    p, g, m, v = (a.astype(f32) for a in (p, g, m, v))
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = lr / bc1
    p = p * f32(1.0 - lr * weight_decay)
    m = m * f32(beta1) + g * f32(1.0 - beta1)
    v = v * f32(beta2) + (f32(1.0 - beta2) * g) * g
    denom = np.sqrt(v) / f32(math.sqrt(bc2)) + f32(eps)
    p = p + f32(-step_size) * (m / denom)
    return p.astype(f32), m.astype(f32), v.astype(f32)


def ema_update(avg, p, decay):
    return (f32(decay) * avg.astype(f32) + f32(1.0 - decay) * p.astype(f32)).astype(f32)
