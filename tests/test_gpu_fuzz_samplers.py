"""GPU: seeded random walks over the OPTIONS of the two sampling loops against the CPU oracle -- what the fixed trajectories in
test_gpu_sampling.py pin at a handful of settings.  DDIM (ddim.py:58-207): schedule length T, S sub-steps in both discretisations, eta,
temperature, RePaint mask in every broadcastable shape or none, classifier-free guidance with concat conditioning or none, log_every_t,
batch size.  DDPM (model.py:47-75, 101-150): clipped / plain reverse step, cond_type "sum" (RePaint mix) / concat conditioning / none,
class labels.  Tiny UNets so that a trajectory through the oracle stays under a second."""
import os

import numpy as np
import pytest
import torch

from tests.gpu_util import DEV
from tests.helpers import rel_l2
from tests.synth import rect_mask, synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("EOD_FUZZ_SAMPLER_CASES", "8"))   # (the hunt of round 3 ran 60 + 60: all inside the gate)
FIRST = int(os.environ.get("EOD_FUZZ_FIRST", "0"))
TOL = {"fp32": 5e-5, "fp32x3": 5e-5}   # (up to 25 recursive steps through the UNet)


def _unet(cfg, prec, seed):
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    sd = synth_state_dict(unet_param_shapes(**cfg), seed)
    u = UNetModel(**cfg).set_precision(prec)
    u.load_state_dict(sd)
    return u, sd


@pytest.mark.parametrize("i", range(FIRST, FIRST + N_CASES))
def test_random_ddim_call_vs_oracle(i):
    from eo_diffusion_amd.diffusion.ddim import DDIMSampler
    from eo_diffusion_amd.diffusion.model import EODiffusion
    from oracle import sampler_ref as SR
    from oracle import schedule as SCH
    from oracle import unet_ref as UR
    r = np.random.RandomState(5000 + i)
    T = int(r.choice([20, 50, 100]))
    S = int(r.choice([s for s in (4, 5, 10, 20, 25) if s <= T]))
    method = str(r.choice(["uniform", "uniform", "quad"]))
    eta = float(r.choice([0.0, 0.3, 1.0]))
    temperature = float(r.choice([1.0, 1.0, 0.7]))
    N, H = int(r.choice([1, 2, 3])), int(r.choice([8, 16]))
    guided = bool(r.rand() < 0.4)
    ccond = 4 if guided else int(r.choice([0, 0, 2]))
    masked = bool(r.rand() < 0.5)
    prec = str(r.choice(["fp32", "fp32x3"]))
    cfg = dict(image_size=H, in_channels=3 + ccond, model_channels=32, out_channels=3, num_res_blocks=1,
               attention_resolutions=[2] if r.rand() < 0.5 else [], channel_mult=[1, 2], num_heads=2)
    u, sd = _unet(cfg, prec, 70 + i)
    m = EODiffusion(u, timesteps=T, image_size=H, in_channels=3, device=DEV).to(DEV).eval()
    s = DDIMSampler(m)
    s.make_schedule(ddim_num_steps=S, ddim_discretize=method, ddim_eta=eta, verbose=False)
    steps = SCH.ddim_timesteps(method, S, T)
    assert np.array_equal(np.asarray(s.ddim_timesteps, np.int64), np.asarray(steps, np.int64))
    n_steps = len(steps)
    shape = (N, 3, H, H)
    xT = synth_input(f"sx{i}", shape, 1)
    stp = [synth_input(f"ss{i}_{k}", shape, 2 + k) for k in range(n_steps)]
    mix = [synth_input(f"sm{i}_{k}", shape, 40 + k) for k in range(n_steps)] if masked else None
    x0 = synth_input(f"s0{i}", shape, 3, uniform=True) if masked else None
    c = synth_input(f"sc{i}", (N, ccond, H, H), 4, uniform=True) if ccond else None
    scale = float(r.choice([2.5, 0.5])) if guided else 1.0
    mask = full_mask = None
    if masked:
        kind = int(r.randint(4))
        base = rect_mask(N, H, H, 9 + i)
        mask = [base, base[0, 0], base[:1], (synth_input(f"sk{i}", shape, 5) > 0.0).float()][kind]   # [N,1,H,W] [H,W] [1,1,H,W] [N,C,H,W]
        full_mask = torch.broadcast_to(mask, shape)
    log_every = int(r.choice([1, 3, 100]))
    out, inter = s.ddim_sampling(None if c is None else c.to(DEV), shape, x_T=xT, mask=mask, x0=x0, temperature=temperature,
                                 log_every_t=log_every, unconditional_guidance_scale=scale,
                                 unconditional_conditioning=None if not guided else torch.zeros_like(c).to(DEV),
                                 step_noises=stp, mix_noises=mix, progress=False)

    def eps(img, ts):
        if not guided or scale == 1.0:
            return UR.unet_forward(sd, cfg, img, ts, cond=c)
        e_u = UR.unet_forward(sd, cfg, img, ts, cond=torch.zeros_like(c))
        return e_u + scale * (UR.unet_forward(sd, cfg, img, ts, cond=c) - e_u)

    tb = SCH.eo_cosine_tables(T)
    dd = SCH.ddim_tables(tb["alphas_cumprod"], steps, eta)
    with torch.no_grad():
        ref, ref_p0 = SR.ddim_sampling(tb, dd, steps, eps, xT, stp, x0=x0, mask=full_mask, mix_noises=mix, temperature=temperature)
    e1, e2 = rel_l2(out.cpu(), ref), rel_l2(inter["pred_x0"][-1].cpu(), ref_p0)
    print(f"ddim case {i}: T {T} S {S} {method} eta {eta} temp {temperature} N {N} H {H} cond {ccond} guided {guided} x{scale} masked {masked} "
          f"[{prec}] -> {e1:.2e} {e2:.2e}")
    assert e1 < TOL[prec] and e2 < TOL[prec]
    n_log = sum(1 for k in range(n_steps) if (n_steps - k - 1) % log_every == 0 or k == 0)
    assert len(inter["x_inter"]) == 1 + n_log == len(inter["pred_x0"])


@pytest.mark.parametrize("i", range(FIRST, FIRST + N_CASES))
def test_random_ddpm_call_vs_oracle(i):
    from eo_diffusion_amd.diffusion.model import EODiffusion
    from oracle import sampler_ref as SR
    from oracle import schedule as SCH
    from oracle import unet_ref as UR
    r = np.random.RandomState(7000 + i)
    T = int(r.choice([8, 20, 30]))
    N, H = int(r.choice([1, 2, 4])), int(r.choice([8, 16]))
    mode = str(r.choice(["none", "sum", "concat"]))
    clip = bool(r.rand() < 0.6)
    labelled = bool(r.rand() < 0.3)
    prec = str(r.choice(["fp32", "fp32x3"]))
    ccond = 2 if mode == "concat" else 0
    cfg = dict(image_size=H, in_channels=3 + ccond, model_channels=32, out_channels=3, num_res_blocks=1,
               attention_resolutions=[1] if r.rand() < 0.4 else [], channel_mult=[1, 2], num_heads=2,
               use_scale_shift_norm=bool(r.rand() < 0.4))
    if labelled:
        cfg["num_classes"] = 4
    u, sd = _unet(cfg, prec, 90 + i)
    m = EODiffusion(u, timesteps=T, image_size=H, in_channels=3, cond_type="sum" if mode == "sum" else None, device=DEV).to(DEV).eval()
    shape = (N, 3, H, H)
    xT = synth_input(f"px{i}", shape, 1)
    noises = [synth_input(f"pn{i}_{k}", shape, 2 + k) for k in range(T)]
    y = torch.tensor([(i + k) % 4 for k in range(N)]) if labelled else None
    cond = gt_img = mask = None
    if mode == "sum":   # inference.py:100-109: cond = cat(image, mask) with mask = 1 inside the KEPT region
        gt_img, mask = synth_input(f"pg{i}", shape, 3, uniform=True), rect_mask(N, H, H, 11 + i)
        cond = torch.cat([gt_img, mask], 1)
    elif mode == "concat":
        cond = synth_input(f"pc{i}", (N, ccond, H, H), 4, uniform=True)
    out = m.sampling(N, clipped_reverse_diffusion=clip, device=DEV, cond=None if cond is None else cond.to(DEV),
                     y=None if y is None else y.to(DEV), x_T=xT, noises=noises, progress=False)
    tb = SCH.eo_cosine_tables(T)
    eps = lambda img, ts: UR.unet_forward(sd, cfg, img, ts, cond=cond if mode == "concat" else None, y=y)
    with torch.no_grad():
        ref = SR.ddpm_sampling(tb, eps, xT, noises, T, clip=clip, gt=gt_img, mask=mask)
    e = rel_l2(out.cpu(), ref)
    print(f"ddpm case {i}: T {T} N {N} H {H} {mode} clip {clip} labelled {labelled} film {cfg['use_scale_shift_norm']} [{prec}] -> {e:.2e}")
    assert torch.isfinite(out).all() and e < TOL[prec]
