"""GPU regression tests for state that lives OUTSIDE torch's view of the parameters: packed-weight caches behind raw-pointer
updates (fused EMA / AdamW), trainers whose descriptors bake parameter pointers, parameters without gradients, and
caller-supplied timesteps that index the schedule tables."""
import pytest
import torch

from tests.gpu_util import DEV
from tests.helpers import bits_equal, unet_cfgs
from tests.synth import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu


def _diffusion(prec="fp32", seed=7, T=20):
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    from eo_diffusion_amd.diffusion.model import EODiffusion
    cfg = unet_cfgs()["u_a1_tiny"]
    u = UNetModel(**cfg).set_precision(prec)
    u.load_state_dict(synth_state_dict(unet_param_shapes(**cfg), seed))
    return EODiffusion(u, timesteps=T, image_size=cfg["image_size"], in_channels=3, device=DEV).to(DEV), cfg


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_ema_forward_sees_every_update(prec):
    """train.py:149 samples from model_ema.module again and again: the EMA copy's cached launch program (packed conv / qkv /
    proj weights) must be rebuilt after EVERY fused EMA update, not only after the first two"""
    import copy
    from eo_diffusion_amd.optim import ExponentialMovingAverage
    model, cfg = _diffusion(prec)
    ema = ExponentialMovingAverage(model, decay=0.5, device=DEV)
    x = synth_input("ema_x", (2, 3, cfg["image_size"], cfg["image_size"]), 1).to(DEV)
    t = torch.tensor([3, 11], device=DEV)
    ema.eval()
    for step in range(4):
        with torch.no_grad():
            for k, p in enumerate(model.parameters()):  # "optimizer step": move the live weights
                p.add_(0.01 * (step + 1) * torch.sign(p))
        ema.update_parameters(model)
        with torch.no_grad():
            out = ema.module.model(x, t)
            fresh = copy.deepcopy(model).eval()  # a module that has never run: no cache of any kind
            fresh.load_state_dict(ema.module.state_dict())
            ref = fresh.model(x, t)
        assert bits_equal(out, ref), f"EMA forward after update {step + 1} used stale packed weights"


def test_trainer_notices_repointed_parameters():
    """optim.AdamW moves every parameter into one flat buffer.  A UNetTrainer built BEFORE that holds pointers to the old
    storage: the direct API raises, the autograd bridge rebuilds -- and the next forward sees the optimizer's bias / GroupNorm /
    timestep-MLP updates."""
    import eo_diffusion_amd.backbones.unet_openai as U
    from eo_diffusion_amd._lib import EodError
    from eo_diffusion_amd.optim import AdamW
    from eo_diffusion_amd.training import UNetTrainer
    cfg = unet_cfgs()["u_a0_tiny"]
    m = U.UNetModel(**cfg)
    m.load_state_dict(synth_state_dict(U.unet_param_shapes(**cfg), 7))
    m = m.to(DEV).train()
    S = cfg["image_size"]
    x, noise = synth_input("rp_x", (2, 3, S, S), 1).to(DEV), synth_input("rp_n", (2, 3, S, S), 2).to(DEV)
    t = torch.tensor([5, 900], device=DEV)
    tr = UNetTrainer(m, 2, S, S, DEV)
    tr.forward(x, t)
    loss0 = torch.nn.functional.mse_loss(m(x, t), noise)  # autograd bridge: builds its own trainer
    loss0.backward()
    opt = AdamW(m.parameters(), lr=5e-2)  # re-points p.data of every parameter
    assert tr.stale()
    with pytest.raises(EodError):
        tr.forward(x, t)
    opt.step()
    opt.zero_grad()
    # biases / GroupNorm affine / timestep MLP are read through live pointers, conv weights are re-packed: compare against a
    # fresh module that loads the UPDATED state_dict
    loss1 = torch.nn.functional.mse_loss(m(x, t), noise)  # must rebuild, not reuse the stale trainer
    fresh = U.UNetModel(**cfg)
    fresh.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    fresh = fresh.to(DEV).train()
    loss_ref = torch.nn.functional.mse_loss(fresh(x, t), noise)
    assert float(loss1.detach()) == float(loss_ref.detach()), (float(loss1.detach()), float(loss_ref.detach()))
    assert float(loss1.detach()) != float(loss0)


def test_unused_head_parameters_are_left_alone():
    """nout / conv_out (unet_openai.py:744) never take part in the forward: autograd gives them no gradient and torch.optim.AdamW
    skips them -- the fused AdamW must not apply weight decay to them either"""
    import eo_diffusion_amd.backbones.unet_openai as U
    from eo_diffusion_amd.optim import AdamW
    cfg = unet_cfgs()["u_a0_tiny"]
    m = U.UNetModel(**cfg)
    m.load_state_dict(synth_state_dict(U.unet_param_shapes(**cfg), 7))
    m = m.to(DEV).train()
    S = cfg["image_size"]
    x, noise = synth_input("uh_x", (2, 3, S, S), 1).to(DEV), synth_input("uh_n", (2, 3, S, S), 2).to(DEV)
    t = torch.tensor([5, 900], device=DEV)
    opt = AdamW(m.parameters(), lr=1e-2, weight_decay=0.1)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    for _ in range(3):
        torch.nn.functional.mse_loss(m(x, t), noise).backward()
        dead = [n for n, p in m.named_parameters() if p.grad is None]
        assert sorted(dead) == ["conv_out.bias", "conv_out.weight", "nout.bias", "nout.weight"], dead
        opt.step()
        opt.zero_grad()
    for n, p in m.named_parameters():
        same = bits_equal(p.detach(), before[n])
        assert same == (n.startswith("nout.") or n.startswith("conv_out.")), n


def test_out_of_range_timestep_is_loud_and_memory_safe():
    model, cfg = _diffusion()
    S = cfg["image_size"]
    x = synth_input("oor_x", (2, 3, S, S), 1).to(DEV)
    z = synth_input("oor_z", (2, 3, S, S), 2).to(DEV)
    with pytest.raises(IndexError):  # host-side tensor: checked before anything is launched
        model._forward_diffusion(x, torch.tensor([3, 20]), z)
    with pytest.raises(IndexError):
        model._forward_diffusion(x, torch.tensor([-1, 3]), z)
    # device-side tensor: no host sync; the offending sample comes back as NaN, its neighbour is untouched
    ok = model._forward_diffusion(x, torch.tensor([3, 7], device=DEV), z)
    bad = model._forward_diffusion(x, torch.tensor([3, 1 << 40], device=DEV), z)
    assert bits_equal(bad[0], ok[0]) and bool(torch.isnan(bad[1]).all())
    bad = model._ddpm_update(x, z, z, torch.tensor([-5, 7], device=DEV), clip=True)
    assert bool(torch.isnan(bad[0]).all()) and bool(torch.isfinite(bad[1]).all())


def test_fused_adamw_skips_steps_with_nonfinite_gradients():
    """fp16 training with a static loss scale: an overflowing gradient must not poison the weights -- the fused AdamW skips that
    step on the device (parameters AND moments untouched), counts it, and carries on with the next finite one"""
    from eo_diffusion_amd.optim import AdamW
    ps = [torch.nn.Parameter(synth_input(f"nf_p{k}", shp, 1).to(DEV)) for k, shp in enumerate([(64, 32, 3, 3), (64,), (1000, 7)])]
    opt = AdamW(ps, lr=1e-2)
    ref = torch.optim.AdamW([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-2)
    grads = [[synth_input(f"nf_g{s}{k}", p.shape, 2).to(DEV) for k, p in enumerate(ps)] for s in range(3)]
    for s in range(3):
        for p, q, g in zip(ps, ref.param_groups[0]["params"], grads[s]):
            p.grad, q.grad = g.clone(), g.clone()
        if s == 1:
            ps[2].grad[17, 3] = float("inf")   # this step overflows
            before = [p.detach().clone() for p in ps]
            opt.step()
            assert all(bits_equal(p.detach(), b) for p, b in zip(ps, before))
            continue
        opt.step()
        ref.step()
    assert opt.skipped_steps() == 1
    # two applied steps: the bias corrections follow the number of APPLIED steps (counted on the device next to the skip flag), so the
    # result is torch.optim.AdamW's after its two steps -- a skipped step never reaches the optimizer, as with GradScaler
    for p, q in zip(ps, ref.param_groups[0]["params"]):
        assert torch.allclose(p.detach(), q.detach(), rtol=1e-5, atol=1e-6)
        assert bool(torch.isfinite(p).all())
