"""GPU: the cached launch plan of a UNetModel through the things a caller does to a module between forwards -- other batch sizes and
map sizes, new weights (load_state_dict, in-place optimizer-style updates, parameter re-pointing), precision switches, train / eval,
deepcopy (what AveragedModel does, utils.py:56-67) -- every forward checked against the oracle ON THE WEIGHTS OF THAT MOMENT.  One
live plan per model is kept; a stale plan would show up as the previous call's result."""
import copy

import pytest
import torch

from tests.gpu_util import DEV, TOL
from tests.helpers import rel_l2
from tests.synth import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
CFG = dict(image_size=16, in_channels=3, out_channels=3, model_channels=32, channel_mult=[1, 2], num_res_blocks=1, attention_resolutions=[2],
           num_heads=2)


def _check(u, sd, shape, prec, tag, seed):
    from oracle import unet_ref as UR
    N, _, H, W = shape
    x = synth_input(f"lc{tag}", shape, seed)
    t = torch.tensor([(97 * seed + 211 * k) % 1000 for k in range(N)])
    with torch.no_grad():
        out = u(x.to(DEV), t.to(DEV)).cpu()
        ref = UR.unet_forward({k: v.detach().cpu() for k, v in sd.items()}, CFG, x, t)
    e = rel_l2(out, ref)
    assert e < TOL[prec], (tag, prec, e)


@pytest.mark.parametrize("prec", ["fp32x3", "fp16"])
def test_plan_follows_the_module_through_its_life(prec):
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    shapes = unet_param_shapes(**CFG)
    sd = synth_state_dict(shapes, 7)
    u = UNetModel(**CFG).set_precision(prec)
    u.load_state_dict(sd)
    u = u.to(DEV).eval()
    live = lambda: {k: v for k, v in u.state_dict().items()}
    _check(u, live(), (2, 3, 16, 16), prec, "first", 1)
    _check(u, live(), (3, 3, 16, 16), prec, "other batch", 2)
    _check(u, live(), (1, 3, 32, 16), prec, "other map", 3)
    _check(u, live(), (2, 3, 16, 16), prec, "back to the first shape", 4)
    u.load_state_dict(synth_state_dict(shapes, 8))                      # new weights into the same parameter tensors
    _check(u, live(), (2, 3, 16, 16), prec, "after load_state_dict", 5)
    with torch.no_grad():                                               # in-place updates, as an optimizer makes them
        for p in u.parameters():
            p.mul_(0.9).add_(0.01)
    _check(u, live(), (2, 3, 16, 16), prec, "after in-place updates", 6)
    with torch.no_grad():                                               # a parameter re-pointed to new storage (p.data = ...)
        w = u.input_blocks[0][0].weight
        w.data = (w.data * 1.5).clone()
    _check(u, live(), (2, 3, 16, 16), prec, "after re-pointing a parameter", 7)
    other = "fp32" if prec != "fp32" else "fp16"
    u.set_precision(other)
    _check(u, live(), (2, 3, 16, 16), other, "after a precision switch", 8)
    u.set_precision(prec)
    _check(u, live(), (2, 3, 16, 16), prec, "and back", 9)
    u.train()
    _check(u, live(), (2, 3, 16, 16), prec, "train mode without autograd (dropout 0)", 10)
    u.eval()
    c = copy.deepcopy(u)                                                # AveragedModel's deep copy
    with torch.no_grad():
        for p in c.parameters():
            p.mul_(1.1)
    _check(c, {k: v for k, v in c.state_dict().items()}, (2, 3, 16, 16), prec, "deep copy with its own weights", 11)
    _check(u, live(), (2, 3, 16, 16), prec, "the original after the copy changed", 12)
    u2 = u.to("cpu").to(DEV)                                            # a round trip through the host moves every parameter
    _check(u2, {k: v for k, v in u2.state_dict().items()}, (2, 3, 16, 16), prec, "after .to(cpu).to(gpu)", 13)


def test_forward_accepts_the_input_forms_torch_modules_accept():
    """what the reference's nn.Module forward takes without complaint: non-contiguous / channels_last / fp64 / fp16 images, timesteps as
    int32, on the host, or as whole-number floats, cond in another dtype -- all give the bits of the plain call (fp16 / fp64 inputs are
    converted to fp32 first, as `h = x.type(self.dtype)` does, unet_openai.py:762); fractional timesteps follow the reference's
    `timesteps.float()` embedding in inference and are refused loudly by the training step"""
    from eo_diffusion_amd._lib import EodError
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    from oracle import unet_ref as UR
    cfg = dict(CFG, in_channels=6)
    sd = synth_state_dict(unet_param_shapes(**cfg), 5)
    u = UNetModel(**cfg).set_precision("fp32x3")
    u.load_state_dict(sd)
    u = u.to(DEV).eval()
    N, H = 2, cfg["image_size"]
    x = synth_input("ifx", (N, 3, H, H), 1)
    c = synth_input("ifc", (N, 3, H, H), 2)
    t = torch.tensor([3, 977])
    with torch.no_grad():
        base = u(x.to(DEV), t.to(DEV), cond=c.to(DEV))
        assert rel_l2(base.cpu(), UR.unet_forward(sd, cfg, x, t, cond=c)) < TOL["fp32x3"]
        wide = torch.zeros(N, 3, H, 2 * H)
        wide[..., ::2] = x
        forms = {
            "strided": (wide.to(DEV)[..., ::2], t.to(DEV), c.to(DEV)),
            "channels_last": (x.to(DEV).contiguous(memory_format=torch.channels_last), t.to(DEV), c.to(DEV)),
            "fp64": (x.double().to(DEV), t.to(DEV), c.to(DEV)),
            "t_int32": (x.to(DEV), t.int().to(DEV), c.to(DEV)),
            "t_host": (x.to(DEV), t, c.to(DEV)),
            "t_float": (x.to(DEV), t.float().to(DEV), c.to(DEV)),
            "cond_fp64_host": (x.to(DEV), t.to(DEV), c.double()),
        }
        for name, (xi, ti, ci) in forms.items():
            out = u(xi, ti, cond=ci)
            assert out.dtype == xi.dtype and torch.equal(out.float(), base), name
        xh = x.half().float()  # (an fp16 image is an fp32 image whose values are fp16 numbers)
        out16 = u(x.half().to(DEV), t.to(DEV), cond=c.to(DEV))
        assert out16.dtype == torch.float16 and torch.equal(out16, u(xh.to(DEV), t.to(DEV), cond=c.to(DEV)).half())
        tfrac = torch.tensor([3.5, 976.25])   # fractional timesteps: `timesteps[:, None].float() * freqs` (unet_openai.py:95)
        for tf in (tfrac.to(DEV), tfrac.double()):
            assert rel_l2(u(x.to(DEV), tf, cond=c.to(DEV)).cpu(), UR.unet_forward(sd, cfg, x, tfrac, cond=c)) < TOL["fp32x3"]
        assert rel_l2(u(x.to(DEV), tfrac.to(DEV), cond=c.to(DEV)), base) > 1e-4
    with pytest.raises(EodError, match="fractional"):   # (the training step's backward recomputes the sinusoid from int64)
        u.train()(x.to(DEV), torch.tensor([3.5, 10.0]).to(DEV), cond=c.to(DEV))
