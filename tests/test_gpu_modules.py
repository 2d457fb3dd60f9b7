"""GPU parity: product modules (ResBlock / AttentionBlock / Upsample / Downsample / UNetModel) through their
reference-shaped Python API vs the golden vectors generated from the reference."""
import pytest
import torch

from tests.gpu_util import DEV, TOL, load_into
from tests.helpers import gt, rel_l2, unet_cfgs
from tests.synth import synth_state_dict
from tests.test_oracle_golden import ATT, RES, attn_shapes, res_shapes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("name", list(RES))
def test_resblock_vs_golden(prec, name, monkeypatch):
    from eo_diffusion_amd.backbones.unet_openai import ResBlock
    monkeypatch.setenv("EOD_PRECISION", prec)
    cin, cout, kw = RES[name]
    g = gt("mod_" + name)
    blk = ResBlock(cin, 128, 0.0, out_channels=cout, use_conv=kw.get("skip3", False),
                   use_scale_shift_norm=kw.get("film", False), up=kw.get("up", False), down=kw.get("down", False))
    load_into(blk, synth_state_dict(res_shapes(cin, cout, kw.get("film", False), kw.get("skip3", False)), 3))
    with torch.no_grad():
        y = blk(g["x"].to(DEV), g["emb"].to(DEV)).cpu()
    assert y.shape == g["y"].shape
    assert rel_l2(y, g["y"]) < TOL[prec]


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("name", list(ATT))
def test_attention_vs_golden(prec, name, monkeypatch):
    from eo_diffusion_amd.backbones.unet_openai import AttentionBlock
    monkeypatch.setenv("EOD_PRECISION", prec)
    C, heads, new = ATT[name]
    g = gt("mod_" + name)
    nhc = 128 if name == "attn_c128_d128_new" else -1
    blk = AttentionBlock(C, num_heads=heads, num_head_channels=nhc, use_new_attention_order=new)
    load_into(blk, synth_state_dict(attn_shapes(C), 4))
    with torch.no_grad():
        y = blk(g["x"].to(DEV)).cpu()
    assert rel_l2(y, g["y"]) < TOL[prec]


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
def test_resample_vs_golden(prec, monkeypatch):
    from eo_diffusion_amd.backbones.unet_openai import Downsample, Upsample
    monkeypatch.setenv("EOD_PRECISION", prec)
    conv = lambda p: {p + ".weight": (32, 32, 3, 3), p + ".bias": (32,)}
    for name, cls, use_conv, key in (("up_conv", Upsample, True, "conv"), ("up_conv_3x3", Upsample, True, "conv"),
                                     ("up_noconv", Upsample, False, None), ("down_conv", Downsample, True, "op"),
                                     ("down_conv_odd", Downsample, True, "op"), ("down_pool", Downsample, False, None)):
        g = gt("mod_" + name)
        blk = cls(32, use_conv)
        if use_conv:
            blk.load_state_dict(synth_state_dict(conv(key), 5))
        blk = blk.to(DEV).eval()
        with torch.no_grad():
            y = blk(g["x"].to(DEV)).cpu()
        assert y.shape == g["y"].shape, name
        assert rel_l2(y, g["y"]) < TOL[prec], name


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("name", list(unet_cfgs()))
def test_unet_vs_golden(prec, name):
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    cfg = unet_cfgs()[name]
    g = gt("unet_" + name)
    u = UNetModel(**cfg).set_precision(prec)
    load_into(u, synth_state_dict(unet_param_shapes(**cfg), 7))
    cond = g["cond"].to(DEV) if "cond" in g else None
    y = g["y"].to(DEV) if "y" in g else None
    with torch.no_grad():
        out = u(g["x"].to(DEV), g["t"].to(DEV), cond=cond, y=y)
        out2 = u(g["x"].to(DEV), g["t"].to(DEV), cond=cond, y=y)  # cached program replay
    assert out.shape == g["y_out"].shape and out.dtype == torch.float32
    assert rel_l2(out.cpu(), g["y_out"]) < TOL[prec]
    assert torch.equal(out, out2)  # deterministic (no atomics anywhere on the path)


def test_product_refuses_cpu_tensors():
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.backbones.unet_openai import UNetModel
    cfg = unet_cfgs()["u_a0_tiny"]
    u = UNetModel(**cfg)
    with pytest.raises(_lib.EodError, match="no CPU"):
        with torch.no_grad():
            u(torch.zeros(1, 3, 16, 16), torch.zeros(1, dtype=torch.long))


def test_state_dict_roundtrip_and_repack_on_update():
    """load_state_dict after a forward must invalidate the packed-weight plan (keyed on param versions)."""
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    cfg = unet_cfgs()["u_a0_tiny"]
    g = gt("unet_u_a0_tiny")
    u = UNetModel(**cfg).to(DEV).eval()
    with torch.no_grad():
        y0 = u(g["x"].to(DEV), g["t"].to(DEV))  # default init: zero_module convs -> output is exactly the out-conv bias (0)
        assert float(y0.abs().max()) == 0.0
        u.load_state_dict(synth_state_dict(unet_param_shapes(**cfg), 7))
        y1 = u(g["x"].to(DEV), g["t"].to(DEV))
    assert rel_l2(y1.cpu(), g["y_out"]) < TOL["fp32"]


@pytest.mark.parametrize("C,heads,hw", [(128, 2, (32, 32)), (96, 2, (40, 24)), (64, 4, (17, 9)), (256, 4, (64, 64))])
def test_flash_attention_longer_sequences(C, heads, hw, monkeypatch):
    """fused flash-style kernel (fp16, d in {64, 48, 16}) on multi-tile sequences incl. ragged T, vs the CPU oracle, and
    vs the product's own materialised GEMM path (EOD_ATTN=gemm) on the same inputs"""
    from eo_diffusion_amd.backbones.unet_openai import AttentionBlock
    from oracle import unet_ref as UR
    from tests.synth import synth_input
    monkeypatch.setenv("EOD_PRECISION", "fp16")
    sd = synth_state_dict(attn_shapes(C), 4)
    x = synth_input(f"fa{C}{hw}", (1, C) + hw, 2)
    blk = AttentionBlock(C, num_heads=heads)
    load_into(blk, sd)
    with torch.no_grad():
        y = blk(x.to(DEV)).cpu()
        monkeypatch.setenv("EOD_ATTN", "gemm")
        y2 = blk(x.to(DEV)).cpu()
    assert rel_l2(y, y2) < 2e-3
    if hw[0] * hw[1] <= 1024:  # the oracle materialises [heads, T, T] on the CPU
        ref = UR.attention_block({"a." + k: v for k, v in sd.items()}, "a", x, heads, False)
        assert rel_l2(y, ref) < TOL["fp16"]


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("C,heads,hw,new", [(512, 1, (16, 16), False), (512, 1, (4, 4), False), (256, 2, (9, 7), True), (192, 1, (32, 32), False),
                                            (320, 1, (12, 12), True)])
def test_attention_block_wide_heads(prec, C, heads, hw, new, monkeypatch):
    """AttentionBlock with head dims above 64 -- the train.py:50 architecture's middle block is ONE head of 512 channels
    (unet_openai.py:675-681) at 16 x 16 (256 x 256 input) or 4 x 4 (64 x 64): fp16 / fp32x3 run the fused wide-head kernel
    (csrc/attn_wide.hip), fp32 the materialised path; vs the CPU oracle, and vs the product's own materialised path on the same inputs"""
    from eo_diffusion_amd.backbones.unet_openai import AttentionBlock
    from oracle import unet_ref as UR
    from tests.synth import synth_input
    monkeypatch.setenv("EOD_PRECISION", prec)
    sd = synth_state_dict(attn_shapes(C), 4)
    x = synth_input(f"wa{C}{hw}", (2, C) + hw, 2)
    x[1] *= 37.0  # (per-image operand scales)
    blk = AttentionBlock(C, num_heads=heads, use_new_attention_order=new)
    load_into(blk, sd)
    with torch.no_grad():
        y = blk(x.to(DEV)).cpu()
        monkeypatch.setenv("EOD_ATTN", "gemm")
        y2 = blk(x.to(DEV)).cpu()
    ref = UR.attention_block({"a." + k: v for k, v in sd.items()}, "a", x, heads, new)
    for n in range(2):
        assert rel_l2(y[n], ref[n]) < TOL[prec], n
        assert rel_l2(y[n], y2[n]) < 2 * TOL[prec], n


@pytest.mark.parametrize("prec", ["fp16", "fp32x3"])
@pytest.mark.parametrize("cls,heads,d,T", [("QKVAttentionLegacy", 1, 512, 256), ("QKVAttention", 2, 128, 77), ("QKVAttention", 1, 200, 16)])
def test_qkv_attention_standalone_wide_heads(prec, cls, heads, d, T, monkeypatch):
    """QKVAttention(Legacy).forward(qkv) on their own with head dims above 64 (fused, csrc/attn_wide.hip); the exact fp32 mode has no
    fused kernel for them and says so"""
    import eo_diffusion_amd.backbones.unet_openai as U
    from oracle import unet_ref as UR
    from tests.synth import synth_input
    monkeypatch.setenv("EOD_PRECISION", prec)
    qkv = synth_input(f"sw{cls}{T}", (2, 3 * heads * d, T), 6)
    ref = (UR.qkv_attention_legacy if cls == "QKVAttentionLegacy" else UR.qkv_attention_new)(qkv, heads)
    with torch.no_grad():
        out = getattr(U, cls)(heads)(qkv.to(DEV)).cpu()
    assert rel_l2(out, ref) < (2e-3 if prec == "fp16" else 5e-6)
    monkeypatch.setenv("EOD_PRECISION", "fp32")
    with pytest.raises(U._lib.EodError):
        getattr(U, cls)(heads)(qkv.to(DEV))


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("factory,size", [("UNetSmall", 32), ("UNet", 28), ("UNetBig", 32)])   # (UNetBig: base width 192 -- the first conv spans two N-tiles)
def test_factory_presets_vs_oracle(prec, factory, size):
    """UNetBig/UNet/UNetSmall presets (unet_openai.py:783-922): FiLM, resblock_updown, new attention order,
    num_head_channels, class conditioning, 3 attention resolutions; 28x28 exercises ragged maps (14x14, 7x7, T=49)."""
    import eo_diffusion_amd.backbones.unet_openai as U
    from oracle import unet_ref as UR
    from tests.synth import synth_input
    m = getattr(U, factory)(size, in_channels=3, out_channels=3, num_classes=4)
    m.dropout = 0.0
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth_state_dict(shapes, 7)
    m.load_state_dict(sd)
    m = m.set_precision(prec).to(DEV).eval()
    mults = {32: (1, 2, 2, 2), 28: (1, 2, 2, 2)}[size]
    res = "28,14,7" if size == 28 else "32,16,8"
    cfg = dict(model_channels=m.model_channels, num_res_blocks=m.num_res_blocks, channel_mult=mults,
               attention_resolutions=tuple(size // int(r) for r in res.split(",")), num_classes=4, num_heads=4,
               num_head_channels=m.num_head_channels, use_scale_shift_norm=True, resblock_updown=True,
               use_new_attention_order=True)
    x = synth_input("fx" + factory, (2, 3, size, size), 2)
    t = torch.tensor([5, 900])
    y = torch.tensor([0, 3])
    with torch.no_grad():
        out = m(x.to(DEV), t.to(DEV), y=y.to(DEV)).cpu()
    ref = UR.unet_forward(sd, cfg, x, t, y=y)
    assert rel_l2(out, ref) < (1e-2 if prec == "fp16" else 2e-5)


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("shape", [(1, 3, 16, 48), (3, 3, 40, 24)])
def test_unet_non_square_and_odd_batch(prec, shape):
    """the UNet itself is shape-agnostic (only EODiffusion assumes square images): non-square maps, batch 1 / 3"""
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    from oracle import unet_ref as UR
    from tests.synth import synth_input
    cfg = unet_cfgs()["u_a1_tiny"]
    sd = synth_state_dict(unet_param_shapes(**cfg), 7)
    u = UNetModel(**cfg).set_precision(prec)
    load_into(u, sd)
    x = synth_input(f"ns{shape}", shape, 2)
    t = torch.arange(shape[0]) * 7 + 1
    with torch.no_grad():
        out = u(x.to(DEV), t.to(DEV)).cpu()
    assert rel_l2(out, UR.unet_forward(sd, cfg, x, t)) < TOL[prec]


def test_timestep_embedding_standalone_vs_golden():
    """timestep_embedding (unet_openai.py:81-99) as its own call.
    (1) SAME-MACHINE check, gate 2e-6: the kernel against the oracle's own restatement evaluated on THIS host -- both take their
        frequency table exp(-ln(1e4) k / half) from this machine's torch CPU exp, so nothing machine-dependent is left between them
        (arguments reach 999 rad: this pins the range reduction of sin / cos).
    (2) against what the reference returned on the machine that wrote the fixture (t in {0, 1, 499, 999}, dims 32 / 128 / 33): torch's
        CPU exp differs in the last bit between CPU models (SLEEF AVX2 vs AVX-512 paths), one ulp of a frequency (6e-8 relative) moves
        the argument t*f by up to 1.2e-7 * t, and the sinusoid by as much: gate 2e-6 + 1.2e-7 * t, the derived bound."""
    from eo_diffusion_amd.backbones.unet_openai import timestep_embedding
    from oracle import unet_ref as UR
    g = gt("temb")
    for dim in (32, 128, 33):
        e = timestep_embedding(g["t"].to(DEV), dim).cpu()
        ref = g[f"d{dim}"]
        assert e.shape == ref.shape and e.dtype == torch.float32
        here = UR.timestep_embedding(g["t"], dim)  # the oracle on this host
        assert float((e - here).abs().max()) <= 2e-6, (dim, float((e - here).abs().max()))
        tol = 2e-6 + 1.2e-7 * g["t"].float()[:, None]  # values are in [-1, 1]: absolute = relative to the scale
        assert bool(((e - ref).abs() <= tol).all()), (dim, float((e - ref).abs().max()))
    tt = torch.tensor([0, 1, 7, 250, 499, 731, 998, 999])
    for dim in (64, 128):  # more timesteps on the same-machine gate
        e = timestep_embedding(tt.to(DEV), dim).cpu()
        assert float((e - UR.timestep_embedding(tt, dim)).abs().max()) <= 2e-6
    frac = timestep_embedding(torch.tensor([0.5, 10.25], device=DEV), 8).cpu()  # fractional timesteps are allowed
    assert bool(torch.isfinite(frac).all()) and float((frac[:, :4] ** 2 + frac[:, 4:] ** 2 - 1).abs().max()) < 1e-6


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("cls,heads,d,T", [("QKVAttentionLegacy", 4, 16, 49), ("QKVAttention", 8, 48, 256), ("QKVAttentionLegacy", 1, 64, 1024),
                                           ("QKVAttention", 2, 32, 200)])
def test_qkv_attention_standalone_vs_oracle(prec, cls, heads, d, T, monkeypatch):
    """QKVAttentionLegacy / QKVAttention .forward(qkv) (unet_openai.py:465-481, 497-515) called on their own, both channel orders,
    ragged lengths, vs the oracle restatement of the same two functions"""
    import eo_diffusion_amd.backbones.unet_openai as U
    from oracle import unet_ref as UR
    from tests.synth import synth_input
    monkeypatch.setenv("EOD_PRECISION", prec)
    qkv = synth_input(f"sa{cls}{T}", (2, 3 * heads * d, T), 6)
    ref = (UR.qkv_attention_legacy if cls == "QKVAttentionLegacy" else UR.qkv_attention_new)(qkv, heads)
    with torch.no_grad():
        out = getattr(U, cls)(heads)(qkv.to(DEV)).cpu()
    assert out.shape == ref.shape == (2, heads * d, T)
    assert rel_l2(out, ref) < (2e-3 if prec == "fp16" else 5e-6)
    with pytest.raises(U._lib.EodError):
        getattr(U, cls)(1)(torch.zeros(1, 3 * 20, 8, device=DEV))  # head dim 20: refused, not approximated


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
def test_train_mode_forward_without_grad_applies_dropout(prec):
    """train.py:149 samples previews from modules that may still be in train mode: under no_grad the inference program runs with
    nn.Dropout live (unet_openai.py:339).  The masks are Philox draws keyed by (seed, layer, forward counter): rebuilt here through
    the same entry point and injected into the oracle; a second forward draws new masks; eval mode has none."""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    from eo_diffusion_amd.engine import current_stream_ptr
    from oracle import unet_ref as UR
    from tests.synth import synth_input
    cfg = dict(unet_cfgs()["u_a1_tiny"], dropout=0.3)
    sd = synth_state_dict(unet_param_shapes(**cfg), 7)
    u = UNetModel(**cfg).set_precision(prec)
    u.load_state_dict(sd)
    u = u.to(DEV).train()
    S = cfg["image_size"]
    x, t = synth_input("do_x", (2, 3, S, S), 1), torch.tensor([4, 15])
    with torch.no_grad():
        out1 = u(x.to(DEV), t.to(DEV)).cpu()
        prog = u.program_for(2, 3, 0, S, S, torch.device(DEV), False)
        assert len(prog.drop_ops) == sum(1 for m in u.modules() if type(m).__name__ == "ResBlock") and prog.drop_step == 1
        masks = []
        for layer, idx in enumerate(prog.drop_ops):
            sm = prog._arr[idx].u.small
            ones = torch.ones((sm.l[0],), dtype=prog.tdtype, device=DEV)
            mk = torch.empty_like(ones)
            _lib.check(_lib.lib().eod_dropout(ones.data_ptr(), mk.data_ptr(), prog.dt, sm.l[0], 0.3, prog.drop_seed, layer, prog.drop_step,
                                              current_stream_ptr(torch.device(DEV))), "eod_dropout")
            masks.append(mk.float().cpu())
        out2 = u(x.to(DEV), t.to(DEV)).cpu()
        u.eval()
        out_eval = u(x.to(DEV), t.to(DEV)).cpu()
    keep = torch.cat([m.flatten() for m in masks])
    assert abs(float((keep > 0).float().mean()) - 0.7) < 0.02 and abs(float(keep.max()) - 1 / 0.7) < 1e-3
    order = iter(masks)
    drop = lambda pfx, h: next(order).reshape(h.shape[0], h.shape[2], h.shape[3], h.shape[1]).permute(0, 3, 1, 2).to(h.dtype)
    ref = UR.unet_forward(sd, {k: v for k, v in cfg.items() if k != "dropout"}, x, t, drop=drop)
    assert rel_l2(out1, ref) < TOL[prec]
    assert rel_l2(out2, out1) > 1e-2          # a new mask every forward
    assert rel_l2(out_eval, UR.unet_forward(sd, {k: v for k, v in cfg.items() if k != "dropout"}, x, t)) < TOL[prec]
