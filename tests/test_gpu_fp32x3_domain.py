"""GPU parity: the validity DOMAIN of the `fp32x3` precision mode (fp32 storage, products as three fp16 MFMAs on split operands).

The reference multiplies every conv / attention input in IEEE fp32 (unet_openai.py:352, 262-264, 227, 609, 414-422;
diffusion/model.py:101-122 lets x_t grow without bound under --no_clip), so the mode must hold its 1e-5 gate for inputs of any
magnitude and dynamic range -- in particular on the inputs that are NOT behind a GroupNorm: the first conv on x_t, the (fused)
1x1 skip convs over the raw block input, the stride-2 and upsample convs, proj_out, q / k / v.  Round 2 scaled activations by a
fixed 16: |x| >= 4094 gave inf / NaN and values of ~1e-4 lost their low halves.  Now every split consumer scales by a power of
two per image, derived on the device from the tensor's bound table (csrc/common.h).

Each case is checked against a float64 evaluation of the same op, gate 1e-5 -- never a silent NaN."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from eo_diffusion_amd import _lib
from eo_diffusion_amd.engine import Act, Program, current_stream_ptr, round_up
from tests.gpu_util import DEV
from tests.helpers import rel_l2
from tests.synth import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
GATE = 1e-5


def adversarial(kind, tag, shape, seed=71):
    """inputs with the dynamic ranges the fixed scale of round 2 could not represent (NCHW fp32)"""
    x = synth_input(f"dom{tag}{kind}", shape, seed)
    r = np.random.default_rng([seed, len(tag), shape[1]])
    if kind == "heavy":      # 1 % of the elements 1e4 times larger
        m = torch.from_numpy((r.random(shape) < 0.01).astype(np.float32))
        return x * (1.0 + (1e4 - 1.0) * m)
    if kind == "mixed":      # channels from 1e-6 to 1e3 inside one tensor
        mag = torch.from_numpy((10.0 ** r.uniform(-6, 3, size=(1, shape[1], 1, 1))).astype(np.float32))
        return x * mag
    if kind == "tiny":       # everything ~1e-6
        return x * 1e-6
    if kind == "huge":       # everything far beyond 4094 (what --no_clip sampling produces)
        return x * 1e5
    if kind == "outlier_image":  # ONE image carries a few elements 50 times larger: after a GroupNorm the images still need DIFFERENT scales
        m = torch.zeros(shape)
        m[1 % shape[0]].view(-1)[::97] = 1.0
        return x * (1.0 + 49.0 * m)
    if kind == "outlier_last_group":  # the largest values of ONE image sit in its last channels: the maximum of its bound table is entry 31
        m = torch.zeros(shape)
        m[1 % shape[0], -(shape[1] // 32):].view(-1)[::7] = 1.0
        return x * (1.0 + 5.0 * m)
    if kind == "per_image":  # images of one batch ten orders of magnitude apart: the scale is per image
        mag = torch.tensor([10.0 ** (5 - 10 * (i % 2)) for i in range(shape[0])]).view(-1, 1, 1, 1)
        return x * mag
    raise ValueError(kind)


KINDS = ["heavy", "mixed", "tiny", "huge", "per_image"]


def nhwc(prog, t):
    return prog.own(t.to(DEV).permute(0, 2, 3, 1).contiguous().to(prog.tdtype))


def run(emit_build, _unused=None):
    prog = Program(DEV, "fp32x3")
    y = emit_build(prog)
    prog.run()
    torch.cuda.synchronize()
    got = y.t.float().permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(got).all(), "inf / NaN out of the split product"
    return got, prog


def rel_per_image(got, ref):
    """worst image: an image of tiny values must not hide behind a large one in a batch-wide norm"""
    return max(rel_l2(got[i], ref[i]) for i in range(got.shape[0]))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("shape", [(2, 3, 32, 32, 128), (2, 13, 16, 48, 64)])
def test_first_conv_on_unbounded_input(kind, shape):
    """UNetModel.input_blocks[0] (unet_openai.py:609) on x_t: thin-input conv, K over the flattened [tap][channel] axis"""
    N, Cin, H, W, Cout = shape
    x = adversarial(kind, "first", (N, Cin, H, W))
    w = synth_input("domfw", (Cout, Cin, 3, 3), 71, scale=1.0 / math.sqrt(Cin * 9))
    b = synth_input("domfb", (Cout,), 71, scale=0.1)

    def build(prog):
        cp = round_up(Cin, prog.epc)
        a, idx = prog.to_nhwc(N, Cin, 0, H, W, cp)
        build.x = x.to(DEV).contiguous()
        prog.ops[idx].u.small.p[0] = build.x.data_ptr()
        y, i = prog.conv(a, prog.pack_conv_tapmajor(w.to(DEV), cp), prog.f32(b.to(DEV)), Cout, w_tapmajor=True)
        assert prog.ops[i].u.conv.w_split == 1 and prog.ops[i].u.conv.a_bound
        return y

    got, _ = run(build, None)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    assert rel_per_image(got, ref) < GATE


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("case", [
    # N, C (3x3 input), H, W, Cout, skip sources, GroupNorm fused into the 3x3 phase
    (2, 128, 16, 32, 128, (256, 128), True),   # the A0 up-path block: conv2 + skip over cat(256, 128)
    (2, 96, 16, 16, 192, (40,), True),          # tails in both phases
    (8, 96, 32, 64, 512, (40, 24), True),       # the 8-wave 256-column form
    (2, 128, 8, 16, 128, (64,), False),
])
def test_fused_skip_conv_over_unnormalised_block_input(kind, case):
    """`skip_connection(x) + h` (unet_openai.py:352, 385) in one launch: the 1x1 phase reads the RAW block input (one or two concat
    sources) -- adversarial here -- next to a normalised 3x3 phase; both feed one accumulator, so the launch runs on the smaller of
    the two per-image scales"""
    N, C, H, W, Cout, scs, gn = case
    h = synth_input(f"domsh{case}", (N, C, H, W), 72)
    xs = [adversarial(kind, f"sk{i}", (N, sc, H, W), 72 + i) for i, sc in enumerate(scs)]
    w3 = synth_input(f"domsw3{case}", (Cout, C, 3, 3), 72, scale=1.0 / math.sqrt(C * 9))
    b3 = synth_input("domsb3", (Cout,), 72, scale=0.1)
    xmag = float(torch.cat(xs, 1).abs().mean())
    w1 = synth_input(f"domsw1{case}", (Cout, sum(scs), 1, 1), 73, scale=1.0 / math.sqrt(sum(scs)))
    b1 = synth_input("domsb1", (Cout,), 73, scale=0.1)
    gam = 1.0 + 0.2 * synth_input("domsg", (C,), 72)
    bet = 0.1 * synth_input("domsbt", (C,), 72)

    def build(prog):
        ah = Act(nhwc(prog, h), N, H, W, C)
        axs = [Act(nhwc(prog, t), N, H, W, t.shape[1]) for t in xs]
        if not prog.conv_skip_ok(ah, Cout, axs):
            pytest.skip("fused skip conv switched off")
        g = None
        if gn:
            g = (prog.gn_stats([ah], prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV))), True)
        if sum(scs) % 32 == 0:
            # the production route (ResBlock._emit): in_layers' GroupNorm has taken statistics of exactly these tensors, and its finalize
            # wrote the raw bound table the skip phase uses; the other cases take the fallback (one direct max|x| pass per source)
            prog.gn_stats(axs, prog.f32(torch.ones(sum(scs)).to(DEV)), prog.f32(torch.zeros(sum(scs)).to(DEV)))
        y, i = prog.conv(ah, prog.pack_conv(w3.to(DEV)), prog.f32(b3.to(DEV)), Cout, gn=g, skip=(axs, w1.to(DEV), b1.to(DEV)))
        d = prog.ops[i].u.conv
        assert d.w_split == 1 and d.a_bound and d.skip_bound
        return y

    got, _ = run(build, None)
    hd = h.double()
    if gn:
        hd = F.silu(F.group_norm(hd, 32, gam.double(), bet.double(), eps=1e-5))
    ref = F.conv2d(hd, w3.double(), b3.double(), padding=1) + F.conv2d(torch.cat(xs, 1).double(), w1.double(), b1.double())
    assert rel_per_image(got, ref) < GATE


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("case", [
    # N, Cin, H, W, Cout, ksize, stride, two sources
    (2, 160, 16, 16, 128, 1, 1, False),    # stand-alone 1x1 skip conv
    (2, 384, 32, 32, 128, 1, 1, True),     # ... over a virtual concat (256 | 128)
    (1, 256, 24, 16, 768, 1, 1, False),    # qkv-shaped 1x1, the 8-wave 256-column form
    (2, 128, 32, 32, 128, 3, 2, False),    # Downsample.op (unet_openai.py:262-264)
    (2, 128, 16, 32, 512, 3, 2, False),
    (3, 32, 7, 7, 64, 3, 2, False),        # odd map, tiles straddle images: per-ROW scales inside one tile
    (4, 64, 4, 4, 64, 1, 1, False),        # 16-pixel images, 8 per tile (split-K route)
    (5, 96, 8, 8, 96, 3, 1, False),        # 3x3 on an 8x8 map (generic kernel, split-K), 2 images per tile
])
def test_generic_split_kernel_on_unnormalised_input(kind, case):
    """1x1 and stride-2 convs (and 3x3 convs on maps too small for the halo kernel) run on the generic split kernel, which re-splits its
    pixel rows every K-step; a tile of 128 rows can span several images, each with its own scale"""
    N, Cin, H, W, Cout, k, stride, two = case
    x = adversarial(kind, f"gen{case}", (N, Cin, H, W), 74)
    w = synth_input(f"domgw{case}", (Cout, Cin, k, k), 74, scale=1.0 / math.sqrt(Cin * k * k))
    b = synth_input("domgb", (Cout,), 74, scale=0.1)

    def build(prog):
        if two:
            c0 = Cin * 2 // 3
            a, a2 = Act(nhwc(prog, x[:, :c0]), N, H, W, c0), Act(nhwc(prog, x[:, c0:]), N, H, W, Cin - c0)
        else:
            a, a2 = Act(nhwc(prog, x), N, H, W, Cin), None
        y, i = prog.conv(a, prog.pack_conv(w.to(DEV)), prog.f32(b.to(DEV)), Cout, x2=a2, ksize=k, stride=stride, pad=k // 2)
        assert prog.ops[i].u.conv.w_split == 1 and prog.ops[i].u.conv.a_bound
        return y

    got, _ = run(build, None)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=k // 2)
    assert rel_per_image(got, ref) < GATE


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("form", ["up4", "ups"])
@pytest.mark.parametrize("case", [(2, 128, 8, 16, 128), (1, 96, 16, 32, 192)])
def test_upsample_conv_on_unnormalised_input(kind, form, case):
    """Upsample.conv (unet_openai.py:227, 236-241) reads the raw block output: parity-class form (conv_up4_halo_kernel) and the nine-tap
    form on the virtual nearest-2x image"""
    N, Cin, H, W, Cout = case
    x = adversarial(kind, f"up{case}", (N, Cin, H, W), 75)
    w = synth_input(f"domuw{case}", (Cout, Cin, 3, 3), 75, scale=1.0 / math.sqrt(Cin * 9))
    b = synth_input("domub", (Cout,), 75, scale=0.1)

    def build(prog):
        a = Act(nhwc(prog, x), N, H, W, Cin)
        if form == "up4":
            if not prog.conv_up4_ok(a, Cout):
                pytest.skip("parity-class form switched off")
            y, i = prog.conv(a, prog.pack_conv_up4(w.to(DEV)), prog.f32(b.to(DEV)), Cout, upsample="up4")
        else:
            y, i = prog.conv(a, prog.pack_conv(w.to(DEV)), prog.f32(b.to(DEV)), Cout, upsample=True)
        assert prog.ops[i].u.conv.w_split == 1 and prog.ops[i].u.conv.a_bound
        return y

    got, _ = run(build, None)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
    assert rel_per_image(got, ref) < GATE


@pytest.mark.parametrize("kind", KINDS)
def test_bounds_from_conv_epilogue_statistics_feed_the_next_conv(kind):
    """the production route: conv A (epilogue statistics) -> its raw output feeds a stride-2 conv and an upsample conv whose scales
    come from the sums of squares A's epilogue wrote (eod_act_bound, parts mode: no pass over the tensor)"""
    N, C, H, W = 2, 128, 16, 32
    x = adversarial(kind, "chain", (N, C, H, W), 76)
    wa = synth_input("domcwa", (C, C, 3, 3), 76, scale=1.0 / math.sqrt(C * 9))
    wd = synth_input("domcwd", (C, C, 3, 3), 77, scale=1.0 / math.sqrt(C * 9))
    wu = synth_input("domcwu", (C, C, 3, 3), 78, scale=1.0 / math.sqrt(C * 9))
    outs = {}

    def build(prog):
        a = Act(nhwc(prog, x), N, H, W, C)
        ya, _ = prog.conv(a, prog.pack_conv(wa.to(DEV)), None, C, stats=True)
        assert ya.stats is not None
        yd, i = prog.conv(ya, prog.pack_conv(wd.to(DEV)), None, C, stride=2)
        n_ops = len(prog.ops)
        yu, j = prog.conv(ya, prog.pack_conv_up4(wu.to(DEV)), None, C, upsample="up4") if prog.conv_up4_ok(ya, C) else \
            prog.conv(ya, prog.pack_conv(wu.to(DEV)), None, C, upsample=True)
        assert len(prog.ops) == n_ops + 1, "the second consumer reuses the first one's bound table"
        assert prog.ops[i].u.conv.a_bound == prog.ops[j].u.conv.a_bound != 0
        outs["d"], outs["u"] = yd, yu
        return ya

    got_a, _ = run(build, None)
    ra = F.conv2d(x.double(), wa.double(), None, padding=1)
    assert rel_per_image(got_a, ra) < GATE
    ga = got_a.double()  # (the consumers see the stored fp32 tensor)
    gd = outs["d"].t.float().permute(0, 3, 1, 2).cpu()
    gu = outs["u"].t.float().permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(gd).all() and torch.isfinite(gu).all()
    assert rel_per_image(gd, F.conv2d(ga, wd.double(), None, stride=2, padding=1)) < GATE
    assert rel_per_image(gu, F.conv2d(F.interpolate(ga, scale_factor=2, mode="nearest"), wu.double(), None, padding=1)) < GATE


@pytest.mark.parametrize("kind", ["heavy", "mixed", "tiny", "huge"])
@pytest.mark.parametrize("T,heads,d", [(256, 2, 48), (1000, 1, 64)])
def test_attention_qkv_of_any_magnitude(kind, T, heads, d):
    """q / k / v (unet_openai.py:476-480) of any magnitude through the fused fp32-storage attention: v carries the adversarial
    range (q, k are kept at logits a softmax can resolve: beyond |S| ~ 1e3 the fp32 softmax of the reference is itself one-hot noise),
    plus -- `huge` -- all three scaled by 300 (logits of ~1e6: one-hot rows), against float64"""
    L = _lib.lib()
    N, C = 2, heads * d
    qkv = synth_input(f"domat{T}{d}", (N, T, 3 * C), 79, scale=0.8)
    hs = 3 * d
    v_cols = torch.cat([torch.arange(2 * d + h * hs, 3 * d + h * hs) for h in range(heads)])
    if kind == "huge":
        qkv = qkv * 300.0
    else:
        adv = adversarial(kind, f"at{T}", (N, C, T, 1), 79).squeeze(-1).permute(0, 2, 1)  # [N][T][C]
        qkv[:, :, v_cols] = adv
    pick = lambda off: torch.stack([qkv.double()[:, :, off + h * hs: off + h * hs + d] for h in range(heads)], 1)
    q, k, v = pick(0), pick(d), pick(2 * d)
    S = q @ k.transpose(-1, -2) / math.sqrt(d)
    ref = (torch.softmax(S, -1) @ v).permute(0, 2, 1, 3).reshape(N, T, C)
    f32 = rel_l2((torch.softmax(S.float(), -1) @ v.float()).permute(0, 2, 1, 3).reshape(N, T, C), ref)
    qd = qkv.to(DEV)
    out = torch.full((N, T, C), 9.0, dtype=torch.float32, device=DEV)
    ab = torch.zeros((N, 32), dtype=torch.float32, device=DEV)
    st = current_stream_ptr(torch.device(DEV))
    _lib.check(L.eod_act_bound(qd.data_ptr(), _lib.EOD_F32, N, T * 3 * C, 0, 0, 0, 0, 0, 0, ab.data_ptr(), 0, st), "act_bound")
    _lib.check(L.eod_attention_fwd_nat(qd.data_ptr(), out.data_ptr(), 0, _lib.EOD_F32, N, T, C, heads, d, 0, d, 2 * d, hs, ab.data_ptr(), 0, st),
               "attention_fwd_nat")
    torch.cuda.synchronize()
    got = out.cpu()
    assert torch.isfinite(got).all()
    err = max(rel_l2(got[i], ref[i]) for i in range(N))
    print(f"{kind} T={T} d={d}: fused fp32x3 attention {err:.2e}, torch fp32 {f32:.2e}")
    assert err < max(GATE, 4 * f32)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("cfg", ["attn_d32", "attn_wide"])
def test_attention_block_on_unnormalised_residual_stream(kind, cfg, monkeypatch):
    """AttentionBlock (unet_openai.py:427-433) as a module: the residual stream x is adversarial; norm() brings the qkv input back to
    O(1), but q / k / v, the attention output in front of proj_out and the residual add all see whatever the weights make of it.
    attn_d32: fused kernel (head dim 32); attn_wide: one head of 128 channels -> the split-fp16 GEMM path with per-image tables"""
    from eo_diffusion_amd.backbones.unet_openai import AttentionBlock
    monkeypatch.setenv("EOD_PRECISION", "fp32x3")
    C, heads = (128, 4) if cfg == "attn_d32" else (128, 1)
    blk = AttentionBlock(C, num_heads=heads)
    shapes = {"norm.weight": (C,), "norm.bias": (C,), "qkv.weight": (3 * C, C, 1), "qkv.bias": (3 * C,),
              "proj_out.weight": (C, C, 1), "proj_out.bias": (C,)}
    sd = synth_state_dict(shapes, 4)
    blk.load_state_dict(sd)
    blk = blk.to(DEV).eval()
    x = adversarial(kind, f"ab{cfg}", (2, C, 16, 16), 80)
    with torch.no_grad():
        y = blk(x.to(DEV)).cpu()
    assert torch.isfinite(y).all()
    # float64 evaluation of unet_openai.py:427-433 + 465-481 (the oracle's own functions round to fp32 inside, like the reference)
    sd64 = {k: v.double() for k, v in sd.items()}
    xd = x.double()
    hn = F.group_norm(xd, 32, sd64["norm.weight"], sd64["norm.bias"], eps=1e-5).reshape(2, C, -1)
    qkv = F.conv1d(hn, sd64["qkv.weight"], sd64["qkv.bias"])
    d = C // heads
    q, k, v = qkv.reshape(2 * heads, 3 * d, -1).split(d, dim=1)
    sc = 1 / math.sqrt(math.sqrt(d))
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", q * sc, k * sc), -1)
    a = torch.einsum("bts,bcs->bct", wgt, v).reshape(2, C, -1)
    ref = (xd.reshape(2, C, -1) + F.conv1d(a, sd64["proj_out.weight"], sd64["proj_out.bias"])).reshape(x.shape)
    assert rel_per_image(y, ref) < GATE


def test_bound_table_of_a_groupnorm_finalize_bounds_both_tensors():
    """eod_gn_finalize's two tables: ab_raw >= max|x| and ab_norm >= max|x*scale + shift| per image (and neither is wildly loose)"""
    L = _lib.lib()
    N, C, H, W = 3, 96, 16, 24
    x = adversarial("mixed", "gnb", (N, C, H, W), 81)
    gam = 1.0 + 0.2 * synth_input("gnbg", (C,), 81)
    bet = 0.1 * synth_input("gnbb", (C,), 81)
    prog = Program(DEV, "fp32x3")
    a = Act(nhwc(prog, x), N, H, W, C)
    ss = prog.gn_stats([a], prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV)))
    prog.run()
    torch.cuda.synchronize()
    raw = a.bound.cpu().max(1).values
    nrm = ss.eod_bound_norm.cpu().max(1).values
    true_raw = x.abs().amax((1, 2, 3))
    true_nrm = F.group_norm(x, 32, gam, bet, eps=1e-5).abs().amax((1, 2, 3))
    assert (raw >= true_raw).all() and (raw <= 64 * true_raw).all(), (raw, true_raw)
    assert (nrm >= true_nrm).all() and (nrm <= 256 * true_nrm).all(), (nrm, true_nrm)


# ------------------------------------------------------------------------------------------------ whole UNet / sampler
@pytest.mark.parametrize("kind", ["huge", "heavy", "tiny", "per_image"])
def test_unet_forward_on_x_t_of_any_magnitude(kind):
    """UNetModel.forward (unet_openai.py:746-780) at the BASELINE architecture (A0: base 128, mults [1,2,3,4]; 64 x 64, batch 2) on an
    x_t the fixed scale of round 2 could not take: every kernel of the bench step runs (first conv, GroupNorm-fused halo convs with the
    1x1 skip inside, stride-2 convs, parity-class upsample convs, the one-head middle attention on the split-fp16 GEMMs, the head)"""
    from tests.test_gpu_baseline_sizes import _eps_fn, _unet
    x = adversarial(kind, "unet", (2, 3, 64, 64), 82)
    t = torch.tensor([999, 3])
    with torch.no_grad():
        ref = _eps_fn("A0", 64)(x, t)
        out = _unet("A0", 64, "fp32x3").to(DEV).eval()(x.to(DEV), t.to(DEV)).cpu()
    assert torch.isfinite(out).all()
    err = rel_per_image(out, ref)
    print(f"A0@64 forward on {kind} x_t (max|x| = {float(x.abs().max()):.3g}): rel-L2 = {err:.3e}")
    assert err < GATE


def test_no_clip_sampling_whose_x_t_exceeds_4094():
    """EODiffusion.sampling(clipped_reverse_diffusion=False) (diffusion/model.py:101-122, inference.py --no_clip): without the x0 clamp
    x_t is multiplied by 1/sqrt(alpha_t) every step (31.6 at t = T-1 of the cosine schedule) and nothing pulls it back; started from
    300 x N(0,1) every x_t of the 20-step trajectory lies far beyond 4094 (asserted on the oracle's own states).  fp32x3 follows the
    fp32 CPU oracle inside the trajectory gate of the exact mode."""
    from eo_diffusion_amd.diffusion.model import EODiffusion
    from oracle import sampler_ref as SR
    from oracle import schedule as SCH
    from tests.test_gpu_baseline_sizes import _eps_fn, _unet
    T = 20
    xT = 300.0 * synth_input("ncl_xT", (2, 3, 64, 64), 83)
    noises = [synth_input(f"ncl_n{k}", (2, 3, 64, 64), 84) for k in range(T)]
    states = []
    with torch.no_grad():
        ref = SR.ddpm_sampling(SCH.eo_cosine_tables(T), _eps_fn("A0", 64), xT, noises, T, clip=False, record=states)
    assert all(float(s.abs().max()) > 4094.0 for s in states), [float(s.abs().max()) for s in states]
    m = EODiffusion(_unet("A0", 64, "fp32x3"), timesteps=T, image_size=64, in_channels=3, device=DEV).to(DEV).eval()
    out = m.sampling(2, clipped_reverse_diffusion=False, device=DEV, x_T=xT, noises=noises, progress=False).cpu()
    assert torch.isfinite(out).all()
    err = rel_per_image(out, ref)
    print(f"20 no-clip DDPM steps, max|x_t| {min(float(s.abs().max()) for s in states):.3g} .. {max(float(s.abs().max()) for s in states):.3g}: rel-L2 = {err:.3e}")
    assert err < 2e-5


# ------------------------------------------------------------------------------------------------ pre-split producers
@pytest.mark.parametrize("kind", ["heavy", "mixed", "huge", "per_image", "outlier_image", "outlier_last_group"])
@pytest.mark.parametrize("case", [(2, 128, 16, 16, 384), (1, 384, 64, 64, 1152), (3, 96, 7, 9, 40),
                                  (3, 288, 6, 6, 864), (3, 288, 6, 6, 288), (2, 192, 12, 12, 576),   # (small maps, K >= 8 steps: split-K, tiles across images)
                                  (2, 288, 16, 16, 96), (2, 96, 8, 8, 96), (1, 160, 16, 32, 64)])   # (channel counts whose pass runs in blocks of a partial last wave)
def test_groupnorm_written_presplit_feeds_a_1x1_conv(kind, case):
    """AttentionBlock.norm -> qkv (unet_openai.py:427-428, 414): the normalising pass writes its output PRE-SPLIT ([8 x hi | 8 x lo] per
    8 channels, scaled per image from the finalize's bound table) and the 1x1 conv DMAs those rows straight into its LDS image
    (eod_conv_desc.x_presplit) -- same result as the float64 GroupNorm + conv, for a residual stream of any magnitude"""
    N, C, H, W, Cout = case
    x = adversarial(kind, f"ps{case}", (N, C, H, W), 85)
    w = synth_input(f"dompw{case}", (Cout, C, 1, 1), 85, scale=1.0 / math.sqrt(C))
    b = synth_input("dompb", (Cout,), 85, scale=0.1)
    gam = 1.0 + 0.2 * synth_input("dompg", (C,), 85)
    bet = 0.1 * synth_input("dompbt", (C,), 85)

    def build(prog):
        a = Act(nhwc(prog, x), N, H, W, C)
        xn = prog.group_norm([a], prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV)), silu=False, split_out=True)
        assert xn.presplit and xn.bound is not None
        y, i = prog.conv(xn, prog.pack_conv(w.to(DEV)), prog.f32(b.to(DEV)), Cout, ksize=1, stride=1, pad=0)
        d = prog.ops[i].u.conv
        assert d.w_split == 1 and d.x_presplit == 1 and d.a_bound
        return y

    got, _ = run(build)
    ref = F.conv2d(F.group_norm(x.double(), 32, gam.double(), bet.double(), eps=1e-5), w.double(), b.double())
    assert rel_per_image(got, ref) < GATE


# ------------------------------------------------------------------------------------------------ weights: output rows of any relative magnitude
def _row_magnitudes(cout, span, seed):
    """output-channel magnitudes over `span` decades (a trained net's dead channels next to live ones): the split weights carry a scale per
    ROW (csrc/misc.hip: row_exp_kernel), so every output channel keeps its own 22 bits; with one scale per tensor a channel 1e8 times
    smaller than the largest one was off by 1e-3 and one 1e12 times smaller was lost -- visible as soon as a GroupNorm renormalises it"""
    g = torch.Generator().manual_seed(seed)
    mag = 10.0 ** (torch.rand(cout, generator=g) * span - span / 2)
    mag[0], mag[-1] = 10.0 ** (span / 2), 10.0 ** (-span / 2)
    return mag


def worst_channel(got, ref):
    d, r = (got.double() - ref).flatten(2).norm(dim=2), ref.flatten(2).norm(dim=2)
    return float((d / r.clamp_min(1e-300)).max())


@pytest.mark.parametrize("case", [
    # kind, N, Cin, H, W, Cout
    ("halo", 2, 128, 16, 32, 128), ("halo_gn", 2, 128, 16, 32, 256), ("halo_gn", 2, 96, 16, 16, 384), ("halo", 1, 64, 8, 16, 64),
    ("skip", 2, 128, 16, 32, 128), ("1x1", 2, 160, 16, 16, 128), ("1x1", 1, 256, 24, 16, 768), ("s2", 2, 128, 32, 32, 128),
    ("splitk", 5, 96, 8, 8, 96), ("up4", 2, 128, 8, 16, 128), ("ups", 1, 96, 16, 32, 192), ("first", 2, 3, 32, 32, 128),
    ("head", 2, 128, 16, 32, 3),
])
def test_output_channels_of_any_relative_magnitude(case):
    kind, N, Cin, H, W, Cout = case
    span = 12
    mag = _row_magnitudes(Cout, span, 7)
    k = 1 if kind == "1x1" else 3
    x = synth_input(f"wr{case}", (N, Cin, H, W), 81)
    w = synth_input(f"wrw{case}", (Cout, Cin, k, k), 82, scale=1.0 / math.sqrt(Cin * k * k)) * mag[:, None, None, None]
    b = synth_input("wrb", (Cout,), 83, scale=0.1) * mag
    gam, bet = 1.0 + 0.2 * synth_input("wrg", (Cin,), 84), 0.1 * synth_input("wrbt", (Cin,), 85)
    xs = synth_input(f"wrs{case}", (N, 64, H, W), 86)
    w1 = synth_input(f"wrw1{case}", (Cout, 64, 1, 1), 87, scale=0.125) * mag[:, None, None, None]
    out = {}

    def build(prog):
        a = Act(nhwc(prog, x), N, H, W, Cin)
        bb = prog.f32(b.to(DEV))
        if kind in ("halo", "splitk"):
            y, i = prog.conv(a, prog.pack_conv(w.to(DEV)), bb, Cout)
        elif kind == "halo_gn":
            g = (prog.gn_stats([a], prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV))), True)
            y, i = prog.conv(a, prog.pack_conv(w.to(DEV)), bb, Cout, gn=g)
        elif kind == "skip":
            ax = [Act(nhwc(prog, xs), N, H, W, 64)]
            if not prog.conv_skip_ok(a, Cout, ax):
                pytest.skip("fused skip conv switched off")
            y, i = prog.conv(a, prog.pack_conv(w.to(DEV)), bb, Cout, skip=(ax, w1.to(DEV), None))
        elif kind == "1x1":
            y, i = prog.conv(a, prog.pack_conv(w.to(DEV)), bb, Cout, ksize=1, stride=1, pad=0)
        elif kind == "s2":
            y, i = prog.conv(a, prog.pack_conv(w.to(DEV)), bb, Cout, ksize=3, stride=2, pad=1)
        elif kind == "up4":
            if not prog.conv_up4_ok(a, Cout):
                pytest.skip("parity-class form switched off")
            y, i = prog.conv(a, prog.pack_conv_up4(w.to(DEV)), bb, Cout, upsample="up4")
        elif kind == "ups":
            y, i = prog.conv(a, prog.pack_conv(w.to(DEV)), bb, Cout, upsample=True)
        elif kind == "first":
            cp = round_up(Cin, prog.epc)
            a, idx = prog.to_nhwc(N, Cin, 0, H, W, cp)
            build.x = x.to(DEV).contiguous()
            prog.ops[idx].u.small.p[0] = build.x.data_ptr()
            y, i = prog.conv(a, prog.pack_conv_tapmajor(w.to(DEV), cp), bb, Cout, w_tapmajor=True)
        elif kind == "head":
            g = (prog.gn_stats([a], prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV))), True)
            out["y"] = torch.empty((N, Cout, H, W), dtype=torch.float32, device=DEV)
            _, i = prog.conv(a, prog.pack_conv(w.to(DEV)), bb, Cout, out_nchw_f32=True, gn=g)
            prog.ops[i].u.conv.y = out["y"].data_ptr()
            y = None
        assert prog.ops[i].u.conv.w_split == 1
        return y

    if kind == "head":
        prog = Program(DEV, "fp32x3")
        build(prog)
        prog.run()
        torch.cuda.synchronize()
        got = out["y"].cpu()
    else:
        got, _ = run(build, None)
    xd = x.double()
    if kind in ("halo_gn", "head"):
        xd = F.silu(F.group_norm(xd, 32, gam.double(), bet.double(), eps=1e-5))
    if kind in ("up4", "ups"):
        xd = F.interpolate(xd, scale_factor=2, mode="nearest")
    ref = F.conv2d(xd, w.double(), b.double(), stride=2 if kind == "s2" else 1, padding=k // 2)
    if kind == "skip":
        ref = ref + F.conv2d(xs.double(), w1.double())
    e = worst_channel(got, ref)
    print(f"{case}: worst output channel over {span} decades of row magnitudes: {e:.2e}")
    assert torch.isfinite(got).all() and e < GATE


@pytest.mark.parametrize("i", range(6))
def test_unet_with_dead_and_loud_channels_in_front_of_every_groupnorm(i):
    """whole random UNets (tests/test_gpu_fuzz_archs.py) whose first conv and every ResBlock in_layers conv -- the convs a GroupNorm
    renormalises channel group by channel group -- have output rows spread over 12 decades (bias with them): with one scale per weight
    TENSOR the small rows came out of the split product with few or no bits and the next GroupNorm blew that up; with a scale per row
    the fp32x3 result sits where the exact-fp32 mode's does (3e-7 ... 1.5e-6 against the CPU oracle)"""
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    from oracle import unet_ref as UR
    from tests.test_gpu_fuzz_archs import _random_cfg
    cfg, N, H, W, in_ch, cond_ch = _random_cfg(i)
    sd = synth_state_dict(unet_param_shapes(**cfg), 40 + i)
    g = torch.Generator().manual_seed(i)
    for k, v in list(sd.items()):
        if k.endswith("weight") and v.dim() == 4 and ("in_layers.2" in k or k.startswith("input_blocks.0.0")):
            mag = 10.0 ** (torch.rand(v.shape[0], generator=g) * 12 - 6)
            sd[k] = v * mag.view(-1, 1, 1, 1)
            sd[k[:-6] + "bias"] = sd[k[:-6] + "bias"] * mag
    x = synth_input(f"fz_x{i}", (N, in_ch, H, W), 41 + i)
    cond = synth_input(f"fz_c{i}", (N, cond_ch, H, W), 42 + i) if cond_ch else None
    t = torch.tensor([(37 * (i + 1) * (k + 1)) % 1000 for k in range(N)])
    y = torch.tensor([(i + k) % 5 for k in range(N)]) if "num_classes" in cfg else None
    with torch.no_grad():
        ref = UR.unet_forward(sd, cfg, x, t, cond=cond, y=y)
        assert torch.isfinite(ref).all()
        u = UNetModel(**cfg).set_precision("fp32x3")
        u.load_state_dict(sd)
        u = u.to(DEV).eval()
        out = u(x.to(DEV), t.to(DEV), cond=cond.to(DEV) if cond is not None else None, y=y.to(DEV) if y is not None else None).cpu()
    assert torch.isfinite(out).all() and rel_l2(out, ref) < GATE, (i, rel_l2(out, ref))


def test_split_weight_pack_keeps_22_bits_of_every_row():
    """the packed image itself (eod_pack_conv_weight_split): [tap][Cout][Cin/8][8 x hi | 8 x lo] fp16 of s * 2^d_j * w with the scale buffer
    {s, 1/(16 s), -, -, d_0 ... d_{Cout-1}}: decoded, every row reproduces its fp32 weights to 2^-21 of the ROW's maximum over 30 decades of
    row magnitudes, zero rows stay zero (d = 0), and every row's largest |s_j w| lies in (2^12, 2^13]"""
    Cout, Cin, k = 96, 40, 3
    g = torch.Generator().manual_seed(3)
    mag = 10.0 ** (torch.rand(Cout, generator=g) * 30 - 15)
    w = synth_input("pkw", (Cout, Cin, k, k), 5) * mag[:, None, None, None]
    w[7] = 0.0
    prog = Program(DEV, "fp32x3")
    dst, scale = prog.pack_conv(w.to(DEV)).split()
    torch.cuda.synchronize()
    s = float(scale[0])
    d = scale.view(torch.int32)[_lib.WSCALE_ROWS:_lib.WSCALE_ROWS + Cout].cpu()
    assert float(scale[1]) == 1.0 / (16.0 * s) and int(d[7]) == 0 and int(d.min()) == 0 and int(d.max()) <= 100
    img = dst.view(torch.float16).view(k * k, Cout, Cin // 8, 2, 8).float().cpu()     # [tap][co][group][hi|lo][8]
    rec = (img[:, :, :, 0] + img[:, :, :, 1]).reshape(k * k, Cout, Cin).permute(1, 2, 0).reshape(Cout, Cin, k, k).double()
    sj = s * torch.pow(torch.tensor(2.0, dtype=torch.float64), d.double())
    top = (w.double().abs().flatten(1).amax(1) * sj)
    live = top > 0
    assert bool(((top[live] > 2.0 ** 12) & (top[live] <= 2.0 ** 13)).all())
    err = ((rec / sj[:, None, None, None] - w.double()).abs().flatten(1).amax(1) / w.double().abs().flatten(1).amax(1).clamp_min(1e-300))
    assert float(err[live].max()) < 2.0 ** -21 and float(rec[7].abs().max()) == 0.0
