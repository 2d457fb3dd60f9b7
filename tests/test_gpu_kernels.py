"""GPU parity: the MFMA implicit-GEMM conv / batched GEMM, GroupNorm and softmax kernels, called through the
C ABI (engine.Program -> eod_program_run), against plain torch fp32 CPU ops of the same math."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from eo_diffusion_amd.engine import Program
from tests.gpu_util import DEV, TOL, run_program
from tests.helpers import rel_l2
from tests.synth import synth_input

pytestmark = pytest.mark.gpu

CONV_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad, upsample
    (2, 32, 8, 8, 32, 3, 1, 1, False),
    (2, 64, 16, 16, 128, 3, 1, 1, False),
    (1, 128, 32, 32, 256, 3, 1, 1, False),      # patch-mode tiles, 2 N-tiles
    (2, 96, 7, 7, 64, 3, 1, 1, False),          # ragged spatial (linear tiles), K tail (96 % 64)
    (3, 32, 28, 28, 64, 3, 2, 1, False),        # stride 2
    (2, 32, 7, 7, 32, 3, 2, 1, False),          # stride 2, odd
    (2, 32, 8, 8, 32, 3, 1, 1, True),           # virtual nearest-2x
    (2, 32, 3, 3, 32, 3, 1, 1, True),           # 3x3 -> 7x7 pad hack
    (2, 160, 8, 8, 96, 1, 1, 0, False),         # 1x1
    (1, 128, 16, 16, 3, 3, 1, 1, False),        # tiny Cout (BN=32 tiles)
    (1, 8, 16, 16, 128, 3, 1, 1, False),        # tiny Cin (first conv, padded)
    (1, 384, 16, 16, 384, 3, 1, 1, False),
    (2, 96, 16, 32, 128, 3, 1, 1, False),       # halo-patch kernel, K tail (96 % 64), non-square map
    (3, 128, 8, 16, 192, 3, 1, 1, False),       # halo-patch kernel, one patch per image, N tail (192 % 128)
    (1, 640, 32, 32, 256, 3, 1, 1, False),      # halo-patch kernel, 10 channel chunks
    (2, 128, 8, 8, 128, 3, 1, 1, True),         # upsample halo-patch kernel (6x10 input patch), 16x16 output
    (1, 96, 16, 24, 192, 3, 1, 1, True),        # upsample halo, K tail, N tail, non-square
    (2, 96, 16, 16, 256, 1, 1, 0, False),       # 1x1 to 256 columns (fp32x3: the 8-wave 256-column form of the generic kernel), K tail
    (1, 256, 24, 16, 768, 1, 1, 0, False),      # qkv-shaped 1x1: three 256-column tiles
    (2, 128, 16, 32, 512, 3, 2, 1, False),      # stride 2 to 512 columns
]


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_vs_torch(prec, case):
    N, Cin, H, W, Cout, k, stride, pad, ups = case
    x = synth_input(f"cx{case}", (N, Cin, H, W), 31)
    w = synth_input(f"cw{case}", (Cout, Cin, k, k), 31, scale=1.0 / math.sqrt(Cin * k * k))
    b = synth_input(f"cb{case}", (Cout,), 31, scale=0.1)

    def emit(prog, a):
        y, _ = prog.conv(a, prog.pack_conv(w.to(DEV)), prog.f32(b.to(DEV)), Cout, ksize=k, stride=stride, pad=pad,
                         upsample=ups, pad_tl=(ups and H == 3 and W == 3))
        return y

    got = run_program(prec, x, emit)
    xin = x
    if ups:
        xin = F.interpolate(x, scale_factor=2, mode="nearest")
        if H == 3 and W == 3:
            xin = F.pad(xin, (1, 0, 1, 0))
    ref = F.conv2d(xin, w, b, stride=stride, padding=pad)
    assert got.shape == ref.shape
    assert rel_l2(got, ref) < TOL[prec]


@pytest.mark.parametrize("prec", ["fp16", "fp32x3"])
@pytest.mark.parametrize("gn", [False, True])
@pytest.mark.parametrize("case", [
    # N, C (3x3 input), H, W, Cout, skip channels (one source, or two = virtual concat)
    (2, 128, 8, 16, 128, (64,)),          # one K-step of skip (fp16), two (fp32 storage)
    (1, 128, 16, 32, 128, (256, 128)),    # the A0 up-path block: conv2 128->128 + skip over cat(256, 128)
    (2, 96, 16, 16, 192, (40,)),          # channel tails in both phases (96 % 64, 40 % 32), N tail (192 % 128)
    (1, 256, 24, 16, 256, (72, 200)),     # two N tiles; tails in both skip sources
    (3, 160, 8, 16, 136, (320,)),         # one patch per image, Cout % 128 = 8
    (8, 64, 64, 64, 384, (72,)),          # enough tiles for the two-launch form of 384 columns (256 on 8 waves + 128 on 4)
    (8, 96, 32, 64, 512, (40, 24)),       # ... and for the 8-wave 256-column form (2 x 256)
])
def test_conv3x3_with_fused_1x1_skip_vs_torch(prec, gn, case):
    """eod_conv_desc.skip_x (conv3x3_halo_kernel<SKIP>): y = conv3x3([GroupNorm+SiLU](h)) + conv1x1(cat(x...)) + both biases, the ResBlock
    tail `skip_connection(x) + h` (unet_openai.py:352, 385) in one launch, vs the two F.conv2d of the reference; fp32x3: the two
    weights share one split scale (eod_pack_conv_weight_split_pair) -- the skip weight is 8x the 3x3 one here to exercise that"""
    from eo_diffusion_amd.engine import Act
    N, C, H, W, Cout, scs = case
    h = synth_input(f"skh{case}", (N, C, H, W), 47, scale=1.3) + 0.1
    xs = [synth_input(f"skx{case}{i}", (N, c, H, W), 47) for i, c in enumerate(scs)]
    w3 = synth_input(f"skw3{case}", (Cout, C, 3, 3), 47, scale=1.0 / math.sqrt(C * 9))
    w1 = synth_input(f"skw1{case}", (Cout, sum(scs), 1, 1), 47, scale=8.0 / math.sqrt(C * 9))
    b3 = synth_input(f"skb3{case}", (Cout,), 47, scale=0.1)
    b1 = synth_input(f"skb1{case}", (Cout,), 47, scale=0.1)
    gam = 1.0 + 0.2 * synth_input("skg", (C,), 47)
    bet = 0.1 * synth_input("ske", (C,), 47)
    prog = Program(DEV, prec)
    to_act = lambda t: Act(prog.own(t.to(DEV).permute(0, 2, 3, 1).contiguous().to(prog.tdtype)), t.shape[0], H, W, t.shape[1])
    ah, axs = to_act(h), [to_act(t) for t in xs]
    if not prog.conv_skip_ok(ah, Cout, axs):
        pytest.skip("fused skip conv not available in this configuration (EOD_SKIP_FUSE=0)")
    g = (prog.gn_stats([ah], prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV))), True) if gn else None
    y, _ = prog.conv(ah, prog.pack_conv(w3.to(DEV)), prog.f32(b3.to(DEV)), Cout, gn=g, stats=True, skip=(axs, w1.to(DEV), b1.to(DEV)))
    prog.run()
    torch.cuda.synchronize()
    got = y.t.float().permute(0, 3, 1, 2).cpu()
    hin = F.silu(F.group_norm(h, 32, gam, bet, eps=1e-5)) if gn else h
    if prec == "fp16":  # the oracle sees the stored (fp16-rounded) operands of the skip path
        xs = [t.half().float() for t in xs]
    ref = F.conv2d(hin, w3, b3, padding=1) + F.conv2d(torch.cat(xs, 1), w1, b1)
    assert rel_l2(got, ref) < TOL[prec]
    if y.stats is not None:  # the epilogue's GroupNorm partial sums cover the fused result
        st, slots = y.stats
        tot = st.sum(1).cpu()  # [N][Cout][2]
        assert torch.allclose(tot[..., 0], got.sum((2, 3)), rtol=2e-3, atol=2e-2)


def _random_halo_cases(n, seed):
    import random
    rnd = random.Random(seed)
    out = []
    for _ in range(n):
        N = rnd.choice([1, 2, 3])
        H, W = 8 * rnd.choice([1, 2, 3, 5]), 16 * rnd.choice([1, 2, 3])
        C = 8 * rnd.choice([4, 8, 12, 16, 20, 33, 48])
        Cout = rnd.choice([128, 256, 512, 136, 384, 200, 256])
        nsrc = rnd.choice([0, 1, 2])
        scs = tuple(8 * rnd.choice([1, 4, 5, 8, 16, 23, 40]) for _ in range(nsrc))
        out.append((N, C, H, W, Cout, scs, rnd.random() < 0.7))
    return out


@pytest.mark.parametrize("prec", ["fp16", "fp32x3"])
@pytest.mark.parametrize("case", _random_halo_cases(int(os.environ.get("EOD_TEST_RANDOM_CASES", "14")),
                                                    int(os.environ.get("EOD_TEST_RANDOM_SEED", "20260101"))))  # (soak runs: more cases / other seeds)
def test_halo_conv_random_shapes_vs_torch(prec, case):
    """seeded random geometries through every instance of the halo kernel's fast paths: 4-wave / 8-wave (256-column) tiles, GroupNorm
    fused or not, with or without the fused 1x1 skip conv over one or two sources, channel and column tails, several images"""
    from eo_diffusion_amd.engine import Act
    N, C, H, W, Cout, scs, gn = case
    h = synth_input(f"rh{case}", (N, C, H, W), 53, scale=1.2) - 0.1
    xs = [synth_input(f"rx{case}{i}", (N, c, H, W), 53) for i, c in enumerate(scs)]
    w3 = synth_input(f"rw3{case}", (Cout, C, 3, 3), 53, scale=1.0 / math.sqrt(C * 9))
    b3 = synth_input(f"rb3{case}", (Cout,), 53, scale=0.1)
    gam = 1.0 + 0.2 * synth_input("rg", (C,), 53)
    bet = 0.1 * synth_input("re", (C,), 53)
    prog = Program(DEV, prec)
    to_act = lambda t: Act(prog.own(t.to(DEV).permute(0, 2, 3, 1).contiguous().to(prog.tdtype)), t.shape[0], H, W, t.shape[1])
    ah, axs = to_act(h), [to_act(t) for t in xs]
    gn = gn and C % 32 == 0
    g = (prog.gn_stats([ah], prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV))), True) if gn else None
    hin = F.silu(F.group_norm(h, 32, gam, bet, eps=1e-5)) if gn else h
    ref = F.conv2d(hin if prec != "fp16" or gn else h.half().float(), w3, b3, padding=1)
    skip = None
    if xs and prog.conv_skip_ok(ah, Cout, axs):
        w1 = synth_input(f"rw1{case}", (Cout, sum(scs), 1, 1), 53, scale=1.0 / math.sqrt(sum(scs)))
        b1 = synth_input(f"rb1{case}", (Cout,), 53, scale=0.1)
        skip = (axs, w1.to(DEV), b1.to(DEV))
        ref = ref + F.conv2d(torch.cat([t.half().float() if prec == "fp16" else t for t in xs], 1), w1, b1)
    y, _ = prog.conv(ah, prog.pack_conv(w3.to(DEV)), prog.f32(b3.to(DEV)), Cout, gn=g, stats=True, skip=skip)
    prog.run()
    torch.cuda.synchronize()
    got = y.t.float().permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(got).all()
    assert rel_l2(got, ref) < TOL[prec]


STREAM_CASES = [  # N, C0, C1, H, W, Cout, gn : 3x3 convs on the streaming halo instances (128- and 256-column tiles), K tails, several images
    (2, 128, 0, 32, 32, 128, True), (1, 96, 64, 24, 48, 256, True), (3, 64, 0, 16, 32, 128, False), (2, 40, 0, 32, 16, 384, False),
    (1, 32, 0, 40, 16, 128, True), (2, 256, 0, 8, 32, 512, True),
]


@pytest.mark.parametrize("prec", ["fp32x3"])
@pytest.mark.parametrize("case", STREAM_CASES)
def test_halo_stream_runs_are_bit_identical_to_single_tiles(prec, case):
    """conv3x3_halo_kernel's STREAM form (a workgroup runs several pixel tiles as one stream of chunks: halo_tpw option): outputs and
    GroupNorm partial sums must be the bits of the one-tile-per-workgroup launch for every run length, with the residual, the per-sample
    bias and a fused input GroupNorm across a concat seam in play, and stay inside the mode's gate against torch."""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import Act
    N, C0, C1, H, W, Cout, gn = case
    xs = [synth_input(f"sx0{case}", (N, C0, H, W), 59, scale=1.5) + 0.3] + ([synth_input(f"sx1{case}", (N, C1, H, W), 59) - 0.2] if C1 else [])
    C = C0 + C1
    w = synth_input(f"sw{case}", (Cout, C, 3, 3), 59, scale=1.0 / math.sqrt(C * 9))
    b = synth_input(f"sb{case}", (Cout,), 59, scale=0.1)
    res = synth_input(f"sr{case}", (N, Cout, H, W), 59)
    temb = synth_input(f"st{case}", (N, Cout), 59, scale=0.3)
    gam = 1.0 + 0.2 * synth_input("sg", (C,), 59)
    bet = 0.1 * synth_input("se", (C,), 59)
    L = _lib.lib()

    def run(tpw):
        prev = L.eod_set_option(b"halo_tpw", tpw)
        try:
            prog = Program(DEV, prec)
            to_act = lambda t: Act(prog.own(t.to(DEV).permute(0, 2, 3, 1).contiguous().to(prog.tdtype)), t.shape[0], H, W, t.shape[1])
            srcs = [to_act(t) for t in xs]
            g = (prog.gn_stats(srcs, prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV))), True) if gn and C % 32 == 0 else None
            y, _ = prog.conv(srcs[0], prog.pack_conv(w.to(DEV)), prog.f32(b.to(DEV)), Cout, x2=srcs[1] if C1 else None, gn=g, stats=True,
                             res=to_act(res), cbias=prog.f32(temb.to(DEV)), cbias_stride=Cout)
            prog.run()
            torch.cuda.synchronize()
            return y.t.clone(), y.stats[0].clone()
        finally:
            L.eod_set_option(b"halo_tpw", prev)

    y1, s1 = run(1)
    for tpw in (2, 3, 4, 8, 0):
        yt, st = run(tpw)
        assert torch.equal(yt, y1) and torch.equal(st, s1), f"run length {tpw}"
    hin = torch.cat(xs, 1)
    if gn and C % 32 == 0:
        hin = F.silu(F.group_norm(hin, 32, gam, bet, eps=1e-5))
    ref = F.conv2d(hin, w, b, padding=1) + temb[:, :, None, None] + res
    assert rel_l2(y1.float().permute(0, 3, 1, 2).cpu(), ref) < TOL[prec]


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("case", [  # N, C0, C1, H, W, Cout, gn, skip channels
    (3, 256, 0, 16, 16, 384, True, 0), (2, 512, 384, 16, 16, 384, True, 0), (2, 384, 0, 16, 16, 384, True, 256), (1, 640, 0, 8, 32, 256, False, 0),
    (2, 256, 0, 16, 16, 256, True, 512), (3, 136, 0, 16, 16, 200, False, 0),
    # 8-wide maps: one image row per tile row, the right half of the 8 x 16 tile masked
    (3, 512, 0, 8, 8, 512, True, 0), (2, 512, 512, 8, 8, 512, True, 0), (2, 384, 0, 8, 8, 512, True, 384), (2, 256, 0, 16, 8, 384, False, 0),
])
def test_halo_conv_split_in_k_vs_torch_and_batch_invariant(prec, case):
    """3x3 convs on maps with fewer than two workgroups per CU run the halo kernel with its channel chunks split over gridDim.y workgroups
    (conv3x3_halo_kernel, p.splitk) + the reduce pass: against torch, against the unsplit arm (halo_splitk = 0), with the statistics of
    the reduce pass, and every image's bits independent of the batch it is computed in (the factor is a function of the per-image geometry)"""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import Act
    N, C0, C1, H, W, Cout, gn, sc = case
    C = C0 + C1
    xs = [synth_input(f"kx0{case}", (N, C0, H, W), 61, scale=1.3) + 0.2] + ([synth_input(f"kx1{case}", (N, C1, H, W), 61) - 0.1] if C1 else [])
    xk = synth_input(f"kxs{case}", (N, max(sc, 8), H, W), 61)
    w = synth_input(f"kw{case}", (Cout, C, 3, 3), 61, scale=1.0 / math.sqrt(C * 9))
    b = synth_input(f"kb{case}", (Cout,), 61, scale=0.1)
    w1 = synth_input(f"kw1{case}", (Cout, max(sc, 8), 1, 1), 61, scale=1.0 / math.sqrt(max(sc, 8)))
    res = synth_input(f"kr{case}", (N, Cout, H, W), 61)
    gam, bet = 1.0 + 0.2 * synth_input("kg", (C,), 61), 0.1 * synth_input("ke", (C,), 61)
    L = _lib.lib()

    def run(n, splitk_on):
        prev = L.eod_set_option(b"halo_splitk", int(splitk_on))
        try:
            prog = Program(DEV, prec)
            to_act = lambda t: Act(prog.own(t[:n].to(DEV).permute(0, 2, 3, 1).contiguous().to(prog.tdtype)), n, H, W, t.shape[1])
            srcs = [to_act(t) for t in xs]
            g = (prog.gn_stats(srcs, prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV))), True) if gn and C % 32 == 0 else None
            skip = None
            if sc and prec != "fp32":
                ax = [to_act(xk)]
                if prog.conv_skip_ok(srcs[0], Cout, ax):
                    skip = (ax, w1.to(DEV), None)
            y, i = prog.conv(srcs[0], prog.pack_conv(w.to(DEV)), prog.f32(b.to(DEV)), Cout, x2=srcs[1] if C1 else None, gn=g, stats=True,
                             res=None if skip else to_act(res), skip=skip)
            used = bool(prog.ops[i].u.conv.workspace)
            prog.run()
            torch.cuda.synchronize()
            return y.t.clone(), y.stats[0].clone(), skip is not None, used
        finally:
            L.eod_set_option(b"halo_splitk", prev)

    y3, s3, skipped, used = run(N, True)
    # (fp16 storage has 64-channel chunks: 136 channels are three chunks, and fewer than four stay unsplit; csrc/igemm.hip: conv_splitk)
    assert used == ((prec, C0) != ("fp16", 136)), "which arm this geometry runs on"
    y0, _, _, used0 = run(N, False)
    assert not used0
    y1, s1, _, _ = run(1, True)
    assert torch.equal(y1[0], y3[0]) and torch.equal(s1[0], s3[0]), "image 0 differs between batch 1 and the full batch"
    hin = torch.cat(xs, 1)
    if gn and C % 32 == 0:
        hin = F.silu(F.group_norm(hin, 32, gam, bet, eps=1e-5))
    elif prec == "fp16":
        hin = hin.half().float()
    ref = F.conv2d(hin, w, b, padding=1)
    ref = ref + (F.conv2d(xk.half().float() if prec == "fp16" else xk, w1) if skipped else (res.half().float() if prec == "fp16" else res))
    got = y3.float().permute(0, 3, 1, 2).cpu()
    assert rel_l2(got, ref) < TOL[prec] and rel_l2(y0.float().permute(0, 3, 1, 2).cpu(), ref) < TOL[prec]
    tot = s3.sum(1).cpu()  # [N][Cout][2]: the reduce pass's sums of the stored values
    assert torch.allclose(tot[..., 0], got.sum((2, 3)), rtol=2e-3, atol=2e-2)


@pytest.mark.parametrize("prec", ["fp16", "fp32x3"])
@pytest.mark.parametrize("gn", [False, True])
@pytest.mark.parametrize("case", [(2, 128, 8, 16, 128), (1, 96, 16, 32, 192), (3, 64, 24, 16, 256), (1, 640, 16, 16, 160),
                                  (2, 512, 8, 8, 512), (3, 72, 16, 8, 160)])   # (8-wide stored maps: the 8 x 8 level of a 64 x 64 image; half of every tile masked)
def test_conv_nearest_upsample_parity_class_form_vs_torch(prec, gn, case):
    """upsample='up4' (conv_up4_halo_kernel): the 3x3 conv over the nearest-2x image as four 2x2-tap parity classes with pre-summed
    weights (4/9 of the MACs) vs F.interpolate + F.conv2d: one / several patches per image, K tail (96 % 64), N tail, 10 chunks;
    gn: a GroupNorm behind it consumes the statistics accumulated in the conv's epilogue (one slot per (tile, class, wave row))"""
    N, Cin, H, W, Cout = case
    x = synth_input(f"u4x{case}", (N, Cin, H, W), 43)
    w = synth_input(f"u4w{case}", (Cout, Cin, 3, 3), 43, scale=1.0 / math.sqrt(Cin * 9))
    b = synth_input(f"u4b{case}", (Cout,), 43, scale=0.1)
    gam = 1.0 + 0.2 * synth_input("u4g", (Cout,), 43)
    bet = 0.1 * synth_input("u4e", (Cout,), 43)

    def emit(prog, a):
        if os.environ.get("EOD_UP4") == "0":
            pytest.skip("the parity-class kernels are switched off by the environment (A/B run)")
        assert prog.conv_up4_ok(a, Cout)
        y, _ = prog.conv(a, prog.pack_conv_up4(w.to(DEV)), prog.f32(b.to(DEV)), Cout, ksize=3, stride=1, pad=1, upsample="up4", stats=True)
        assert y.stats is not None and y.stats[1] == (H // 8) * ((W + 15) // 16) * 8
        return prog.group_norm([y], prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV)), silu=False) if gn else y

    got = run_program(prec, x, emit)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
    if gn:
        ref = F.group_norm(ref, 32, gam, bet, eps=1e-5)
    assert got.shape == ref.shape
    assert rel_l2(got, ref) < (TOL[prec] if not gn else (2e-5 if prec == "fp32x3" else 3e-3))


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
@pytest.mark.parametrize("case", [
    (2, 64, 16, 16, 128, True),    # four parity-class launches, patch-mode tiles on the quarter grid, residual
    (1, 96, 12, 20, 192, False),   # raster tiles (20 % 16), K tail, N tail
    (3, 128, 32, 32, 128, True),   # tiles of one class span whole images
    (2, 32, 4, 4, 32, False),      # small map: split-K route (single full-grid launch that multiplies the inserted zeros)
])
def test_conv_zero_insertion_upsample_vs_torch(prec, case):
    """upsample=2 (zero insertion: the backward-data of a stride-2 conv, unet_openai.py:107-125 Downsample): the conv over the
    zero-stuffed (2H x 2W) map, computed per output parity class with only the taps that meet stored samples"""
    N, Cin, H, W, Cout, with_res = case
    x = synth_input(f"zx{case}", (N, Cin, H, W), 41)
    w = synth_input(f"zw{case}", (Cout, Cin, 3, 3), 41, scale=1.0 / math.sqrt(Cin * 9 / 4))
    r = synth_input(f"zr{case}", (N, Cout, 2 * H, 2 * W), 41)

    def emit(prog, a):
        from eo_diffusion_amd.engine import Act
        rr = Act(prog.own(r.to(DEV).permute(0, 2, 3, 1).contiguous().to(prog.tdtype)), N, 2 * H, 2 * W, Cout) if with_res else None
        y, _ = prog.conv(a, prog.pack_conv(w.to(DEV)), None, Cout, ksize=3, stride=1, pad=1, upsample=2, res=rr)
        return y

    got = run_program(prec, x, emit)
    xin = torch.zeros(N, Cin, 2 * H, 2 * W)
    xin[:, :, ::2, ::2] = x
    ref = F.conv2d(xin, w, None, padding=1) + (r if with_res else 0)
    assert got.shape == ref.shape
    assert rel_l2(got, ref) < TOL[prec]


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("dims", [(2, 64, 32, 8, 8, 64), (2, 64, 32, 16, 16, 128), (1, 128, 96, 32, 16, 256)])
def test_conv_fused_epilogue_concat_residual_temb(prec, dims):
    """two A sources (virtual concat), per-sample bias (timestep embedding) and residual in one launch
    (second / third shape: the halo-patch kernel with a K tail in the second source)"""
    N, C0, C1, H, W, Cout = dims
    x0 = synth_input("fx0", (N, C0, H, W), 32)
    x1 = synth_input("fx1", (N, C1, H, W), 32)
    r = synth_input("fr", (N, Cout, H, W), 32)
    w = synth_input("fw", (Cout, C0 + C1, 3, 3), 32, scale=0.03)
    b = synth_input("fb", (Cout,), 32, scale=0.1)
    te = synth_input("fte", (N, Cout + 5), 32)

    def emit(prog, a):
        from eo_diffusion_amd.engine import Act
        # two NHWC sources with separate storage (test-side layout plumbing)
        t0 = prog.own(x0.to(DEV).permute(0, 2, 3, 1).contiguous().to(prog.tdtype))
        t1 = prog.own(x1.to(DEV).permute(0, 2, 3, 1).contiguous().to(prog.tdtype))
        rr = prog.own(r.to(DEV).permute(0, 2, 3, 1).contiguous().to(prog.tdtype))
        tem = prog.own(te.to(DEV))
        y, _ = prog.conv(Act(t0, N, H, W, C0), prog.pack_conv(w.to(DEV)), prog.f32(b.to(DEV)), Cout,
                         x2=Act(t1, N, H, W, C1), cbias=tem[:, 5:], cbias_stride=Cout + 5, res=Act(rr, N, H, W, Cout))
        return y

    got = run_program(prec, torch.cat([x0, x1], 1), emit)
    ref = F.conv2d(torch.cat([x0, x1], 1), w, b, padding=1) + te[:, 5:, None, None] + r
    assert rel_l2(got, ref) < TOL[prec]


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("shape", [(2, 32, 8, 8), (2, 96, 7, 7), (1, 128, 32, 32), (3, 224, 5, 3), (2, 1024, 4, 4),
                                   (2, 1280, 6, 5), (1, 2560, 3, 3)])  # (the last two: more than 256 16-byte chunks per pixel -> channel blocks)
@pytest.mark.parametrize("silu", [True, False])
def test_group_norm_silu(prec, shape, silu):
    x = synth_input(f"gx{shape}", shape, 33, scale=2.0) + 0.5
    C = shape[1]
    gam = 1.0 + 0.2 * synth_input("gg", (C,), 33)
    bet = 0.1 * synth_input("gb", (C,), 33)

    def emit(prog, a):
        return prog.group_norm([a], prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV)), silu=silu)

    got = run_program(prec, x, emit)
    ref = F.group_norm(x, 32, gam, bet, eps=1e-5)
    if silu:
        ref = F.silu(ref)
    assert rel_l2(got, ref) < (2e-3 if prec == "fp16" else 2e-6)


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
def test_group_norm_across_concat_seam(prec):
    """GN over a virtual concat whose groups straddle the seam: 64 | 32 channels -> 3 per group"""
    from eo_diffusion_amd.engine import Act
    N, C0, C1, H, W = 2, 64, 32, 6, 6
    x0 = synth_input("sx0", (N, C0, H, W), 34, scale=3.0)
    x1 = synth_input("sx1", (N, C1, H, W), 34) - 1.0
    gam = 1.0 + 0.2 * synth_input("sg", (C0 + C1,), 34)
    bet = 0.1 * synth_input("sb", (C0 + C1,), 34)

    def emit(prog, a):
        t0 = prog.own(x0.to(DEV).permute(0, 2, 3, 1).contiguous().to(prog.tdtype))
        t1 = prog.own(x1.to(DEV).permute(0, 2, 3, 1).contiguous().to(prog.tdtype))
        return prog.group_norm([Act(t0, N, H, W, C0), Act(t1, N, H, W, C1)], prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV)), silu=True)

    got = run_program(prec, torch.cat([x0, x1], 1), emit)
    ref = F.silu(F.group_norm(torch.cat([x0, x1], 1), 32, gam, bet, eps=1e-5))
    assert rel_l2(got, ref) < (2e-3 if prec == "fp16" else 2e-6)


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])   # fp32x3: both operands split into fp16 pairs in LDS (eod_gemm_desc.x3)
@pytest.mark.parametrize("mnk", [(128, 128, 64), (200, 72, 48), (49, 49, 128), (1024, 96, 512), (33, 3, 16)])
def test_gemm_nt_batched(prec, mnk):
    from eo_diffusion_amd.engine import Program
    M, Nn, K = mnk
    nb0, nb1 = 2, 3
    a = synth_input(f"ga{mnk}", (nb0, nb1, M, K), 35)
    b = synth_input(f"gb{mnk}", (nb0, nb1, Nn, K), 35)
    bias = synth_input(f"gbias{mnk}", (Nn,), 35)
    prog = Program(DEV, prec)
    ad, bd = prog.own(a.to(DEV).to(prog.tdtype)), prog.own(b.to(DEV).to(prog.tdtype))
    c = prog.empty((nb0, nb1, M, Nn), torch.float32)
    prog.gemm(ad, bd, c, M, Nn, K, K, K, Nn, bias=prog.f32(bias.to(DEV)), bias_mode=1, alpha=0.5, c_f32=True, nb0=nb0,
              nb1=nb1, sa=(nb1 * M * K, M * K), sb=(nb1 * Nn * K, Nn * K), sc=(nb1 * M * Nn, M * Nn))
    prog.run()
    torch.cuda.synchronize()
    ref = 0.5 * torch.einsum("xymk,xynk->xymn", a, b) + bias
    assert rel_l2(c.cpu(), ref) < TOL[prec]


def test_softmax_rows_pad():
    from eo_diffusion_amd.engine import Program
    prog = Program(DEV, "fp32")
    s = synth_input("sm", (37, 56), 36, scale=4.0)
    sd = prog.own(s.to(DEV))
    p = prog.empty((37, 56), torch.float32)
    prog.softmax_rows(sd, 56, p, 56, 37, 49)
    prog.run()
    torch.cuda.synchronize()
    ref = torch.softmax(s[:, :49], -1)
    out = p.cpu()
    assert rel_l2(out[:, :49], ref) < 1e-6 and float(out[:, 49:].abs().max()) == 0.0


def test_bad_arguments_fail_loudly():
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import Program
    prog = Program(DEV, "fp16")
    a = prog.act(1, 4, 4, 12)  # 12 channels: not a multiple of 8 halves
    w = prog.empty((9, 8, 12))
    prog.conv(a, w, None, 8)
    with pytest.raises(_lib.EodError, match="multiples of 8"):
        prog.run()


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("dims", [(2, 64, 32, 16, 16, 128), (1, 128, 0, 32, 16, 256), (2, 96, 0, 8, 16, 192), (2, 32, 0, 8, 8, 32)])
@pytest.mark.parametrize("silu", [True, False])
def test_conv_with_fused_input_groupnorm(prec, dims, silu):
    """GroupNorm(+SiLU) of the conv input applied inside the halo-patch kernel (last shape: not fusable -> the engine
    falls back to the separate apply pass); concat seam, K tail, zero padding must stay zero AFTER normalisation"""
    from eo_diffusion_amd.engine import Act
    N, C0, C1, H, W, Cout = dims
    x0 = synth_input("nx0", (N, C0, H, W), 37, scale=2.0) + 0.7
    xs = [x0] + ([synth_input("nx1", (N, C1, H, W), 37) - 0.3] if C1 else [])
    C = C0 + C1
    gam = 1.0 + 0.2 * synth_input("ng", (C,), 37)
    bet = 0.1 * synth_input("nb", (C,), 37)
    w = synth_input("nw", (Cout, C, 3, 3), 37, scale=0.05)
    b = synth_input("nbias", (Cout,), 37, scale=0.1)

    def emit(prog, a):
        srcs = [Act(prog.own(t.to(DEV).permute(0, 2, 3, 1).contiguous().to(prog.tdtype)), N, H, W, t.shape[1]) for t in xs]
        ss = prog.gn_stats(srcs, prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV)))
        y, _ = prog.conv(srcs[0], prog.pack_conv(w.to(DEV)), prog.f32(b.to(DEV)), Cout, x2=srcs[1] if C1 else None, gn=(ss, silu))
        return y

    got = run_program(prec, torch.cat(xs, 1), emit)
    hn = F.group_norm(torch.cat(xs, 1), 32, gam, bet, eps=1e-5)
    if silu:
        hn = F.silu(hn)
    ref = F.conv2d(hn, w, b, padding=1)
    assert rel_l2(got, ref) < TOL[prec]


@pytest.mark.parametrize("mag", [1.0, 1e-2, 1e2, 1e-4, 1e-6, 1e4, 1e8, 1e-12, 3e18])
@pytest.mark.parametrize("wmag", [1.0, 1e-3, 30.0])
def test_fp32x3_product_is_fp32_grade_at_any_magnitude(mag, wmag):
    """the split-fp16 product (three fp16 MFMAs per product, per-tensor power-of-two weight scale, per-IMAGE power-of-two activation
    scale derived on the device from the tensor's bound table, fp16 subnormals for the low halves) against a float64 convolution: as
    accurate as the exact-fp32 MFMA path for activations of ANY magnitude (1e-12 ... 3e18 here; round 2's fixed activation scale of 16
    overflowed above 4094 and lost the low halves below ~1e-4) and weights from 1e-3 to 30 -- not just inside the 1e-5 gate."""
    N, Cin, H, W, Cout = 1, 256, 32, 32, 128
    x = synth_input("x3x", (N, Cin, H, W), 51, scale=mag)
    w = synth_input("x3w", (Cout, Cin, 3, 3), 51, scale=wmag / math.sqrt(Cin * 9))
    b = synth_input("x3b", (Cout,), 51, scale=0.1 * mag * wmag)

    def emit(prog, a):
        y, idx = prog.conv(a, prog.pack_conv(w.to(DEV)), prog.f32(b.to(DEV)), Cout)
        emit.split = prog.ops[idx].u.conv.w_split
        return y

    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    err = {}
    for prec in ("fp32", "fp32x3"):
        err[prec] = rel_l2(run_program(prec, x, emit), ref)
        assert emit.split == (1 if prec == "fp32x3" else 0)  # the split kernel really ran
    print(f"activations ~{mag:g}, weights ~{wmag:g}: exact fp32 {err['fp32']:.2e}, fp32x3 {err['fp32x3']:.2e}")
    assert err["fp32x3"] < 1e-6 and err["fp32x3"] < 4 * err["fp32"] + 2e-7


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("dims", [(2, 3, 16, 16, 128), (1, 7, 32, 16, 128), (2, 13, 8, 24, 64), (1, 4, 7, 9, 32), (1, 3, 64, 64, 128),
                                  (2, 3, 16, 16, 192), (1, 13, 16, 16, 320), (2, 1, 8, 8, 160)])   # (more than one 128-column tile: UNetBig's base width is 192)
def test_first_conv_tapmajor(prec, dims):
    """thin-input 3x3 conv (UNet input conv, unet_openai.py:609): K over the flattened [tap][channel] axis
    (eod_conv_desc.w_tapmajor); image channels zero-padded to whole 16-byte chunks; borders, ragged maps, N tails"""
    N, Cin, H, W, Cout = dims
    x = synth_input(f"tx{dims}", (N, Cin, H, W), 41)
    w = synth_input(f"tw{dims}", (Cout, Cin, 3, 3), 41, scale=1.0 / math.sqrt(Cin * 9))
    b = synth_input(f"tb{dims}", (Cout,), 41, scale=0.1)

    gam = 1.0 + 0.2 * synth_input("tg", (Cout,), 41)
    bet = 0.1 * synth_input("te", (Cout,), 41)
    for gn in (False, True):  # gn: a GroupNorm behind the conv consumes the partial sums of its epilogue (the slots of whichever kernel ran)
        applied = []

        def emit(prog, a):
            if (a.C // prog.epc) not in (1, 2, 4):
                pytest.skip("channel padding outside the tap-major range for this precision")
            y, _ = prog.conv(a, prog.pack_conv_tapmajor(w.to(DEV), a.C), prog.f32(b.to(DEV)), Cout, w_tapmajor=True, stats=True)
            if gn and Cout % 32 == 0:
                applied.append(True)
                return prog.group_norm([y], prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV)), silu=False)
            return y

        got = run_program(prec, x, emit)
        ref = F.conv2d(x, w, b, padding=1)
        if applied:
            assert rel_l2(got, F.group_norm(ref, 32, gam, bet, eps=1e-5)) < (2e-5 if prec != "fp16" else 3e-3)
        else:
            assert rel_l2(got, ref) < TOL[prec]


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("dims", [(2, 128, 16, 16, 3), (1, 64, 8, 32, 13), (2, 96, 16, 16, 3), (3, 128, 40, 48, 3), (1, 160, 24, 32, 16),
                                  (1, 32, 8, 16, 1), (1, 416, 8, 16, 3)])
@pytest.mark.parametrize("fused_gn", [True, False])
def test_head_conv_nchw_f32(prec, dims, fused_gn):
    """output head (unet_openai.py:739-742): [GroupNorm + SiLU ->] 3x3 conv to a few channels written as NCHW fp32: conv_head_kernel
    for the fused form (weights in registers, three-deep patch ring; single and many chunks, channel tails, multi-image, up to 16
    output channels), the 32-column instance of the halo-patch kernel otherwise (no GroupNorm, > 384 input channels, exact fp32)"""
    from eo_diffusion_amd.engine import Act
    N, C, H, W, Cout = dims
    x = synth_input(f"hx{dims}", (N, C, H, W), 43, scale=1.5) + 0.2
    gam = 1.0 + 0.2 * synth_input("hg", (C,), 43)
    bet = 0.1 * synth_input("hb", (C,), 43)
    w = synth_input(f"hw{dims}", (Cout, C, 3, 3), 43, scale=0.05)
    b = synth_input(f"hbias{dims}", (Cout,), 43, scale=0.1)
    prog = Program(DEV, prec)
    a = Act(prog.own(x.to(DEV).permute(0, 2, 3, 1).contiguous().to(prog.tdtype)), N, H, W, C)
    out = torch.empty((N, Cout, H, W), dtype=torch.float32, device=DEV)
    gn = None
    if fused_gn:
        gn = (prog.gn_stats([a], prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV))), True)
    _, idx = prog.conv(a, prog.pack_conv(w.to(DEV)), prog.f32(b.to(DEV)), Cout, out_nchw_f32=True, gn=gn)
    prog.ops[idx].u.conv.y = out.data_ptr()
    prog.run()
    torch.cuda.synchronize()
    hin = F.silu(F.group_norm(x, 32, gam, bet, eps=1e-5)) if fused_gn else x
    ref = F.conv2d(hin, w, b, padding=1)
    assert rel_l2(out.cpu(), ref) < TOL[prec]


@pytest.mark.parametrize("prec", ["fp16", "fp32x3"])
@pytest.mark.parametrize("dims,tpws", [((2, 128, 40, 48, 3), (3, 5, 15)), ((1, 96, 32, 64, 3), (2, 4, 16)), ((3, 160, 8, 32, 16), (2,))])
def test_head_conv_tile_streams_are_bit_identical_to_single_tiles(prec, dims, tpws):
    """conv_head_kernel's chunk stream (head_tpw: a workgroup walks a run of consecutive tiles of one image through one chunk ring; the
    default picks runs only on maps with >= 512 tiles): every run length that divides the tiles of an image gives the bits of the
    one-tile-per-workgroup form -- several tiles per run, runs that end at an image end, channel tails, three images"""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import Act
    L = _lib.lib()
    N, C, H, W, Cout = dims
    x = synth_input(f"hsx{dims}", (N, C, H, W), 47, scale=1.5) + 0.2
    gam = 1.0 + 0.2 * synth_input("hsg", (C,), 47)
    bet = 0.1 * synth_input("hsb", (C,), 47)
    w = synth_input(f"hsw{dims}", (Cout, C, 3, 3), 47, scale=0.05)
    b = synth_input(f"hsbias{dims}", (Cout,), 47, scale=0.1)

    def run(tpw):
        prev = L.eod_set_option(b"head_tpw", tpw)
        try:
            prog = Program(DEV, prec)
            a = prog.act(N, H, W, C)
            a.t.copy_(x.permute(0, 2, 3, 1).to(DEV).to(a.t.dtype))
            out = torch.full((N, Cout, H, W), 7.0, dtype=torch.float32, device=DEV)
            gn = (prog.gn_stats([a], prog.f32(gam.to(DEV)), prog.f32(bet.to(DEV))), True)
            _, idx = prog.conv(a, prog.pack_conv(w.to(DEV)), prog.f32(b.to(DEV)), Cout, out_nchw_f32=True, gn=gn)
            assert prog.op_stats()[-1]["kernel"] == "conv_head_kernel"
            prog.ops[idx].u.conv.y = out.data_ptr()
            prog.run()
            torch.cuda.synchronize()
            return out.cpu()
        finally:
            L.eod_set_option(b"head_tpw", prev)

    one = run(1)
    ref = F.conv2d(F.silu(F.group_norm(x, 32, gam, bet, eps=1e-5)), w, b, padding=1)
    assert rel_l2(one, ref) < TOL[prec]
    for tpw in tpws:
        assert (H // 8) * (W // 16) % tpw == 0, "the test wants run lengths that divide the tiles of an image"
        assert torch.equal(run(tpw), one), tpw


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
@pytest.mark.parametrize("n,ld", [(1024, 1024), (1500, 1504), (4096, 4096), (16384, 16384)])
def test_softmax_long_rows_forward_and_backward(prec, n, ld):
    """attention rows of thousands of keys: the one-workgroup-per-row kernels (row in registers, one pass) for the softmax
    (unet_openai.py:479) and its backward, ragged length with zeroed pad columns"""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import current_stream_ptr
    L = _lib.lib()
    rows = 9
    dt = _lib.EOD_F16 if prec == "fp16" else _lib.EOD_F32
    tdt = torch.float16 if prec == "fp16" else torch.float32
    s = synth_input(f"sl{n}", (rows, ld), 5, scale=3.0)
    sd = s.to(DEV)
    p = torch.full((rows, ld), 7.0, dtype=tdt, device=DEV)
    st = current_stream_ptr(torch.device(DEV))
    _lib.check(L.eod_softmax_rows(sd.data_ptr(), ld, p.data_ptr(), ld, dt, rows, n, st), "softmax")
    ref = torch.softmax(s[:, :n].double(), -1)
    assert rel_l2(p.float().cpu()[:, :n], ref.float()) < (2e-6 if prec == "fp32" else 2e-3)
    assert float(p.float().cpu()[:, n:].abs().max() if ld > n else 0.0) == 0.0
    # backward: dS = P * (dP - sum(dP * P)) with the P the kernel stored
    dp = synth_input(f"sg{n}", (rows, ld), 6).to(DEV)
    ds = torch.full((rows, ld), 3.0, dtype=tdt, device=DEV)
    _lib.check(L.eod_softmax_bwd_rows(p.data_ptr(), ld, dp.data_ptr(), ld, ds.data_ptr(), dt, rows, n, st), "softmax_bwd")
    pf = p.float().cpu()[:, :n].double()
    g = dp.cpu()[:, :n].double()
    ref_ds = pf * (g - (g * pf).sum(-1, keepdim=True))
    assert rel_l2(ds.float().cpu()[:, :n], ref_ds.float()) < (2e-6 if prec == "fp32" else 2e-3)
    assert float(ds.float().cpu()[:, n:].abs().max() if ld > n else 0.0) == 0.0


@pytest.mark.parametrize("T,heads,d,new_order", [(128, 2, 16, False), (256, 2, 32, True), (256, 1, 48, False), (384, 2, 64, False), (128, 3, 8, True),
                                                 (49, 2, 16, False), (196, 1, 32, True), (200, 2, 48, False), (64, 2, 64, False)])
@pytest.mark.parametrize("neg_logits", [False, True])
def test_flash_attention_backward_vs_autograd(T, heads, d, new_order, neg_logits, dO_scale=0.5):
    """eod_attention_fwd (with log-sum-exp) + eod_rowdot + eod_attention_bwd against torch autograd of
    softmax(q k^T / sqrt(d)) v on the natural qkv layout (legacy [h][q|k|v][d] and new [q|k|v][h][d] orders).
    neg_logits: every logit of every row is around -35 (q and k carry opposite offsets, as a negative b_q . b_k bias term gives):
    lse << 0, so exp(-lse) is huge -- on a ragged last key tile (T = 49, 196, 200) a zero-filled key row then has an unbounded
    dS = -exp(-lse) D that becomes inf in fp16 and NaN against the zero K row unless the tile masks keys beyond T"""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import current_stream_ptr
    L = _lib.lib()
    N, C = 2, heads * d
    qkv = synth_input(f"fa{T}{d}", (N, T, 3 * C), 7, scale=0.7)
    dO = synth_input(f"fd{T}{d}", (N, T, C), 8, scale=dO_scale)
    qo, ko, vo, hs = (0, C, 2 * C, d) if new_order else (0, d, 2 * d, 3 * d)
    if neg_logits:
        off = math.sqrt(35.0 / math.sqrt(d))  # q . k / sqrt(d) = -off^2 d / sqrt(d) + O(1) = -35
        for h in range(heads):
            qkv[:, :, qo + h * hs: qo + h * hs + d] = 0.3 * qkv[:, :, qo + h * hs: qo + h * hs + d] + off
            qkv[:, :, ko + h * hs: ko + h * hs + d] = 0.3 * qkv[:, :, ko + h * hs: ko + h * hs + d] - off

    def split(t):  # -> q, k, v as [N, heads, T, d]
        idx = lambda off: torch.stack([t[:, :, off + h * hs: off + h * hs + d] for h in range(heads)], 1)
        return idx(qo), idx(ko), idx(vo)

    qh = qkv.half()
    ref_in = qh.float().clone().requires_grad_(True)
    q, k, v = split(ref_in)
    P = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), -1)
    O = (P @ v).permute(0, 2, 1, 3).reshape(N, T, C)
    O.backward(dO.half().float())
    # HIP path: O and lse from the fused forward on the natural layout, then the backward
    st = current_stream_ptr(torch.device(DEV))
    qd = qh.to(DEV)
    out = torch.empty((N * T, C), dtype=torch.float16, device=DEV)
    lse = torch.empty((N, heads, T), dtype=torch.float32, device=DEV)
    _lib.check(L.eod_attention_fwd_nat(qd.data_ptr(), out.data_ptr(), lse.data_ptr(), _lib.EOD_F16, N, T, C, heads, d, qo, ko, vo, hs, 0, 0, st),
               "attention_fwd_nat")
    assert rel_l2(out.float().cpu().reshape(N, T, C), O.detach()) < 3e-3
    lse_ref = torch.logsumexp(q.detach() @ k.detach().transpose(-1, -2) / math.sqrt(d), -1)
    assert float((lse.cpu() - lse_ref).abs().max()) < 2e-3
    dOd = dO.half().to(DEV)
    D = torch.empty((N, heads, T), dtype=torch.float32, device=DEV)
    od = out.reshape(N, T, C)
    _lib.check(L.eod_rowdot(dOd.data_ptr(), od.data_ptr(), _lib.EOD_F16, N, heads, T, T * C, d, C, d, D.data_ptr(), st), "rowdot")
    dqkv = torch.zeros((N, T, 3 * C), dtype=torch.float16, device=DEV)
    _lib.check(L.eod_attention_bwd(qd.data_ptr(), dOd.data_ptr(), lse.data_ptr(), D.data_ptr(), dqkv.data_ptr(), _lib.EOD_F16, N, T, C, heads, d,
                                   qo, ko, vo, hs, st), "attention_bwd")
    torch.cuda.synchronize()
    g = dqkv.float().cpu()
    assert torch.isfinite(g).all()
    for name, off in (("dq", qo), ("dk", ko), ("dv", vo)):
        for h in range(heads):
            a, b_ = g[:, :, off + h * hs: off + h * hs + d], ref_in.grad[:, :, off + h * hs: off + h * hs + d]
            # (neg_logits: q and k carry a common offset of +-off that cancels in dQ = sum_s dS_s k_s because sum_s dS_s = 0 -- the
            #  fp16 rounding of dS is amplified by |off| / spread = ~10x against the result; 1.5e-2 there)
            assert rel_l2(a, b_) < (1.5e-2 if neg_logits else 6e-3), (name, h, rel_l2(a, b_))


@pytest.mark.parametrize("T,heads,d", [(4, 4, 16), (16, 6, 16), (36, 2, 32), (64, 2, 16)])
def test_flash_attention_backward_short_sequence_with_loss_scaled_gradient(T, heads, d):
    """a short sequence (P ~ 1/T is NOT small) with an output gradient as large as a loss scale of 1024 makes it on a tiny prediction
    tensor (|dO| ~ 50): dS = P (dP - D) carried on 2^12, the scale T >= 4096 needs, overflowed fp16 here (NaN gradients, training
    fuzz case 15); the scale now follows T (attn_bwd.hip: AttnBwdP::ds_log2)"""
    test_flash_attention_backward_vs_autograd(T, heads, d, False, False, dO_scale=50.0)


WIDE_HEADS = [(256, 1, 512, False), (16, 1, 512, True), (77, 2, 128, False), (300, 1, 96, True), (1024, 1, 256, False), (50, 3, 72, False),
              (40, 1, 384, True), (33, 2, 200, False)]  # head dims above 64: csrc/attn_wide.hip (512 = the train.py:50 middle block)


@pytest.mark.parametrize("T,heads,d,new_order", [(49, 2, 16, False), (196, 1, 32, True), (128, 2, 48, False), (1000, 2, 64, False), (4096, 1, 48, True),
                                                 (64, 4, 8, False), (144, 6, 32, True), (36, 9, 32, True), (2304, 3, 32, True), (300, 2, 24, True)]
                         + WIDE_HEADS)
def test_attention_forward_natural_layout(T, heads, d, new_order):
    """eod_attention_fwd_nat: fused attention straight on the qkv conv output (both channel orders), ragged sequence lengths,
    with the log-sum-exp output -- vs torch softmax(q k^T / sqrt(d)) v"""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import current_stream_ptr
    L = _lib.lib()
    N, C = 2, heads * d
    qkv = synth_input(f"fn{T}{d}", (N, T, 3 * C), 9, scale=0.8).half()
    qo, ko, vo, hs = (0, C, 2 * C, d) if new_order else (0, d, 2 * d, 3 * d)
    pick = lambda off: torch.stack([qkv.float()[:, :, off + h * hs: off + h * hs + d] for h in range(heads)], 1)
    q, k, v = pick(qo), pick(ko), pick(vo)
    S = q @ k.transpose(-1, -2) / math.sqrt(d)
    ref = (torch.softmax(S, -1) @ v).permute(0, 2, 1, 3).reshape(N, T, C)
    qd = qkv.to(DEV)
    out = torch.full((N, T, C), 9.0, dtype=torch.float16, device=DEV)
    lse = torch.zeros((N, heads, T), dtype=torch.float32, device=DEV)
    _lib.check(L.eod_attention_fwd_nat(qd.data_ptr(), out.data_ptr(), lse.data_ptr(), _lib.EOD_F16, N, T, C, heads, d, qo, ko, vo, hs, 0, 0,
                                       current_stream_ptr(torch.device(DEV))), "attention_fwd_nat")
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu(), ref) < 3e-3
    assert float((lse.cpu() - torch.logsumexp(S, -1)).abs().max()) < 2e-3


@pytest.mark.parametrize("T,heads,d,new_order", [(49, 2, 16, False), (196, 1, 32, True), (128, 2, 48, False), (1000, 2, 64, False), (4096, 1, 48, True),
                                                 (64, 4, 8, False), (4096, 2, 64, False),
                                                 (144, 6, 32, True), (36, 9, 32, True), (2304, 3, 32, True), (300, 2, 24, True)] + WIDE_HEADS)
@pytest.mark.parametrize("mag", [0.8, 4.0])
@pytest.mark.parametrize("bound", [False, True])
def test_attention_forward_natural_layout_fp32(T, heads, d, new_order, mag, bound):
    """the fp32-storage instance of eod_attention_fwd_nat (fp32 online softmax, both contractions as three fp16 MFMAs per product
    on split operands; the T x T weights never exist) vs a float64 softmax(q k^T / sqrt(d)) v: fp32-grade, also for peaked
    softmax rows (mag 4: logits of +-50); with the fixed operand scale (no table) and with the per-image scale from a bound table"""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import current_stream_ptr
    L = _lib.lib()
    N, C = 2, heads * d
    qkv = synth_input(f"fn32{T}{d}", (N, T, 3 * C), 9, scale=mag)
    qo, ko, vo, hs = (0, C, 2 * C, d) if new_order else (0, d, 2 * d, 3 * d)
    pick = lambda off: torch.stack([qkv.double()[:, :, off + h * hs: off + h * hs + d] for h in range(heads)], 1)
    q, k, v = pick(qo), pick(ko), pick(vo)
    S = q @ k.transpose(-1, -2) / math.sqrt(d)
    ref = (torch.softmax(S, -1) @ v).permute(0, 2, 1, 3).reshape(N, T, C)
    qd = qkv.to(DEV)
    out = torch.full((N, T, C), 9.0, dtype=torch.float32, device=DEV)
    lse = torch.zeros((N, heads, T), dtype=torch.float32, device=DEV)
    ab = torch.zeros((N, 32), dtype=torch.float32, device=DEV)
    st = current_stream_ptr(torch.device(DEV))
    if bound:
        _lib.check(L.eod_act_bound(qd.data_ptr(), _lib.EOD_F32, N, T * 3 * C, 0, 0, 0, 0, 0, 0, ab.data_ptr(), 0, st), "act_bound")
    _lib.check(L.eod_attention_fwd_nat(qd.data_ptr(), out.data_ptr(), lse.data_ptr(), _lib.EOD_F32, N, T, C, heads, d, qo, ko, vo, hs,
                                       ab.data_ptr() if bound else 0, 0, st), "attention_fwd_nat")
    torch.cuda.synchronize()
    if bound:
        assert torch.equal(ab.cpu().max(1).values, qkv.abs().amax((1, 2)))  # the direct pass is the exact max|x| per image
    err = rel_l2(out.cpu(), ref)
    f32 = rel_l2((torch.softmax(S.float(), -1) @ v.float()).permute(0, 2, 1, 3).reshape(N, T, C), ref)  # what plain fp32 torch gives
    print(f"T={T} d={d} mag={mag}: fused fp32x3 attention {err:.2e}, torch fp32 {f32:.2e}")
    assert err < 2e-6
    assert float((lse.cpu().double() - torch.logsumexp(S, -1)).abs().max()) < 2e-5 * max(1.0, float(S.abs().max()))


@pytest.mark.parametrize("T,heads,d,new_order", [(49, 2, 16, False), (196, 1, 32, True), (128, 2, 48, False), (1000, 2, 64, False), (4096, 1, 48, True),
                                                 (64, 4, 8, False), (300, 3, 24, False), (256, 1, 40, True), (130, 2, 56, False),
                                                 (144, 6, 32, True), (36, 9, 32, True), (2304, 3, 32, True), (300, 2, 24, True)])
@pytest.mark.parametrize("mag", [0.8, 4.0])
def test_attention_forward_natural_layout_exact_fp32(T, heads, d, new_order, mag):
    """the exact-fp32 instance of eod_attention_fwd_nat (EOD_ATTN_EXACT_F32: IEEE fp32 products on v_mfma_f32_32x32x2_f32, fp32 online
    softmax; the fused kernel of the exact `fp32` precision mode, csrc/attn_f32.hip) vs a float64 softmax(q k^T / sqrt(d)) v: every
    head dim that is a multiple of 8 up to 64, both channel orders, ragged sequence lengths, peaked softmax rows (mag 4)"""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import current_stream_ptr
    L = _lib.lib()
    N, C = 2, heads * d
    qkv = synth_input(f"fx32{T}{d}", (N, T, 3 * C), 9, scale=mag)
    qo, ko, vo, hs = (0, C, 2 * C, d) if new_order else (0, d, 2 * d, 3 * d)
    pick = lambda off: torch.stack([qkv.double()[:, :, off + h * hs: off + h * hs + d] for h in range(heads)], 1)
    q, k, v = pick(qo), pick(ko), pick(vo)
    S = q @ k.transpose(-1, -2) / math.sqrt(d)
    ref = (torch.softmax(S, -1) @ v).permute(0, 2, 1, 3).reshape(N, T, C)
    qd = qkv.to(DEV)
    out = torch.full((N, T, C), 9.0, dtype=torch.float32, device=DEV)
    lse = torch.zeros((N, heads, T), dtype=torch.float32, device=DEV)
    _lib.check(L.eod_attention_fwd_nat(qd.data_ptr(), out.data_ptr(), lse.data_ptr(), _lib.EOD_F32, N, T, C, heads, d, qo, ko, vo, hs, 0,
                                       _lib.ATTN_EXACT_F32, current_stream_ptr(torch.device(DEV))), "attention_fwd_nat")
    torch.cuda.synchronize()
    err = rel_l2(out.cpu(), ref)
    f32 = rel_l2((torch.softmax(S.float(), -1) @ v.float()).permute(0, 2, 1, 3).reshape(N, T, C), ref)  # what plain fp32 torch gives
    print(f"T={T} d={d} mag={mag}: fused exact-fp32 attention {err:.2e}, torch fp32 {f32:.2e}")
    assert err < 2e-6 and err < 4 * f32 + 3e-7
    assert float((lse.cpu().double() - torch.logsumexp(S, -1)).abs().max()) < 2e-5 * max(1.0, float(S.abs().max()))


@pytest.mark.parametrize("N,H,W,Cx,Cout,ups", [(2, 16, 64, 40, 24, False), (1, 32, 32, 136, 128, False), (3, 16, 16, 8, 200, False),
                                               (1, 8, 128, 64, 64, False), (2, 16, 32, 24, 40, True), (1, 8, 8, 16, 16, True)])
def test_conv3x3_backward_weights_kernel(N, H, W, Cx, Cout, ups):
    """eod_conv3x3_wgrad (pixel-major staging + transposed LDS operand reads) + eod_wgrad_reduce vs torch's conv2d weight
    gradient: 64- / 32- / 16-wide strips, channel counts that are not tile multiples, the nearest-2x input variant, several splits"""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import current_stream_ptr
    L = _lib.lib()
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    x = synth_input(f"wx{N}{H}{W}{Cx}", (N, Cx, H, W), 11).half()
    dy = synth_input(f"wy{N}{H}{W}{Cout}", (N, Cout, Ho, Wo), 12, scale=0.5).half()
    xin = F.interpolate(x.float(), scale_factor=2, mode="nearest") if ups else x.float()
    ref = torch.nn.grad.conv2d_weight(xin, (Cout, Cx, 3, 3), dy.float(), padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    st = current_stream_ptr(torch.device(DEV))
    for S in (1, 5):
        partial = torch.full((S, 9, Cout, Cx), 7.0, dtype=torch.float32, device=DEV)
        dw = torch.zeros((Cout, Cx, 3, 3), dtype=torch.float32, device=DEV)
        _lib.check(L.eod_conv3x3_wgrad(dyd.data_ptr(), xd.data_ptr(), _lib.EOD_F16, N, H, W, Cx, Ho, Wo, Cout, Cout, int(ups), partial.data_ptr(), Cx, S, st),
                   "conv3x3_wgrad")
        _lib.check(L.eod_wgrad_reduce(partial.data_ptr(), S, 3, Cout, Cx, Cx, 0, Cx, 1.0, dw.data_ptr(), st), "wgrad_reduce")
        torch.cuda.synchronize()
        assert rel_l2(dw.cpu(), ref) < 2e-3, (S, rel_l2(dw.cpu(), ref))


@pytest.mark.parametrize("N,H,W,Cx,Cy,Cout", [(3, 5, 7, 40, 48, 40), (2, 16, 16, 136, 128, 128), (1, 32, 32, 8, 200, 200), (2, 8, 8, 264, 16, 16)])
def test_conv1x1_backward_weights_kernel(N, H, W, Cx, Cy, Cout):
    """eod_conv1x1_wgrad (gemm_tn_kernel over pixel ranges, operands read pixel-major as stored) + eod_wgrad_reduce vs torch's
    conv2d weight gradient: ragged pixel counts (the last 64-row strip is partial), channel counts that are not tile multiples,
    Cout < row pitch of dY, split counts that leave trailing splits empty"""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import current_stream_ptr
    L = _lib.lib()
    x = synth_input(f"w1x{N}{H}{W}{Cx}", (N, Cx, H, W), 13).half()
    dy = synth_input(f"w1y{N}{H}{W}{Cy}", (N, Cy, H, W), 14, scale=0.5).half()
    ref = torch.nn.grad.conv2d_weight(x.float(), (Cout, Cx, 1, 1), dy.float()[:, :Cout])
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    st = current_stream_ptr(torch.device(DEV))
    npix = N * H * W
    for S in (1, 3, 7):
        partial = torch.full((S, 1, Cout, Cx), 7.0, dtype=torch.float32, device=DEV)
        dw = torch.zeros((Cout, Cx, 1, 1), dtype=torch.float32, device=DEV)
        _lib.check(L.eod_conv1x1_wgrad(dyd.data_ptr(), xd.data_ptr(), _lib.EOD_F16, npix, Cx, Cy, Cout, partial.data_ptr(), Cx, S, st), "conv1x1_wgrad")
        _lib.check(L.eod_wgrad_reduce(partial.data_ptr(), S, 1, Cout, Cx, Cx, 0, Cx, 1.0, dw.data_ptr(), st), "wgrad_reduce")
        torch.cuda.synchronize()
        assert rel_l2(dw.cpu(), ref) < 2e-3, (S, rel_l2(dw.cpu(), ref))


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
@pytest.mark.parametrize("N,C0,C1,H,W,silu", [(2, 64, 0, 8, 8, True), (2, 104, 88, 7, 5, True), (1, 1280, 0, 6, 5, True), (2, 1536, 1024, 3, 3, False)])
def test_group_norm_silu_backward_kernels(prec, N, C0, C1, H, W, silu):
    """the GroupNorm32(+SiLU) backward chain (statistics -> eod_gn_mean_rstd -> eod_gn_bwd_partial -> _finalize -> _params / _apply)
    over one or two concat sources against torch autograd, called through the C ABI; the wide cases need several channel blocks
    (more than 256 16-byte chunks per pixel) and a group that straddles the concat seam"""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import current_stream_ptr
    L = _lib.lib()
    td, dt = (torch.float32, _lib.EOD_F32) if prec == "fp32" else (torch.float16, _lib.EOD_F16)
    Ct, HW, G = C0 + C1, H * W, 32
    x = (synth_input(f"gbx{N}{Ct}{H}", (N, Ct, H, W), 51, scale=1.5) + 0.3).to(td).float()
    dy = synth_input(f"gby{N}{Ct}{H}", (N, Ct, H, W), 52).to(td).float()
    gam = 1.0 + 0.2 * synth_input("gbg", (Ct,), 53)
    bet = 0.1 * synth_input("gbb", (Ct,), 53)
    xr = x.clone().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    y = F.group_norm(xr, G, gr, br, eps=1e-5)
    if silu:
        y = F.silu(y)
    y.backward(dy)
    st = current_stream_ptr(torch.device(DEV))
    nhwc = lambda t4: t4.permute(0, 2, 3, 1).contiguous().to(td).to(DEV)
    srcs = [(nhwc(x[:, :C0]), C0, 0)] + ([(nhwc(x[:, C0:]), C1, C0)] if C1 else [])
    dyd = nhwc(dy)
    P = max(1, min(256, HW // 64))
    f32 = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=DEV)
    parts = [f32(N, P, c, 2) for _, c, _ in srcs]
    for (xs, c, _), pt in zip(srcs, parts):
        _lib.check(L.eod_gn_partial(xs.data_ptr(), dt, N, HW, c, pt.data_ptr(), P, c, 0, st), "gn_partial")
    p1 = (parts[1].data_ptr(), P, C1) if C1 else (0, 0, 0)
    ss, mr = f32(N, Ct, 2), f32(N, G, 2)
    gd, bd = gam.to(DEV), bet.to(DEV)
    _lib.check(L.eod_gn_finalize(parts[0].data_ptr(), P, C0, p1[0], p1[1], p1[2], N, HW, G, 1e-5, gd.data_ptr(), bd.data_ptr(), 0, 0, ss.data_ptr(), 0, 0, st), "gn_finalize")
    _lib.check(L.eod_gn_mean_rstd(parts[0].data_ptr(), P, C0, p1[0], p1[1], p1[2], N, HW, G, 1e-5, mr.data_ptr(), st), "gn_mean_rstd")
    part, coef, gb = f32(N, P, Ct, 2), f32(N, Ct, 3), f32(N, Ct, 2)
    for xs, c, off in srcs:
        _lib.check(L.eod_gn_bwd_partial(xs.data_ptr(), dyd.data_ptr(), ss.data_ptr(), dt, N, HW, c, part.data_ptr(), P, Ct, off, int(silu), st), "gn_bwd_partial")
    _lib.check(L.eod_gn_bwd_finalize(part.data_ptr(), P, Ct, N, HW, G, mr.data_ptr(), gd.data_ptr(), bd.data_ptr(), 0, 0, 0, 0, coef.data_ptr(), gb.data_ptr(), st), "gn_bwd_finalize")
    dgam, dbet = f32(Ct), f32(Ct)
    _lib.check(L.eod_gn_bwd_params(gb.data_ptr(), N, Ct, 1.0, dgam.data_ptr(), dbet.data_ptr(), st), "gn_bwd_params")
    dxs = []
    for xs, c, off in srcs:
        dx = torch.empty_like(xs)
        _lib.check(L.eod_gn_bwd_apply(xs.data_ptr(), dyd.data_ptr(), ss.data_ptr(), coef.data_ptr(), 0, dt, N, HW, c, Ct, off, int(silu), dx.data_ptr(), 0, st), "gn_bwd_apply")
        dxs.append(dx.float().cpu().permute(0, 3, 1, 2))
    torch.cuda.synchronize()
    tol = 2e-5 if prec == "fp32" else 3e-3
    assert rel_l2(torch.cat(dxs, 1), xr.grad) < tol
    assert rel_l2(dgam.cpu(), gr.grad) < tol and rel_l2(dbet.cpu(), br.grad) < tol


@pytest.mark.parametrize("case", [(2, 128, 8, 16, 128, False), (1, 96, 16, 32, 192, True), (3, 72, 24, 16, 256, False), (1, 640, 16, 16, 160, True)])
def test_conv_nearest_upsample_parity_class_backward_data_vs_autograd(case):
    """upsample='up4b' (conv_up4_halo_kernel<BWD>, fp16): dX of `3x3 conv over the nearest-2x upsampling of x` straight from dY -- four
    2x2-tap convs over the stride-2 views of dY with the transposed class kernels (eod_conv_up4_weights -> eod_pack_conv_weight_dgrad),
    accumulated in one tile, plus an optional second gradient branch (`res`) -- vs torch autograd through F.interpolate + F.conv2d"""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import Act, current_stream_ptr
    N, Cx, H, W, Cy, with_res = case
    x = synth_input(f"u4bx{case}", (N, Cx, H, W), 44).half().float().requires_grad_(True)
    w = synth_input(f"u4bw{case}", (Cy, Cx, 3, 3), 44, scale=1.0 / math.sqrt(Cx * 9)).half().float()
    dy = synth_input(f"u4bd{case}", (N, Cy, 2 * H, 2 * W), 44).half().float()
    r = synth_input(f"u4br{case}", (N, Cx, H, W), 44).half().float()
    F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, None, padding=1).backward(dy)
    ref = x.grad + (r if with_res else 0)

    def emit(prog, a):  # a = dY as the program's input activation
        L = prog.L
        st = current_stream_ptr(prog.device)
        if os.environ.get("EOD_UP4") == "0":
            pytest.skip("the parity-class kernels are switched off by the environment (A/B run)")
        assert prog.conv_up4_bwd_ok(a, Cx)
        wd_ = w.to(DEV).contiguous()
        wc = prog.own(torch.empty((4 * Cy, Cx, 3, 3), dtype=torch.float32, device=DEV))
        _lib.check(L.eod_conv_up4_weights(wd_.data_ptr(), wc.data_ptr(), Cy, Cx, st), "conv_up4_weights")
        wd = prog.empty((9, Cx, 4 * Cy))
        _lib.check(L.eod_pack_conv_weight_dgrad(wc.data_ptr(), wd.data_ptr(), prog.dt, 4 * Cy, Cx, 3, 0, Cx, 4 * Cy, st), "pack_dgrad")
        rr = Act(prog.own(r.to(DEV).permute(0, 2, 3, 1).contiguous().to(prog.tdtype)), N, H, W, Cx) if with_res else None
        g, _ = prog.conv(a, wd, None, Cx, ksize=3, stride=1, pad=1, upsample="up4b", res=rr)
        return g

    got = run_program("fp16", dy, emit)
    assert got.shape == ref.shape
    assert rel_l2(got, ref) < 2e-3, rel_l2(got, ref)


@pytest.mark.parametrize("N,H,W,Cx,Cout", [(2, 8, 64, 40, 24), (1, 16, 32, 136, 128), (3, 8, 16, 8, 200), (1, 4, 128, 64, 64)])
def test_conv3x3_backward_weights_parity_class_form(N, H, W, Cx, Cout):
    """eod_conv3x3_wgrad(ups = 2) + eod_wgrad_reduce(ksize 4) + eod_wgrad_up4_map: the weight gradient of a 3x3 conv over the nearest-2x
    upsampling of X from the 16 class / tap correlations of the stride-2 views of dY (4/9 of the MACs of the nine-tap form) vs torch"""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import current_stream_ptr
    L = _lib.lib()
    x = synth_input(f"wcx{N}{H}{W}{Cx}", (N, Cx, H, W), 15).half()
    dy = synth_input(f"wcy{N}{H}{W}{Cout}", (N, Cout, 2 * H, 2 * W), 16, scale=0.5).half()
    ref = torch.nn.grad.conv2d_weight(F.interpolate(x.float(), scale_factor=2, mode="nearest"), (Cout, Cx, 3, 3), dy.float(), padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    st = current_stream_ptr(torch.device(DEV))
    for S in (1, 3):
        partial = torch.full((S, 16, Cout, Cx), 7.0, dtype=torch.float32, device=DEV)
        t16 = torch.zeros((Cout, Cx, 16), dtype=torch.float32, device=DEV)
        dw = torch.zeros((Cout, Cx, 3, 3), dtype=torch.float32, device=DEV)
        _lib.check(L.eod_conv3x3_wgrad(dyd.data_ptr(), xd.data_ptr(), _lib.EOD_F16, N, H, W, Cx, 2 * H, 2 * W, Cout, Cout, 2, partial.data_ptr(), Cx, S, st),
                   "conv3x3_wgrad")
        _lib.check(L.eod_wgrad_reduce(partial.data_ptr(), S, 4, Cout, Cx, Cx, 0, Cx, 1.0, t16.data_ptr(), st), "wgrad_reduce")
        _lib.check(L.eod_wgrad_up4_map(t16.data_ptr(), Cout, Cx, dw.data_ptr(), st), "wgrad_up4_map")
        torch.cuda.synchronize()
        assert rel_l2(dw.cpu(), ref) < 2e-3, (S, rel_l2(dw.cpu(), ref))


@pytest.mark.parametrize("N,Ho,Wo,Cx,Cout", [(2, 8, 64, 40, 24), (1, 16, 32, 136, 128), (3, 8, 16, 8, 200), (1, 4, 128, 64, 64)])
def test_conv3x3_backward_weights_stride2(N, Ho, Wo, Cx, Cout):
    """eod_conv3x3_wgrad(ups = 3): weight gradient of a stride-2 / pad-1 3x3 conv (Downsample.op) straight from the NHWC tensors -- X
    gathered at pixel stride 2 in two column phases -- vs torch's conv2d weight gradient"""
    from eo_diffusion_amd import _lib
    from eo_diffusion_amd.engine import current_stream_ptr
    L = _lib.lib()
    H, W = 2 * Ho, 2 * Wo
    x = synth_input(f"w2x{N}{H}{W}{Cx}", (N, Cx, H, W), 17).half()
    dy = synth_input(f"w2y{N}{Ho}{Wo}{Cout}", (N, Cout, Ho, Wo), 18, scale=0.5).half()
    ref = torch.nn.grad.conv2d_weight(x.float(), (Cout, Cx, 3, 3), dy.float(), stride=2, padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    st = current_stream_ptr(torch.device(DEV))
    for S in (1, 3):
        partial = torch.full((S, 9, Cout, Cx), 7.0, dtype=torch.float32, device=DEV)
        dw = torch.zeros((Cout, Cx, 3, 3), dtype=torch.float32, device=DEV)
        _lib.check(L.eod_conv3x3_wgrad(dyd.data_ptr(), xd.data_ptr(), _lib.EOD_F16, N, H, W, Cx, Ho, Wo, Cout, Cout, 3, partial.data_ptr(), Cx, S, st),
                   "conv3x3_wgrad")
        _lib.check(L.eod_wgrad_reduce(partial.data_ptr(), S, 3, Cout, Cx, Cx, 0, Cx, 1.0, dw.data_ptr(), st), "wgrad_reduce")
        torch.cuda.synchronize()
        assert rel_l2(dw.cpu(), ref) < 2e-3, (S, rel_l2(dw.cpu(), ref))
