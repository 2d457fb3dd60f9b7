#!/usr/bin/env python3
"""One UNet forward in the three precision modes on the same synthetic weights / input; prints the distance of fp32x3 and fp16 from the
exact-fp32 mode (sizes the CPU oracle cannot reach).   python tools/cross_mode.py [--arch A0] [--size 512] [--ch 3] [--batch 1]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import ARCHS
from eo_diffusion_amd.backbones.unet_openai import UNetModel

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="A0", choices=list(ARCHS))
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--ch", type=int, default=3)
ap.add_argument("--batch", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
x = torch.randn(a.batch, a.ch, a.size, a.size, generator=g).to(dev)
t = torch.randint(0, 1000, (a.batch,), generator=g).to(dev)
torch.manual_seed(0)
ref = UNetModel(a.size, in_channels=a.ch, out_channels=a.ch, **ARCHS[a.arch])
with torch.no_grad():
    for p in ref.parameters():  # the reference zero-initialises conv2 / proj_out / out: re-draw so that every layer contributes
        if p.dim() > 1 and float(p.abs().max()) == 0.0:
            p.copy_(torch.randn(p.shape, generator=g) * 0.02)
sd = ref.state_dict()
out = {}
for prec in ("fp32", "fp32x3", "fp16"):
    m = UNetModel(a.size, in_channels=a.ch, out_channels=a.ch, **ARCHS[a.arch]).set_precision(prec)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    with torch.no_grad():
        out[prec] = m(x, t).double().cpu()
    del m
    torch.cuda.empty_cache()
rel = lambda u, v: float((u - v).norm() / v.norm())
print(f"{a.arch} @ {a.size}x{a.size}x{a.ch}, batch {a.batch}: fp32x3 vs exact fp32 {rel(out['fp32x3'], out['fp32']):.3e}, "
      f"fp16 vs exact fp32 {rel(out['fp16'], out['fp32']):.3e}, finite={bool(torch.isfinite(out['fp16']).all())}")
