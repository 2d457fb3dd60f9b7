#!/usr/bin/env python3
"""Wall time of one training step (forward + MSE gradient + backward [+ torch AdamW]) on the HIP path.
   python tools/train_bench.py [--arch A0] [--size 256] [--batch 16] [--precision fp16] [--steps 10] [--optimizer]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
from eo_diffusion_amd.training import UNetTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="A0")
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--precision", default="fp16")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--optimizer", default="none", choices=["none", "torch", "fused"])
ap.add_argument("--in-ch", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
m = build_model(a.arch, a.size, a.precision, dev, in_ch=a.in_ch)
unet = m.model.train()
x = torch.randn(a.batch, a.in_ch, a.size, a.size, device=dev)
noise = torch.randn_like(x)
t = torch.randint(0, 1000, (a.batch,), device=dev)
opt = None  # (the optimizer first: the fused AdamW moves the parameters into its flat buffer, the trainer bakes their pointers)
if a.optimizer == "torch":
    opt = torch.optim.AdamW(unet.parameters(), lr=1e-4)
elif a.optimizer == "fused":
    from eo_diffusion_amd.optim import AdamW
    opt = AdamW(unet.parameters(), lr=1e-4)

t0 = time.perf_counter()
tr = UNetTrainer(unet, a.batch, a.size, a.size, dev, loss_scale=(1024.0 if a.precision == "fp16" else 1.0))
torch.cuda.synchronize()
print(f"build {time.perf_counter() - t0:.1f} s; forward buffers {tr.prog.nbytes / 2**30:.2f} GiB, backward buffers {tr.bprog.nbytes / 2**30:.2f} GiB, "
      f"{len(tr.bwd)} backward launches", flush=True)

def step():
    xt = m._forward_diffusion(x, t, noise)
    pred = tr.forward(xt, t)
    tr.backward(2.0 * (pred - noise) / pred.numel())
    if opt:
        opt.step()
        opt.zero_grad(set_to_none=True)
    return pred

for _ in range(2):
    pred = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    pred = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
# forward-only and backward-only split
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(a.steps):
    pred = tr.forward(x, t)
torch.cuda.synchronize(); tf = (time.perf_counter() - t1) / a.steps
gn = float(tr.flat_grad.norm())
print(f"{a.arch}@{a.size} batch {a.batch} {a.precision}: {dt * 1e3:.2f} ms / training step ({1 / dt:.2f} steps/s, {a.batch / dt:.1f} images/s), "
      f"forward alone {tf * 1e3:.2f} ms, optimizer={a.optimizer}, |grad|={gn:.4g}, "
      f"finite={bool(torch.isfinite(pred).all())}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
