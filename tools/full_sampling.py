#!/usr/bin/env python3
"""Wall time of a complete 1000-step DDPM sampling call through the public API (BASELINE metric 2: images/s),
including x_T generation; python tools/full_sampling.py [--batch 16] [--size 256] [--arch A0] [--precision fp16]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--arch", default="A0")
ap.add_argument("--precision", default="fp16")
a = ap.parse_args()
dev = torch.device("cuda", 0)
m = build_model(a.arch, a.size, a.precision, dev)
with torch.no_grad():
    m.model(torch.zeros(a.batch, 3, a.size, a.size, device=dev), torch.zeros(a.batch, dtype=torch.int64, device=dev))  # plan build
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x0 = m.sampling(a.batch, device="cuda:0", rng="philox", seed=7, progress=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"{a.arch}@{a.size} batch {a.batch} {a.precision}: {m.timesteps}-step sampling() = {dt:.3f} s -> {a.batch / dt:.3f} images/s, "
      f"{dt / m.timesteps * 1e3:.3f} ms/step, finite={bool(torch.isfinite(x0).all())}", flush=True)
