#!/usr/bin/env python3
"""Achieved HBM bandwidth of the streaming GroupNorm passes (statistics, apply, backward partial sums, backward apply) and of the
plain add kernel as the yardstick, at the shapes of the A0@256 / batch 16 training step.
   python tools/gn_bench.py [--dtype fp16] [--reps 20]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eo_diffusion_amd._lib import lib, check, EOD_F16, EOD_F32

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
L = lib()
dev = torch.device("cuda", 0)
td, dt, es = (torch.float16, EOD_F16, 2) if a.dtype == "fp16" else (torch.float32, EOD_F32, 4)
st = torch.cuda.current_stream().cuda_stream


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e-3


for (N, HW, C, Ctot) in [(16, 65536, 128, 128), (16, 65536, 128, 256), (16, 16384, 256, 256), (16, 16384, 256, 512), (16, 4096, 384, 768),
                         (16, 1024, 512, 1024)]:
    x = torch.randn(N, HW, C, device=dev).to(td)
    dy = torch.randn(N, HW, Ctot, device=dev).to(td)
    y = torch.empty(N, HW, Ctot, device=dev, dtype=td)
    add = torch.randn(N, HW, C, device=dev).to(td)
    dx = torch.empty_like(x)
    ss = torch.randn(N, Ctot, 2, device=dev)
    coef = torch.randn(N, Ctot, 3, device=dev)
    P = max(1, min(256, HW // 64))
    part = torch.empty(N, P, Ctot, 2, device=dev)
    p = lambda t: t.data_ptr()
    nb = N * HW * C * es
    rows = [
        ("add (yardstick)", 3 * nb, lambda: check(L.eod_add(p(x), p(add), p(dx), dt, N * HW * C, st), "add")),
        ("gn_partial", nb, lambda: check(L.eod_gn_partial(p(x), dt, N, HW, C, p(part), P, Ctot, 0, st), "gn_partial")),
        ("gn_apply+silu", 2 * nb, lambda: check(L.eod_gn_apply(p(x), dt, N, HW, C, p(ss), Ctot, 0, 1, p(y), 0, st), "gn_apply")),
        ("gn_bwd_partial", 2 * nb, lambda: check(L.eod_gn_bwd_partial(p(x), p(dy), p(ss), dt, N, HW, C, p(part), P, Ctot, 0, 1, st), "bwd_partial")),
        ("gn_bwd_apply", 3 * nb, lambda: check(L.eod_gn_bwd_apply(p(x), p(dy), p(ss), p(coef), 0, dt, N, HW, C, Ctot, 0, 1, p(dx), 0, st), "bwd_apply")),
        ("gn_bwd_apply+add", 4 * nb, lambda: check(L.eod_gn_bwd_apply(p(x), p(dy), p(ss), p(coef), p(add), dt, N, HW, C, Ctot, 0, 1, p(dx), 0, st), "bwd_apply")),
    ]
    print(f"N={N} HW={HW} C={C} of Ctot={Ctot} ({a.dtype}, tensor {nb / 2**20:.0f} MiB)")
    for name, byts, fn in rows:
        t = timed(fn)
        print(f"   {name:18s} {t * 1e6:8.1f} us   {byts / t / 1e12:5.2f} TB/s", flush=True)
