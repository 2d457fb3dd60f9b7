#!/usr/bin/env python3
"""Calibration: what the vendor GEMM (torch.matmul -> hipBLASLt) reaches on this box for the conv-equivalent GEMM shapes.
Not part of the product path; used to put the conv kernel's roofline fraction into perspective."""
import time, torch
shapes = {"8192^3": (8192, 8192, 8192), "l1_640 equiv": (262144, 256, 5760), "l0_128 equiv": (1048576, 128, 1152),
          "l2_384 equiv": (65536, 384, 3456)}
for name, (M, N, K) in shapes.items():
    a = torch.randn(M, K, device="cuda", dtype=torch.float16) * 0.1
    b = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.1
    for _ in range(3): c = a @ b.t()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): c = a @ b.t()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name:14s} M={M} N={N} K={K}  {dt*1e3:8.3f} ms  {2*M*N*K/dt/1e12:8.1f} TF/s", flush=True)
    del a, b, c
