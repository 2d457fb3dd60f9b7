#!/usr/bin/env python3
"""Where a conv3x3_halo_kernel workgroup spends its life (diagnostic build of igemm.hip with -DEOD_STAMP):
     hipcc ... -DEOD_STAMP -c igemm.hip -> libeodiff_stamp.so;  EOD_LIBRARY=.../libeodiff_stamp.so python tools/debug/halo_stamps.py [--prec fp32x3] [--gn] [--shapes ...]
   per shape: median over workgroups of prologue / K loop / epilogue in microseconds (100 MHz wall counter) and the shader clock each phase
   ran at (s_memtime / s_memrealtime), plus the launch's wall time and how many workgroup "rounds" it had."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from eo_diffusion_amd import _lib
from eo_diffusion_amd.engine import Program
from tools.conv_bench import SHAPES

ap = argparse.ArgumentParser()
ap.add_argument("--prec", default="fp32x3")
ap.add_argument("--shapes", default="l0_128,l0_384,l1_256,l1_640,l2_384,l3_512")
ap.add_argument("--gn", action="store_true")
a = ap.parse_args()
dev = "cuda:0"
L = _lib.lib()
L.eod_debug_read_stamps.restype = C.c_int
L.eod_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
for name in a.shapes.split(","):
    N, H, W, Cin, Cout, k, stride, ups = SHAPES[name]
    prog = Program(dev, a.prec)
    x = prog.act(N, H, W, Cin)
    x.t.normal_()
    w = prog.pack_conv(torch.randn((Cout, Cin, k, k), device=dev) * 0.02)
    b = prog.empty((Cout,), torch.float32); b.normal_()
    gn = None
    if a.gn:
        gn = (prog.gn_stats([x], prog.f32(torch.ones(Cin, device=dev)), prog.f32(torch.zeros(Cin, device=dev))), True)
    y, _i = prog.conv(x, w, b, Cout, ksize=k, stride=stride, pad=k // 2, upsample=ups, gn=gn)
    prog.finalize()
    for _ in range(30):   # (the clock settles under sustained load)
        prog.run()
    torch.cuda.synchronize()
    nwg = min(65536, (N * H * W // 128) * max(1, Cout // 128))
    buf = np.zeros((nwg, 16), dtype=np.uint64)
    rc = L.eod_debug_read_stamps(buf.ctypes.data, nwg)
    assert rc == 0, rc
    buf = buf[buf[:, 0] > 0].astype(np.float64)
    slab = (buf[:, 8] - buf[:, 4]) * 0.01 if (buf[:, 8] > 0).all() else np.zeros(len(buf))   # K-loop end -> LDS transpose done (stamp 4; the direct epilogue has none)
    wall = buf[:, 0:8:2] * 0.01   # us
    clk = buf[:, 1:8:2]
    ph = np.diff(wall, axis=1)
    ck = np.diff(clk, axis=1) / np.maximum(ph, 1e-9) / 1e3  # GHz
    tot = wall[:, 3].max() - wall[:, 0].min()
    life = wall[:, 3] - wall[:, 0]
    print(f"{name:8s} {a.prec}{' gn' if gn else ''}: {len(buf)} workgroups (last launch), launch {tot:8.1f} us, workgroup life median {np.median(life):6.1f} us "
          f"=> {tot / np.median(life):4.1f} rounds | prologue {np.median(ph[:, 0]):5.2f} us  K loop {np.median(ph[:, 1]):6.2f} us  epilogue {np.median(ph[:, 2]):5.2f} us (bias + transpose into LDS {np.median(slab):5.2f}) "
          f"| clock GHz {np.median(ck[:, 0]):.2f} / {np.median(ck[:, 1]):.2f} / {np.median(ck[:, 2]):.2f}", flush=True)
