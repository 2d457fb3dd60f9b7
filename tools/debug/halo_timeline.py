#!/usr/bin/env python3
"""Per-K-step timeline of one wave of conv3x3_halo_kernel (diagnostic build of igemm.hip with -DEOD_TSTAMP):
     EOD_LIBRARY=.../libeodiff_tstamp.so python tools/debug/halo_timeline.py [--prec fp32x3] [--shapes ...]
   wave 1 stamps the shader clock at: step top | after wait + barrier | after the DMA issue | after the last MFMA issued | (next top = after
   the in-place rewrite of a patch piece) for the nine taps of its second chunk; medians over workgroups, in shader cycles."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from eo_diffusion_amd import _lib
from eo_diffusion_amd.engine import Program
from tools.conv_bench import SHAPES

ap = argparse.ArgumentParser()
ap.add_argument("--prec", default="fp32x3")
ap.add_argument("--shapes", default="l0_128,l0_384,l1_256")
a = ap.parse_args()
dev = "cuda:0"
L = _lib.lib()
L.eod_debug_read_tstamps.restype = C.c_int
L.eod_debug_read_tstamps.argtypes = [C.c_void_p, C.c_int]
for name in a.shapes.split(","):
    N, H, W, Cin, Cout, k, stride, ups = SHAPES[name]
    prog = Program(dev, a.prec)
    x = prog.act(N, H, W, Cin)
    x.t.normal_()
    w = prog.pack_conv(torch.randn((Cout, Cin, k, k), device=dev) * 0.02)
    b = prog.empty((Cout,), torch.float32); b.normal_()
    gn = (prog.gn_stats([x], prog.f32(torch.ones(Cin, device=dev)), prog.f32(torch.zeros(Cin, device=dev))), True)
    y, _i = prog.conv(x, w, b, Cout, ksize=k, stride=stride, pad=k // 2, upsample=ups, gn=gn)
    prog.finalize()
    for _ in range(30):
        prog.run()
    torch.cuda.synchronize()
    buf = np.zeros((2048, 40), dtype=np.uint64)
    assert L.eod_debug_read_tstamps(buf.ctypes.data, 2048) == 0
    buf = buf[(buf[:, 0] > 0) & (buf[:, 36] > buf[:, 0])].astype(np.int64)
    d = np.diff(buf[:, :37], axis=1)   # 36 intervals: per tap [wait+barrier | DMA issue | reads + MFMAs | rewrite]
    med = np.median(d, axis=0).reshape(9, 4)
    print(f"{name} {a.prec} gn: {len(buf)} workgroups; chunk of 9 taps: median {np.median(buf[:, 36] - buf[:, 0]):.0f} cycles")
    print("  tap   wait+barrier   DMA issue   reads+MFMAs   rewrite     step")
    for t in range(9):
        print(f"  {t}   {med[t, 0]:10.0f} {med[t, 1]:11.0f} {med[t, 2]:13.0f} {med[t, 3]:9.0f} {med[t].sum():8.0f}")
    print(f"  sum {med[:, 0].sum():10.0f} {med[:, 1].sum():11.0f} {med[:, 2].sum():13.0f} {med[:, 3].sum():9.0f} {med.sum():8.0f}", flush=True)
