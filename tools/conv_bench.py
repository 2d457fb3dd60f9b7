#!/usr/bin/env python3
"""Single-shape conv micro-benchmark (kernel iteration / rocprofv3 target).
   python tools/conv_bench.py [--prec fp16|fp32|fp32x3] [--shapes name,...] [--iters 20] [--gn]   (--gn: GroupNorm+SiLU of the input fused)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eo_diffusion_amd.engine import Program

SHAPES = {  # name: (N, H, W, Cin, Cout, k, stride, upsample)
    "l0_128": (16, 256, 256, 128, 128, 3, 1, False),
    "l0_384": (16, 256, 256, 384, 128, 3, 1, False),
    "l0_256": (16, 256, 256, 256, 128, 3, 1, False),
    "l0_32": (16, 256, 256, 32, 128, 3, 1, False),
    "l0_64": (16, 256, 256, 64, 128, 3, 1, False),
    "l1_256": (16, 128, 128, 256, 256, 3, 1, False),
    "l1_640": (16, 128, 128, 640, 256, 3, 1, False),
    "l2_384": (16, 64, 64, 384, 384, 3, 1, False),
    "l3_512": (16, 32, 32, 512, 512, 3, 1, False),
    "up_256": (16, 128, 128, 256, 256, 3, 1, True),
    "up_384": (16, 64, 64, 384, 384, 3, 1, True),
    "up_512": (16, 32, 32, 512, 512, 3, 1, True),
    "sk_384": (16, 256, 256, 384, 128, 1, 1, False),
    "sk_256": (16, 256, 256, 256, 128, 1, 1, False),
    "sk_640": (16, 128, 128, 640, 256, 1, 1, False),
    "sk_768": (16, 64, 64, 768, 384, 1, 1, False),
    "dn_128": (16, 256, 256, 128, 128, 3, 2, False),
    "dn_256": (16, 128, 128, 256, 256, 3, 2, False),
    "head_128": (16, 256, 256, 128, 3, 3, 1, False),   # output head: NCHW fp32 output (use with --gn: conv_head_kernel)
    "head_192": (16, 128, 128, 192, 3, 3, 1, False),
    "qkv_384": (8, 64, 64, 384, 1152, 1, 1, False),    # A1 attention blocks at T = 4096: qkv and proj_out 1x1 convs
    "proj_384": (8, 64, 64, 384, 384, 1, 1, False),
    "qkv_512": (8, 32, 32, 512, 1536, 1, 1, False),
    "proj_512": (8, 32, 32, 512, 512, 1, 1, False),
    "dn_384": (16, 64, 64, 384, 384, 3, 2, False),
    "first_3": (16, 256, 256, 4, 128, 3, 1, False),
}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", default="fp16")
    ap.add_argument("--shapes", default="l0_128,l0_384,l1_256,l1_640,l2_384,l3_512,up_256,sk_384")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--gn", action="store_true")
    ap.add_argument("--up4", action="store_true", help="upsample shapes through conv_up4_halo_kernel")
    ap.add_argument("--batch", type=int, default=0, help="override the batch size of the shapes (default 16)")
    ap.add_argument("--tpw", default="", help="comma list of halo_tpw option values (tiles per workgroup of the streaming halo instances) to time in turn")
    ap.add_argument("--presplit", action="store_true", help="fp32x3, 1x1 shapes: the input comes PRE-SPLIT from a normalising pass (the qkv conv behind AttentionBlock.norm)")
    a = ap.parse_args()
    dev = "cuda:0"
    for name in a.shapes.split(","):
        N, H, W, Cin, Cout, k, stride, ups = SHAPES[name]
        N = a.batch or N
        prog = Program(dev, a.prec)
        x = prog.act(N, H, W, Cin)
        x.t.normal_()
        w = prog.pack_conv(torch.randn((Cout, Cin, k, k), device=dev) * 0.02)
        b = prog.empty((Cout,), torch.float32); b.normal_()
        gn = None
        if a.presplit and k == 1 and a.prec == "fp32x3":
            x = prog.group_norm([x], prog.f32(torch.ones(Cin, device=dev)), prog.f32(torch.zeros(Cin, device=dev)), silu=False, split_out=True)
        if a.gn and not ups and k == 3:
            gn = (prog.gn_stats([x], prog.f32(torch.ones(Cin, device=dev)), prog.f32(torch.zeros(Cin, device=dev))), True)
        if ups and a.up4 and prog.conv_up4_ok(x, Cout):  # the parity-class form of the nearest-2x conv (4/9 of the MACs; flops below = algorithmic)
            w = prog.pack_conv_up4(torch.randn((Cout, Cin, k, k), device=dev) * 0.02)
            ups = "up4"
        y, _i = prog.conv(x, w, b, Cout, ksize=k, stride=stride, pad=k // 2, upsample=ups, gn=gn, out_nchw_f32=name.startswith("head"))
        if y is None:  # head: the caller binds the NCHW fp32 output
            out = torch.empty((N, Cout, H, W), dtype=torch.float32, device=dev)
            prog.ops[_i].u.conv.y = out.data_ptr()
        prog.finalize()
        for tpw in ([int(v) for v in a.tpw.split(",")] if a.tpw else [None]):
            if tpw is not None:
                prog.L.eod_set_option(b"halo_tpw", tpw)
            for _ in range(3): prog.run()
            torch.cuda.synchronize()
            # HIP events around the conv op only (the program may hold a statistics / bound-table / normalising op in front of it)
            prog.enable_timing(a.iters, only=[_i])
            for _ in range(a.iters): prog.run()
            torch.cuda.synchronize()
            runs, ms = prog.read_timing()
            prog.disable_timing()
            dt = ms[_i] / runs * 1e-3
            fl = 2.0 * N * (y.H * y.W if y is not None else H * W) * Cout * Cin * k * k
            print(f"{name:8s} {a.prec}{' gn' if gn else ''}{' presplit' if x.presplit else ''}{'' if tpw is None else f' tpw {tpw:2d}'} {dt*1e3:8.3f} ms  {fl/dt/1e12:8.1f} TF/s", flush=True)

if __name__ == "__main__":
    main()
