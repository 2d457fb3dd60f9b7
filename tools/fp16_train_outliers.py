#!/usr/bin/env python3
"""fp16 TRAINING on degenerate nets: is the error inherent to fp16 storage?  

A hunt over random small UNets (tests/test_gpu_fuzz_archs.py::_train_case) found fp16-storage training steps whose worst parameter
gradient is 1e-2 ... 1e-1 off the fp32 CPU oracle -- nets whose deepest level is 2 x 2 pixels with one or two channels per group,
i.e. GroupNorms over 4-8 values.  This tool settles whether that is the kernels' doing:

  for every seed: gradients of   (a) the HIP path, fp16 storage          (what is tested)
                                 (b) the CPU oracle in fp32              (the reference, unet_openai.py run by torch autograd)
                                 (c) the SAME CPU oracle with every conv / GroupNorm(+SiLU) / attention output -- and the gradient that
                                     flows back through it -- rounded to fp16, i.e. the reference's arithmetic under fp16 STORAGE
  and reports, for the worst parameter of (a) vs (b):  err(a, b), err(c, b), err(a, c).

If err(c, b) is of the size of err(a, b), the deviation is what fp16 storage does to this net whoever computes it (1 / sigma of a tiny
group amplifies the rounding of its input); if err(c, b) is small, the kernels lose precision the storage format does not force.

    python tools/fp16_train_outliers.py [--first 0] [--count 220] [--gate 1e-2] [--out gpurun_out/fp16_train_outliers.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F


class _Round16(torch.autograd.Function):
    """value and gradient rounded to fp16 (storage emulation; arithmetic stays fp32 like the MFMA accumulators)"""

    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g.half().float()


def oracle_fp16_storage(UR):
    """context manager: oracle/unet_ref.py with fp16 storage emulated at the points the HIP path stores a tensor in training -- conv
    outputs, GroupNorm outputs (the SiLU behind one is part of the same stored tensor: rounding in front of it is the closest the
    functional oracle offers), attention outputs"""
    import contextlib

    @contextlib.contextmanager
    def cm():
        conv, gn, a1, a2 = UR._conv, UR.group_norm32, UR.qkv_attention_legacy, UR.qkv_attention_new
        UR._conv = lambda *a, **k: _Round16.apply(conv(*a, **k))
        UR.group_norm32 = lambda *a, **k: _Round16.apply(gn(*a, **k))
        UR.qkv_attention_legacy = lambda *a, **k: _Round16.apply(a1(*a, **k))
        UR.qkv_attention_new = lambda *a, **k: _Round16.apply(a2(*a, **k))
        try:
            yield
        finally:
            UR._conv, UR.group_norm32, UR.qkv_attention_legacy, UR.qkv_attention_new = conv, gn, a1, a2
    return cm()


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def study_case(i, gate):
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    from oracle import unet_ref as UR
    from tests.synth import synth_input, synth_state_dict
    from tests.test_gpu_fuzz_archs import _random_cfg, poison_allocator_cache
    DEV = "cuda:0"
    cfg, N, H, W, in_ch, cond_ch = _random_cfg(200 + i)
    if max(H, W) > 32:
        H = W = 32 if max(H, W) % 32 == 0 else 16
        cfg["image_size"] = H
        lv = len(cfg["channel_mult"])
        while H // (2 ** (lv - 1)) < 2 or H // (2 ** (lv - 1)) == 3:
            lv -= 1
        cfg["channel_mult"] = cfg["channel_mult"][:lv]
        cfg["attention_resolutions"] = [a for a in cfg["attention_resolutions"] if a < 2 ** lv]
    sd = synth_state_dict(unet_param_shapes(**cfg), 60 + i)
    x = synth_input(f"ft_x{i}", (N, in_ch, H, W), 61 + i)
    cond = synth_input(f"ft_c{i}", (N, cond_ch, H, W), 62 + i) if cond_ch else None
    noise = synth_input(f"ft_n{i}", (N, in_ch, H, W), 63 + i)
    t = torch.tensor([(53 * (i + 1) * (k + 1)) % 1000 for k in range(N)])
    y = torch.tensor([(i + k) % 5 for k in range(N)]) if "num_classes" in cfg else None

    def oracle_grads(fp16_storage):
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        if fp16_storage:
            with oracle_fp16_storage(UR):
                pred = UR.unet_forward(sdg, cfg, x, t, cond=cond, y=y)
                F.mse_loss(pred, noise).backward()
        else:
            pred = UR.unet_forward(sdg, cfg, x, t, cond=cond, y=y)
            F.mse_loss(pred, noise).backward()
        return {k: v.grad for k, v in sdg.items() if v.grad is not None}

    g32 = oracle_grads(False)
    poison_allocator_cache()
    u = UNetModel(**cfg).set_precision("fp16")
    u.load_state_dict(sd)
    u = u.to(DEV).train()
    pred = u(x.to(DEV), t.to(DEV), cond=cond.to(DEV) if cond is not None else None, y=y.to(DEV) if y is not None else None)
    F.mse_loss(pred, noise.to(DEV)).backward()
    torch.cuda.synchronize()
    gmax = max(float(v.norm()) for v in g32.values())
    worst = ("", 0.0)
    ghip = {}
    for name, p in u.named_parameters():
        if name in g32 and p.grad is not None and float(g32[name].norm()) >= 1e-5 * gmax:
            ghip[name] = p.grad.detach().float().cpu()
            e = rel(ghip[name], g32[name])
            if e > worst[1]:
                worst = (name, e)
    out = {"case": i, "cfg": cfg, "N": N, "H": H, "W": W, "deepest_map": [H // 2 ** (len(cfg["channel_mult"]) - 1), W // 2 ** (len(cfg["channel_mult"]) - 1)],
           "worst_param": worst[0], "err_hip_fp16_vs_oracle_fp32": worst[1]}
    if worst[1] > gate:
        g16 = oracle_grads(True)
        out["err_oracle_fp16_storage_vs_oracle_fp32"] = rel(g16[worst[0]], g32[worst[0]])
        out["err_hip_fp16_vs_oracle_fp16_storage"] = rel(ghip[worst[0]], g16[worst[0]])
        # the same three numbers for the oracle's own worst parameter under fp16 storage
        w16 = max(((k, rel(g16[k], g32[k])) for k in ghip), key=lambda kv: kv[1])
        out["oracle_fp16_storage_worst_param"] = w16[0]
        out["oracle_fp16_storage_worst_err"] = w16[1]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--count", type=int, default=220)
    ap.add_argument("--gate", type=float, default=1e-2)
    ap.add_argument("--cases", default="", help="comma list of case numbers instead of a range")
    ap.add_argument("--out", default="gpurun_out/fp16_train_outliers.json")
    a = ap.parse_args()
    cases = [int(v) for v in a.cases.split(",")] if a.cases else list(range(a.first, a.first + a.count))
    rows = []
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    for i in cases:
        r = study_case(i, a.gate)
        rows.append(r)
        with open(a.out + ".progress", "a") as pf:   # (a long run must keep writing: one line per case)
            pf.write(json.dumps([r["case"], r["err_hip_fp16_vs_oracle_fp32"]]) + "\n")
        if r["err_hip_fp16_vs_oracle_fp32"] > a.gate:
            print(f"case {i}: deepest map {r['deepest_map']}, {r['worst_param']}: HIP fp16 vs oracle fp32 {r['err_hip_fp16_vs_oracle_fp32']:.2e} | "
                  f"oracle under fp16 storage vs oracle fp32 {r['err_oracle_fp16_storage_vs_oracle_fp32']:.2e} (its own worst: "
                  f"{r['oracle_fp16_storage_worst_err']:.2e}) | HIP vs oracle under fp16 storage {r['err_hip_fp16_vs_oracle_fp16_storage']:.2e}", flush=True)
        elif i % 20 == 0:
            print(f"case {i}: {r['err_hip_fp16_vs_oracle_fp32']:.2e}", flush=True)
    out = [r for r in rows if r["err_hip_fp16_vs_oracle_fp32"] > a.gate]
    summary = {"cases_run": len(rows), "gate": a.gate, "outliers": out,
               "all_worst_errors": [[r["case"], r["err_hip_fp16_vs_oracle_fp32"]] for r in rows]}
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(summary, f, indent=1)
    print(f"{len(out)} of {len(rows)} cases beyond {a.gate:g}")


if __name__ == "__main__":
    main()
