#!/usr/bin/env python3
"""bench.py -- denoising steps/s of the EODiffusion hot path on MI355X.

One *step* = one pass of the hot path over one synthetic batch: Philox noise generation ->
UNetModel forward (eps prediction) -> fused DDPM reverse update, i.e. one iteration of
EODiffusion.sampling (model.py:54-69).  Workload at N=1 is the shape BASELINE.json's metric is quoted on:
256x256x3, batch 16, UNet base 128 / mults [1,2,3,4] / 1 res-block / no extra attention (train.py:50).
With --gpus N each rank runs the same per-GPU batch (config 4: 8 x 16 = 128 images; weak scaling), no
per-step communication; the final images are all-gathered once over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision fp16|fp32] [--arch A0|A1] [--size 256] [--batch 16]

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (MFMA implicit-GEMM conv kernel, per-launch
HIP-event timing on the launch stream inside the timed region) and `cpu_baseline` (the CPU oracle timed on the
host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

ARCHS = {
    "A0": dict(model_channels=128, channel_mult=[1, 2, 3, 4], attention_resolutions=[], num_res_blocks=1, num_heads=1),
    "A1": dict(model_channels=128, channel_mult=[1, 2, 3, 4], attention_resolutions=[4, 8], num_res_blocks=2, num_heads=8),
}
# algorithmic GFLOP per denoising step, whole batch (conv 2*N*Cout*Ho*Wo*Cin*k*k, linear 2*N*in*out,
# attention 4*N*T^2*C): A0@256/16 = 10303.7
PEAK = {"fp16": 2.5e15, "fp32": 157.3e12}  # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8.0e12


def build_model(arch, size, precision, dev, timesteps=1000, in_ch=3):
    from eo_diffusion_amd.backbones.unet_openai import UNetModel
    from eo_diffusion_amd.diffusion.model import EODiffusion
    torch.manual_seed(0)
    u = UNetModel(size, in_channels=in_ch, out_channels=in_ch, **ARCHS[arch]).set_precision(precision)
    # the reference zero-initialises every ResBlock's second conv, proj_out and the out conv (zero_module):
    # re-draw them so the benchmark does real arithmetic end to end (values do not change the work done)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in u.parameters():
            if p.dim() > 1 and float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    m = EODiffusion(u, timesteps=timesteps, image_size=size, in_channels=in_ch, device=str(dev)).to(dev).eval()
    return m


def host_cores():
    """CPU cores this process may actually use: affinity mask, capped by the cgroup CPU quota (the GPU box
    gives one GPU's share of the host, 16 cores, while sched_getaffinity still reports all 256)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return min(n, int(os.environ.get("EOD_CPU_THREADS", "16")))


def cpu_baseline(arch, size, batch, seconds_budget=25.0):
    """Oracle (CPU restatement of the reference, oracle/) timed on the host cores: bounded sample."""
    from eo_diffusion_amd.backbones.unet_openai import unet_param_shapes
    from oracle import sampler_ref, schedule, unet_ref
    from tests.synth import synth_state_dict
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = dict(image_size=size, in_channels=3, out_channels=3, **ARCHS[arch])
    sd = synth_state_dict(unet_param_shapes(**cfg), 7)
    tb = schedule.eo_cosine_tables(1000)
    bs = min(batch, 2)
    x = torch.randn(bs, 3, size, size)
    t = torch.full((bs,), 500, dtype=torch.int64)

    def step(x):
        noise = torch.randn_like(x)
        eps = unet_ref.unet_forward(sd, cfg, x, t)
        return sampler_ref.ddpm_step_clip(tb, x, t, noise, eps)

    with torch.no_grad():
        t0 = time.perf_counter()
        x = step(x)  # warm-up (also a first estimate)
        warm = time.perf_counter() - t0
        n = max(1, min(4, int(seconds_budget / max(warm, 1e-3)) - 1))
        t0 = time.perf_counter()
        for _ in range(n):
            x = step(x)
        dt = (time.perf_counter() - t0) / n
    steps_per_s_bs = 1.0 / dt
    return {"value": steps_per_s_bs * bs / batch, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} timed step(s) of the CPU oracle (UNet fwd + randn + DDPM update) at batch {bs} "
                      f"({size}x{size}, arch {arch}, {dt:.2f} s/step), scaled by {bs}/{batch} to batch {batch}; torch CPU fp32, "
                      f"{cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--arch", default="A0", choices=list(ARCHS))
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--graph", action="store_true", help="replay the UNet program as one hipGraph launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-op-timing", action="store_true")
    ap.add_argument("--dump-ops", default=None, help="write the per-op timing table (JSON) here")
    ap.add_argument("--train", action="store_true", help="time the TRAINING step (forward + MSE + backward + fused AdamW [+ gradient "
                    "all-reduce for N > 1]) instead of the sampling step; SURVEY section 8f rank 1 / BASELINE config 5")
    ap.add_argument("--in-ch", type=int, default=3)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("EOD_BENCH_FORCE_DIST") == "1"  # (FORCE: rehearse the RCCL path on one GPU)
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl")  # RCCL on ROCm
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    if args.train:
        return train_main(args, world, rank, dev, use_dist)
    m = build_model(args.arch, args.size, args.precision, dev)
    if args.graph:
        m.model.enable_graph(True)
        args.no_op_timing = True  # the event-bracketing executor cannot be captured
    N, S = args.batch, args.size
    shape = (N, 3, S, S)
    seed, sample0 = 3, rank * N
    x_t = m._philox(shape, dev, 2, sample0, m.timesteps, 0)

    def one_step(x_t, i):
        noise = m._philox(shape, dev, seed, sample0, i, 1)
        t = torch.full((N,), i, dtype=torch.int64, device=dev)
        pred = m.model(x_t, t)
        return m._ddpm_update(x_t, pred, noise, t, clip=True)

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    with torch.no_grad():
        i = m.timesteps - 1
        for _ in range(args.warmup):
            x_t = one_step(x_t, i)
            i -= 1
        prog = m.model.program_for(N, 3, 0, S, S, dev, False)
        timing = not args.no_op_timing
        if timing:
            # HIP events only around the dominant kernel's launches (an event pair idles the stream for ~8 us; bracketing
            # all ~130 ops of a step would cost ~1.3 ms/step).  --dump-ops times every op instead.
            stats0 = prog.op_stats()
            only = None if args.dump_ops else [k for k, s in enumerate(stats0) if s.get("kernel") == "conv3x3_halo_kernel"]
            prog.enable_timing(args.steps, only=only)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            x_t = one_step(x_t, i)
            i = i - 1 if i > 0 else m.timesteps - 1
        if use_dist:  # the one collective of the path: gather the final images
            out = torch.empty((world * N, 3, S, S), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(out, x_t.contiguous())
        barrier()
        dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    finite = bool(torch.isfinite(x_t).all())

    roof = None
    if timing:
        runs, ms = prog.read_timing()
        stats = prog.op_stats()
        prog.disable_timing()
        allconv = [(s, t) for s, t in zip(stats, ms) if s["kind"] == "conv"]
        conv = [(s, t) for s, t in allconv if s.get("kernel") == "conv3x3_halo_kernel"]  # the dominant kernel
        fl = sum(s["flops"] for s, _ in conv)            # algorithmic flops of its launches, per step
        tsec = sum(t for _, t in conv) / runs * 1e-3     # their summed duration per step (HIP events on the launch stream)
        nl = len(conv)
        afl = sum(s["flops"] for s, _ in allconv)
        asec = sum(t for _, t in allconv) / runs * 1e-3
        roof = {"bound": "mfma", "kernel": "conv3x3_halo_kernel (3x3 stride-1 convs of one UNet forward, incl. virtual-2x-upsample ones)",
                "achieved": fl / tsec / 1e12, "peak": PEAK[args.precision] / 1e12, "unit": "TFLOP/s",
                "frac": fl / tsec / PEAK[args.precision], "traffic": None,
                "launches_per_step": nl, "avg_launch_ms": tsec * 1e3 / nl, "algorithmic_gflop_per_launch_avg": fl / nl / 1e9,
                "kernel_ms_per_step": tsec * 1e3,
                "kernel_share_of_step": tsec / (dt / args.steps)}
        if args.dump_ops:
            roof["all_conv_launches"] = {"launches_per_step": len(allconv), "ms_per_step": asec * 1e3,
                                         "achieved_tflops": afl / asec / 1e12, "frac": afl / asec / PEAK[args.precision]}
            roof["all_ops_ms_per_step"] = sum(ms) / runs
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                prof = json.load(open(tf))
                roof["traffic"] = prof.get(f"{args.arch}_{args.size}_{args.batch}_{args.precision}")
                if args.precision == "fp16" and "_pmc_mfma" in prof:
                    # PMC evidence from the committed rocprofv3 pass (not measured in this run): the MFMA pipes are busy
                    # this fraction of the GPU cycles; the clock the chip sustains under this load is far below 2.4 GHz
                    roof["pmc_mfma"] = prof["_pmc_mfma"]
            except Exception:
                pass
        if args.dump_ops and rank == 0:
            rows = [dict(s, ms=t / runs) for s, t in zip(stats, ms)]
            with open(args.dump_ops, "w") as f:
                json.dump(rows, f, indent=1)

    if rank == 0:
        steps_per_s = world * args.steps / dt
        res = {
            "metric": "denoising steps/sec (UNet fwd + DDPM update) at 256x256 bs=16",
            "value": steps_per_s, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if args.precision == "fp16" else "f32", "data": "synthetic",
            "config": {"workload": f"DDPM sampling step, {S}x{S}x3, batch {N} per GPU, UNet arch {args.arch} "
                                   f"(base 128, mults [1,2,3,4], {ARCHS[args.arch]['num_res_blocks']} res-block(s), "
                                   f"attn_res {ARCHS[args.arch]['attention_resolutions']}), Philox noise, x0-clipped update",
                       "global_batch": world * N, "image_size": S, "parallelism": f"batch-sharded x{world}",
                       "accumulate": "fp32"},
            "images_per_sec_1000step_ddpm": world * N / (1000.0 * dt / args.steps),
            "outputs_finite": finite,
        }
        if roof:
            res["roofline"] = roof
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args.arch, S, N)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


def train_main(args, world, rank, dev, use_dist):
    """one training step per "step": q_sample -> UNet forward (activations kept) -> MSE -> backward -> [all-reduce] -> AdamW"""
    import torch.distributed as dist
    from eo_diffusion_amd.optim import AdamW, mse_loss
    from eo_diffusion_amd.training import UNetTrainer
    m = build_model(args.arch, args.size, args.precision, dev, in_ch=args.in_ch)
    unet = m.model.train()
    N, S, C = args.batch, args.size, args.in_ch
    tr = UNetTrainer(unet, N, S, S, dev, loss_scale=(1024.0 if args.precision == "fp16" else 1.0))
    opt = AdamW(unet.parameters(), lr=1e-4)
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    x = torch.rand((N, C, S, S), device=dev, generator=g)
    noise = torch.randn((N, C, S, S), device=dev, generator=g)
    t = torch.randint(0, m.timesteps, (N,), device=dev, generator=g)

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def one_step():
        x_t = m._forward_diffusion(x, t, noise)
        pred = tr.forward(x_t, t)
        loss, dpred = mse_loss(pred, noise)
        tr.backward(dpred, allreduce=True)  # bucketed gradient all-reduce overlapped with the backward (no-op for one rank)
        opt.step()
        return loss

    for _ in range(args.warmup):
        loss = one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        res = {"metric": f"training steps/sec (UNet fwd + MSE + bwd + AdamW) at {S}x{S} bs={N}", "value": world * args.steps / dt,
               "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f16" if args.precision == "fp16" else "f32", "data": "synthetic",
               "config": {"workload": f"training step, {S}x{S}x{C}, batch {N} per GPU, UNet arch {args.arch}, MSE(eps) loss, fused AdamW, "
                                      f"loss scale {tr.loss_scale:g}", "global_batch": world * N, "image_size": S,
                          "parallelism": f"data-parallel x{world} (one flat-bucket gradient all-reduce per step)", "accumulate": "fp32"},
               "images_per_sec": world * N * args.steps / dt, "loss": float(loss), "loss_finite": bool(torch.isfinite(loss).all()),
               # algorithmic FLOPs of one training step = 3 x the forward's (backward-data + backward-weights), per GPU
               "roofline": {"bound": "mfma", "achieved": 3.0 * sum(o.get("flops", 0.0) for o in tr.prog.op_stats()) / (dt / args.steps) / 1e12,
                            "peak": PEAK[args.precision] / 1e12, "unit": "TFLOP/s",
                            "frac": 3.0 * sum(o.get("flops", 0.0) for o in tr.prog.op_stats()) / (dt / args.steps) / PEAK[args.precision],
                            "traffic": None, "kernel": "whole training step (forward + backward-data + backward-weights)"},
               "hbm_gib": {"forward": tr.prog.nbytes / 2**30, "backward": tr.bprog.nbytes / 2**30}}
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
