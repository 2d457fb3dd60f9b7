#!/usr/bin/env python3
"""bench.py -- denoising steps/s of the EODiffusion hot path on MI355X.

One *step* = one pass of the hot path over one synthetic batch: Philox noise generation ->
UNetModel forward (eps prediction) -> fused DDPM reverse update, i.e. one iteration of
EODiffusion.sampling (model.py:54-69).  Workload at N=1 is the shape BASELINE.json's metric is quoted on:
256x256x3, batch 16, UNet base 128 / mults [1,2,3,4] / 1 res-block / no extra attention (train.py:50).
With --gpus N each rank runs the same per-GPU batch (config 4: 8 x 16 = 128 images; weak scaling), no
per-step communication; the final images are all-gathered once over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision fp32|fp16] [--arch A0|A1] [--size 256] [--batch 16]

The headline (`value`, `dtype`) is measured in the REFERENCE's precision: fp32 storage and arithmetic (the reference runs this path
in fp32 everywhere, SURVEY.md section 8 legend), rel-L2 <= 1e-5 vs the fp32 CPU oracle.  The fp16-storage / fp16-MFMA mode
(tolerance 5e-3) is timed in the same run and reported as the labelled secondary object `fp16`.

`--gpus N` with N > 1 and no torchrun environment: this process starts the N ranks itself (torch.distributed.run, one rank per
GPU, RCCL) BEFORE touching the GPU and relays their output; under the driver's own `python -m torch.distributed.run ... bench.py
--gpus N` launch the ranks read RANK / LOCAL_RANK / WORLD_SIZE from the environment.  A world size that differs from --gpus is
an error, never a silent single-rank run.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (MFMA implicit-GEMM conv kernel, per-launch
HIP-event timing on the launch stream inside the timed region; `roofline.hbm_bound` = the HBM-bound kernels of the step -- first conv,
head conv, GroupNorm apply, Philox noise, DDPM update -- each with its algorithmic bytes, duration and GB/s against the 8 TB/s peak),
`cpu_baseline` (the CPU oracle timed on the host cores on a bounded sample of the same workload, plus BASELINE config 1 -- MNIST,
T = 200 -- end to end), and short labelled runs of the other BASELINE configurations' step: `config2` (A0 @ 64 x 64, batch 16),
`config3` (A1 @ 256 x 256, batch 8, masked DDIM step) and `train_fp16` (the training step of config 5's path at A0 @ 256 x 256, batch 16).
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
# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md.  fp32x3 issues fp16 MFMAs (three per algorithmic product), so it is
# priced against the fp16 matrix peak: `frac` = ALGORITHMIC flops / 2.5 PFLOP/s can reach at most 1/3.
PEAK = {"fp32": 157.3e12, "fp32x3": 2.5e15, "fp16": 2.5e15}
PEAK_NOTE = {"fp32x3": "fp32 storage; each product = 3 fp16 MFMAs (v_mfma_f32_16x16x32_f16) on split (hi + lo) operands, so the ceiling "
                       "of frac (algorithmic flops / fp16 MFMA peak) is 1/3; executed MFMA flops = 3 x achieved"}
DTYPE_NAME = {"fp32": "f32", "fp32x3": "f32", "fp16": "f16"}
TOLERANCE = {"fp32": "rel-L2 <= 1e-5 per UNet forward vs the fp32 CPU oracle (fp32 storage, exact fp32 MFMA)",
             "fp32x3": "rel-L2 <= 1e-5 per UNet forward vs the fp32 CPU oracle -- the SAME gate as the exact-fp32 mode (fp32 storage; "
                       "every conv / attention product as three fp16 MFMAs on hi+lo split operands, ~2^-22 per product, fp32 accumulate; "
                       "operands scaled by a power of two per image, derived on the device from the tensor's bound table: valid for "
                       "inputs of any magnitude -- tests/test_gpu_fp32x3_domain.py)",
             "fp16": "rel-L2 <= 5e-3 per UNet forward vs the fp32 CPU oracle (fp16 storage, fp16 MFMA, fp32 accumulate)"}
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


def cpu_config1_mnist(seconds_cap=40.0):
    """BASELINE config 1 end to end on the host cores: MNIST-shaped unconditional DDPM (28x28x1, base 32, mults [2,4],
    scripts/train_mnist.py:46-48), T = 200, a complete `sampling()` call of the CPU oracle (x_T draw + 200 x (randn + UNet +
    clipped update)), batch 4 -- the case the reference itself can run without a GPU."""
    from eo_diffusion_amd.backbones.unet_openai import unet_param_shapes
    from oracle import sampler_ref, schedule, unet_ref
    from tests.synth import synth_state_dict
    cfg = dict(image_size=28, in_channels=1, out_channels=1, model_channels=32, channel_mult=[2, 4], attention_resolutions=[],
               num_res_blocks=1, num_heads=1)
    sd = synth_state_dict(unet_param_shapes(**cfg), 7)
    T, bs = 200, 4
    tb = schedule.eo_cosine_tables(T)
    g = torch.Generator().manual_seed(2)
    x = torch.randn((bs, 1, 28, 28), generator=g)
    done = 0
    with torch.no_grad():
        t0 = time.perf_counter()
        for i in range(T - 1, -1, -1):
            t = torch.full((bs,), i, dtype=torch.int64)
            noise = torch.randn(x.shape, generator=g)
            x = sampler_ref.ddpm_step_clip(tb, x, t, noise, unet_ref.unet_forward(sd, cfg, x, t))
            done += 1
            if time.perf_counter() - t0 > seconds_cap:
                break
        dt = time.perf_counter() - t0
    full = dt * T / done
    return {"images_per_s": bs / full, "steps_per_s": done / dt, "seconds_per_call": full, "timesteps": T, "batch": bs,
            "steps_timed": done, "finite": bool(torch.isfinite(x).all()),
            "what": "config 1: 28x28x1 unconditional DDPM, T=200, complete sampling call of the CPU oracle"
                    + ("" if done == T else f" (capped after {done} steps, extrapolated)")}


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
    out = {"value": steps_per_s_bs * bs / batch, "unit": "steps/s", "cores": cores, "kind": "port",
           "sample": f"{n} timed step(s) of the CPU oracle (UNet fwd + randn + DDPM update) at batch {bs} "
                     f"({size}x{size}, arch {arch}, {dt:.2f} s/step), scaled by {bs}/{batch} to batch {batch}; torch CPU fp32, "
                     f"{cores} threads"}
    try:
        out["config1_mnist_T200"] = cpu_config1_mnist()
    except Exception as e:  # the headline baseline above stands on its own
        out["config1_mnist_T200"] = {"error": repr(e)}
    return out


def parity_probe(m, arch, size):
    """In-run evidence that the timed precision mode computes the reference's function: ONE UNet forward of the benchmarked model at
    the metric's image size (batch 2, t = [999, 3], fixed inputs) against the CPU oracle evaluated on the SAME weights.  The oracle
    is the checker here, never the thing timed (its cost: two 256x256 images, ~4 s on the host cores)."""
    from oracle import unet_ref
    dev = next(m.parameters()).device
    g = torch.Generator().manual_seed(11)
    x = torch.randn((2, 3, size, size), generator=g)
    t = torch.tensor([999, 3])
    cfg = dict(image_size=size, in_channels=3, out_channels=3, **ARCHS[arch])
    with torch.no_grad():
        got = m.model(x.to(dev), t.to(dev)).float().cpu()
        sd = {k: v.detach().float().cpu() for k, v in m.model.state_dict().items()}
        ref = unet_ref.unet_forward(sd, cfg, x, t)
    err = float((got.double() - ref.double()).norm() / ref.double().norm())
    return {"rel_l2_vs_cpu_oracle": err, "what": f"UNet forward, {size}x{size}, batch 2, t = [999, 3], same weights, precision mode "
            f"{m.model.precision}", "probe_output": got}


def is_hbm_bound_op(s):
    """program ops whose roof is HBM: the thin-input first conv, the 3-channel head conv, GroupNorm apply passes"""
    if s["kind"] == "gn_apply":
        return True
    return s["kind"] == "conv" and (s.get("kernel") == "conv_head_kernel" or "256x256 4->" in s["label"] or s["label"].endswith("->3"))


def sampler_kernels_hbm(m, shape, dev, iters=20):
    """eod_randn_philox and eod_ddpm_step, back to back between two events on the stream they are launched on (torch's current stream)"""
    n = 1
    for d in shape:
        n *= d
    x = m._philox(shape, dev, 2, 0, 5, 0)
    t = torch.full((shape[0],), 500, dtype=torch.int64, device=dev)
    out = []
    for name, fn, nbytes in (("randn_philox", lambda: m._philox(shape, dev, 3, 0, 7, 1), 4 * n),
                             ("ddpm_step", lambda: m._ddpm_update(x, x, x, t, clip=True), 4 * 4 * n)):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        sec = e0.elapsed_time(e1) / iters * 1e-3
        out.append({"kernel": name, "op": f"{shape[0]}x{shape[1]}x{shape[2]}x{shape[3]} fp32", "bytes_algorithmic": nbytes, "ms": sec * 1e3,
                    "GB/s": nbytes / sec / 1e9, "frac": nbytes / sec / HBM_PEAK})
    return out


def short_config_runs(dev, steps=5, warmup=3):
    """labelled step times of the other BASELINE configurations (driver-run, a few steps each): config 2 = A0 @ 64 x 64, batch 16,
    DDPM step; config 3 = A1 @ 256 x 256, batch 8, one masked (RePaint) DDIM step of a 250-step schedule, through the sampler's own loop"""
    out = {}
    with torch.no_grad():
        m = build_model("A0", 64, "fp32x3", dev)
        shape = (16, 3, 64, 64)
        x = m._philox(shape, dev, 2, 0, m.timesteps, 0)

        def step(x, i):
            noise = m._philox(shape, dev, 3, 0, i, 1)
            t = torch.full((16,), i, dtype=torch.int64, device=dev)
            return m._ddpm_update(x, m.model(x, t), noise, t, clip=True)
        c2steps = 20 * steps  # (3 ms steps: 100 of them, so that the host's launch jitter does not show)
        for i in range(4 * warmup):
            x = step(x, 999 - i)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(c2steps):
            x = step(x, 900 - i)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / c2steps
        out["config2"] = {"workload": "A0 @ 64x64x3, batch 16, DDPM step (Philox noise + UNet + clipped update), fp32x3", "steps": c2steps,
                          "ms_per_step": dt * 1e3, "steps_per_s": 1.0 / dt, "achieved_tflops": 642.1e9 / dt / 1e12,
                          "frac": 642.1e9 / dt / PEAK["fp32x3"], "outputs_finite": bool(torch.isfinite(x).all())}
        del m, x
        torch.cuda.empty_cache()
        from eo_diffusion_amd.diffusion.ddim import DDIMSampler
        m = build_model("A1", 256, "fp32x3", dev)
        smp = DDIMSampler(m)
        g = torch.Generator().manual_seed(4)
        gt = torch.rand((8, 3, 256, 256), generator=g).to(dev)
        mask = torch.ones((8, 1, 256, 256))
        mask[:, :, 64:160, 96:200] = 0.0
        mask = mask.to(dev)
        S = 250
        per = []
        c3steps = 6 * steps  # (the call's fixed part -- schedule tables, x_T draw, ~7 ms -- over 30 steps instead of 5: the complete 250-step
        # call measures 24.9 ms per step, tools/full_ddim_repaint.py)
        for rep, nst in ((0, warmup), (1, c3steps)):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            # `nst` steps of the 250-step schedule (timesteps = n keeps the first n - 1 entries of the table, ddim.py:126-128)
            xs, _ = smp.sample(S, 8, (3, 256, 256), eta=0.0, mask=mask, x0=gt, verbose=False, timesteps=nst + 1, progress=False)
            torch.cuda.synchronize(dev)
            per.append((time.perf_counter() - t0) / nst)
        out["config3"] = {"workload": "A1 @ 256x256x3, batch 8, masked (RePaint) DDIM step of a 250-step schedule through DDIMSampler.sample "
                                      "(schedule tables + x_T draw included in the call), fp32x3", "steps": c3steps, "ms_per_step": per[1] * 1e3,
                          "steps_per_s": 1.0 / per[1], "achieved_tflops": 8818.7e9 / per[1] / 1e12, "frac": 8818.7e9 / per[1] / PEAK["fp32x3"],
                          "outputs_finite": bool(torch.isfinite(xs).all())}
        del m, smp, xs
        torch.cuda.empty_cache()
    return out


def short_train_run(dev, steps=5, warmup=2):
    """the training step of config 5's path (fp16 storage, fp16 MFMA) at A0 @ 256 x 256, batch 16: forward + MSE + backward + fused AdamW"""
    from eo_diffusion_amd.optim import AdamW, mse_loss
    from eo_diffusion_amd.training import UNetTrainer
    m = build_model("A0", 256, "fp16", dev)
    unet = m.model.train()
    opt = AdamW(unet.parameters(), lr=1e-4)
    tr = UNetTrainer(unet, 16, 256, 256, dev, loss_scale=1024.0)
    g = torch.Generator(device=dev).manual_seed(100)
    x = torch.rand((16, 3, 256, 256), device=dev, generator=g)
    noise = torch.randn((16, 3, 256, 256), device=dev, generator=g)
    t = torch.randint(0, m.timesteps, (16,), device=dev, generator=g)

    def one():
        pred = tr.forward(m._forward_diffusion(x, t, noise), t)
        loss, dpred = mse_loss(pred, noise)
        tr.backward(dpred, allreduce=False)
        opt.step()
        return loss
    for _ in range(warmup):
        loss = one()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    fl = 3.0 * sum(o.get("flops", 0.0) for o in tr.prog.op_stats())
    res = {"workload": "training step, A0 @ 256x256x3, batch 16: q_sample + UNet forward + MSE + backward + fused AdamW, fp16 storage / fp16 MFMA, "
                       "fp32 accumulate and master weights", "steps": steps, "ms_per_step": dt * 1e3, "steps_per_s": 1.0 / dt,
           "images_per_s": 16.0 / dt, "achieved_tflops": fl / dt / 1e12, "frac": fl / dt / PEAK["fp16"], "loss_finite": bool(torch.isfinite(loss).all())}
    del tr, opt, m
    torch.cuda.empty_cache()
    return res


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a torchrun environment: start the N ranks as CHILD processes (one per GPU, RCCL) and relay
    their output.  Runs before this process has made any GPU call (a process that has initialised the GPU must never exec)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), EOD_BENCH_CHILD="1")
    return subprocess.run(cmd, env=env).returncode


def time_sampling(args, m, dev, rank, world, use_dist, dist, timing=True):
    """warm-up, then EXACTLY args.steps denoising steps between two barrier + synchronize; returns (seconds [max over ranks],
    final x_t, roofline dict or None)"""
    N, S = args.batch, args.size
    shape = (N, 3, S, S)
    seed, sample0 = 3, rank * N
    stub = m is None
    if stub:  # CPU rehearsal of the launch / barrier / reduction plumbing (tests/test_dist_cpu.py): no GPU, no kernels
        x_t = torch.zeros(shape)

        def one_step(x_t, i):
            return x_t + 1.0

        def sync():
            pass
    else:
        x_t = m._philox(shape, dev, 2, sample0, m.timesteps, 0)

        def one_step(x_t, i):
            noise = m._philox(shape, dev, seed, sample0, i, 1)
            t = torch.full((N,), i, dtype=torch.int64, device=dev)
            pred = m.model(x_t, t)
            return m._ddpm_update(x_t, pred, noise, t, clip=True)

        def sync():
            torch.cuda.synchronize(dev)

    def barrier():
        sync()
        if use_dist:
            dist.barrier()
            sync()

    T = 1000 if stub else m.timesteps
    prog = None
    with torch.no_grad():
        i = T - 1
        for _ in range(args.warmup):
            x_t = one_step(x_t, i)
            i -= 1
        timing = timing and not stub and not args.no_op_timing
        if timing:
            prog = m.model.program_for(N, 3, 0, S, S, dev, False)
            # HIP events only around the dominant kernel's launches (an event pair idles the stream for ~8 us; bracketing
            # all ~130 ops of a step would cost ~1.3 ms/step).  --dump-ops times every op instead.
            stats0 = prog.op_stats()
            only = None if args.dump_ops else [k for k, s in enumerate(stats0) if s.get("kernel") in ("conv3x3_halo_kernel", "conv_up4_halo_kernel")
                                               or is_hbm_bound_op(s)]
            prog.enable_timing(args.steps, only=only)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            x_t = one_step(x_t, i)
            i = i - 1 if i > 0 else T - 1
        gathered = None
        if use_dist:  # the one collective of the path: gather the final images
            gathered = torch.empty((world * N, 3, S, S), dtype=torch.float32, device=x_t.device)
            dist.all_gather_into_tensor(gathered, x_t.contiguous())
        barrier()
        dt = time.perf_counter() - t0
    if gathered is not None:  # (outside the timed region) the collective really delivered this rank's shard
        ones = torch.ones((1,), dtype=torch.float32, device=x_t.device)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        time_sampling.collective = {"op": "all_gather_into_tensor", "backend": dist.get_backend(), "world": world,
                                    "ranks_seen": int(ones.item()),
                                    "bytes_per_rank": x_t.numel() * 4, "device": str(gathered.device),
                                    "own_shard_bit_equal": bool(torch.equal(gathered[rank * N:(rank + 1) * N], x_t))}
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=x_t.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    roof = None
    if timing:
        prec = m.model.precision
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
        peak = PEAK[prec]
        up4 = [(s, t) for s, t in allconv if s.get("kernel") == "conv_up4_halo_kernel"]
        roof = {"bound": "mfma", "kernel": "conv3x3_halo_kernel (3x3 stride-1 convs of one UNet forward)",
                "achieved": fl / tsec / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                "frac": fl / tsec / peak, "traffic": None,
                "launches_per_step": nl, "avg_launch_ms": tsec * 1e3 / nl, "algorithmic_gflop_per_launch_avg": fl / nl / 1e9,
                "kernel_ms_per_step": tsec * 1e3,
                "kernel_share_of_step": tsec / (dt / args.steps)}
        if prec in PEAK_NOTE:
            roof["peak_note"] = PEAK_NOTE[prec]
        nskip = sum(1 for s, _ in conv if "+skip1x1" in s["label"])
        if nskip:  # ResBlocks whose 1x1 skip conv runs inside out_layers' conv: those launches carry its FLOPs too
            roof["launches_with_fused_1x1_skip"] = nskip
        if up4:
            # the sibling kernel of the convs behind a nearest-2x upsampling: the algorithm's nine taps are executed as four pre-summed
            # ones per output parity class, so its ALGORITHMIC rate (the reference's flop count) exceeds what its MFMAs execute
            usec = sum(t for _, t in up4) / runs * 1e-3
            ufl, uex = sum(s["flops"] for s, _ in up4), sum(s["exec_flops"] for s, _ in up4)
            roof["sibling_kernel"] = {"kernel": "conv_up4_halo_kernel (3x3 convs over a nearest-2x upsampling, parity-class form)",
                                      "launches_per_step": len(up4), "kernel_ms_per_step": usec * 1e3,
                                      "algorithmic_tflops": ufl / usec / 1e12, "executed_tflops": uex / usec / 1e12,
                                      "frac_executed": uex / usec / peak, "kernel_share_of_step": usec / (dt / args.steps)}
        # the HBM-bound kernels of the step against the HBM peak (north_star: "HBM GB/s ... against CDNA4 peak"): program ops from the same
        # HIP events, the two sampler kernels from event pairs of their own around back-to-back launches (outside the timed region)
        hb = []
        for s_, t_ in zip(stats, ms):
            if is_hbm_bound_op(s_) and t_ > 0:
                sec = t_ / runs * 1e-3
                hb.append({"kernel": s_.get("kernel") or s_["kind"], "op": s_["label"], "bytes_algorithmic": s_["bytes"], "ms": sec * 1e3,
                           "GB/s": s_["bytes"] / sec / 1e9, "frac": s_["bytes"] / sec / HBM_PEAK})
        hb += sampler_kernels_hbm(m, (N, 3, S, S), dev)
        roof["hbm_bound"] = hb
        if args.dump_ops:
            roof["all_conv_launches"] = {"launches_per_step": len(allconv), "ms_per_step": asec * 1e3,
                                         "achieved_tflops": afl / asec / 1e12, "frac": afl / asec / peak}
            roof["all_ops_ms_per_step"] = sum(ms) / runs
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                prof = json.load(open(tf))
                key = f"{args.arch}_{args.size}_{args.batch}_{prec}"
                if key in prof:
                    # NOT measured in this run: HBM bytes per launch from a committed rocprofv3 --pmc pass over the same command
                    # (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE); file and commit of origin are named in it
                    roof["traffic"] = prof[key]["bytes_per_launch"]
                    roof["traffic_unit"] = "bytes per launch (algorithmic: %.0f)" % prof[key]["algorithmic_bytes_per_launch"]
                    roof["traffic_from_profile"] = prof[key]
            except Exception:
                pass
        if args.dump_ops and rank == 0:
            rows = [dict(s, ms=t / runs) for s, t in zip(stats, ms)]
            with open(args.dump_ops if prec == args.precision else args.dump_ops + "." + prec, "w") as f:
                json.dump(rows, f, indent=1)
    return dt, x_t, roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="fp32x3", choices=list(PEAK),
                    help="headline precision mode.  Default fp32x3: fp32 storage and fp32-grade arithmetic (every conv product as three "
                         "fp16 MFMAs on hi+lo split operands; held to the same 1e-5 gate as the exact-fp32 mode, measured error below "
                         "it).  The exact-fp32-MFMA mode and the fp16 mode are timed in the same run as labelled secondary objects")
    ap.add_argument("--arch", default="A0", choices=list(ARCHS))
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--graph", action="store_true", help="replay the UNet program as one hipGraph launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary (fp16) measurement")
    ap.add_argument("--no-op-timing", action="store_true")
    ap.add_argument("--dump-ops", default=None, help="write the per-op timing table (JSON) here")
    ap.add_argument("--train", action="store_true", help="time the TRAINING step (forward + MSE + backward + fused AdamW [+ gradient "
                    "all-reduce for N > 1]) instead of the sampling step; SURVEY section 8f rank 1 / BASELINE config 5")
    ap.add_argument("--in-ch", type=int, default=3)
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    stub = os.environ.get("EOD_BENCH_STUB") == "1"  # CPU / gloo rehearsal of the multi-rank plumbing (no GPU work)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))  # nothing above has touched the GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    use_dist = world > 1 or os.environ.get("EOD_BENCH_FORCE_DIST") == "1"  # (FORCE: rehearse the RCCL path on one GPU)
    dist = None
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("gloo" if stub else "nccl")  # "nccl" = RCCL on ROCm
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    if stub:
        dev = torch.device("cpu")
    else:
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)

    if args.train:
        return train_main(args, world, rank, dev, use_dist)
    N, S = args.batch, args.size
    m = None if stub else build_model(args.arch, S, args.precision, dev)
    if args.graph and m is not None:
        m.model.enable_graph(True)
        args.no_op_timing = True  # the event-bracketing executor cannot be captured
    dt, x_t, roof = time_sampling(args, m, dev, rank, world, use_dist, dist)
    finite = bool(torch.isfinite(x_t).all())
    parity = None
    if not stub and world == 1 and not args.no_cpu_baseline and not args.graph:
        try:
            parity = parity_probe(m, args.arch, S)
        except Exception as e:  # the measurement stands on its own; say why the probe is missing
            parity = {"error": repr(e)}

    secondary = {}
    if not stub and not args.no_secondary:
        # the other precision modes, same workload, same run: labelled extras, never the headline
        del m, x_t
        for prec in ("fp32", "fp16"):
            if prec == args.precision:
                continue
            torch.cuda.empty_cache()
            m2 = build_model(args.arch, S, prec, dev)
            dt2, x2, roof2 = time_sampling(args, m2, dev, rank, world, use_dist, dist)
            entry = {"value": world * args.steps / dt2, "unit": "steps/s", "ms_per_step": dt2 / args.steps * 1e3, "dtype": DTYPE_NAME[prec],
                     "precision_mode": prec, "tolerance": TOLERANCE[prec], "outputs_finite": bool(torch.isfinite(x2).all()), "roofline": roof2}
            if parity and "probe_output" in parity:  # same seed -> same weights: how far is this mode from the headline mode's output?
                with torch.no_grad():
                    g = torch.Generator().manual_seed(11)
                    xp = torch.randn((2, 3, S, S), generator=g).to(dev)
                    o2 = m2.model(xp, torch.tensor([999, 3], device=dev)).float().cpu()
                ref = parity["probe_output"]
                entry["rel_l2_vs_headline_mode_output"] = float((o2.double() - ref.double()).norm() / ref.double().norm())
            secondary[{"fp32": "fp32_exact_mfma", "fp16": "fp16"}[prec]] = entry
            del m2, x2

    if rank == 0:
        steps_per_s = world * args.steps / dt
        res = {
            "metric": "denoising steps/sec (UNet fwd + DDPM update) at 256x256 bs=16",
            "value": steps_per_s, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_NAME[args.precision], "data": "synthetic",
            "config": {"workload": f"DDPM sampling step, {S}x{S}x3, batch {N} per GPU, UNet arch {args.arch} "
                                   f"(base 128, mults [1,2,3,4], {ARCHS[args.arch]['num_res_blocks']} res-block(s), "
                                   f"attn_res {ARCHS[args.arch]['attention_resolutions']}), Philox noise, x0-clipped update",
                       "global_batch": world * N, "image_size": S, "parallelism": f"batch-sharded x{world}",
                       "accumulate": "fp32", "precision_mode": args.precision, "tolerance": TOLERANCE[args.precision]},
            "images_per_sec_1000step_ddpm": world * N / (1000.0 * dt / args.steps),
            "outputs_finite": finite,
        }
        if stub:
            res["stub"] = True
        if getattr(time_sampling, "collective", None):
            res["collective"] = time_sampling.collective
        if parity:
            parity.pop("probe_output", None)
            res["parity_probe"] = parity
        if roof:
            res["roofline"] = roof
        res.update(secondary)
        if world == 1 and not stub and not args.no_secondary and (args.arch, S, N) == ("A0", 256, 16):
            for name, fn in (("configs", lambda: short_config_runs(dev)), ("train_fp16", lambda: short_train_run(dev))):
                try:
                    r_ = fn()
                    res.update(r_ if name == "configs" else {name: r_})
                except Exception as e:  # the headline stands on its own; say why an extra is missing
                    res[name] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1 and not stub:
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
    opt = AdamW(unet.parameters(), lr=1e-4)  # first: it moves the parameters into its flat buffer (the trainer bakes their pointers)
    tr = UNetTrainer(unet, N, S, S, dev, loss_scale=(1024.0 if args.precision == "fp16" else 1.0))
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
               "dtype": DTYPE_NAME[args.precision], "data": "synthetic",
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
