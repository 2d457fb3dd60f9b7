"""CPU (gloo; world sizes 2, 3 -- ragged shards -- and 8, the node's size) tests of the sharding / gather logic of
eo_diffusion_amd.dist, the bucketed gradient average and the rank invariance of the Philox noise source (via its numpy oracle)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eo_diffusion_amd.dist import gather_samples, shard_bounds


def _free_port():
    """a port nobody listens on right now (a pid-derived number can collide with another process of the machine: the rendezvous
    then waits for its full timeout)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_ranks(target, args_of_rank, q, world=2):
    """start `world` daemon ranks, collect one queue item per rank, never leave a rank behind (a failed rendezvous must not outlive the test)"""
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=target, args=args_of_rank(r), daemon=True) for r in range(world)]
    try:
        for p in procs:
            p.start()
        res = dict(q.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        return res
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()


def test_shard_bounds_cover_and_disjoint():
    for n in (0, 1, 7, 16, 128, 129):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.philox_ref import philox_randn
        lo, hi = shard_bounds(n_total, world, rank)
        # each rank "samples" its shard with noise keyed by the GLOBAL sample index
        local = torch.from_numpy(philox_randn(hi - lo, 3 * 4 * 4, 11, lo, 5, 1)).reshape(hi - lo, 3, 4, 4)
        full = gather_samples(local, n_total)
        q.put((rank, full.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 4), (2, 5), (3, 7), (3, 9), (8, 16), (8, 13), (8, 5)])
def test_gloo_gather_matches_single_rank(world, n_total):
    """even and ragged shards, and (8 ranks, 5 samples) ranks that hold NO sample: every rank ends with the tensor one rank alone computes"""
    from oracle.philox_ref import philox_randn
    q = mp.get_context("spawn").Queue()
    port = _free_port()
    res = _run_ranks(_worker, lambda r: (r, world, port, n_total, q), q, world=world)
    ref = philox_randn(n_total, 48, 11, 0, 5, 1).reshape(n_total, 3, 4, 4)  # what ONE rank would have produced
    for r in range(world):
        assert np.array_equal(res[r], ref)


class _PhiloxModel:
    """stands in for EODiffusion in sharded_sampling: `sampling` returns noise keyed by the GLOBAL sample index plus what it was handed
    (its rows of cond and y), so the test sees the offsets, the slices and the gather -- not a UNet"""

    def sampling(self, n, clipped_reverse_diffusion=True, device=None, cond=None, y=None, rng=None, seed=0, sample_offset=0, progress=False):
        from oracle.philox_ref import philox_randn
        assert rng == "philox"
        out = torch.from_numpy(philox_randn(n, 12, seed, sample_offset, 0, 1)).reshape(n, 3, 2, 2)
        if cond is not None:
            assert cond.shape[0] == n
            out = out + cond
        if y is not None:
            assert y.shape == (n,)
            out = out + y.view(n, 1, 1, 1).float()
        return out


def _sharded_worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from eo_diffusion_amd.dist import sharded_sampling
        cond = torch.arange(n_total * 12, dtype=torch.float32).reshape(n_total, 3, 2, 2) * 1e-3
        y = torch.arange(n_total) * 10
        full = sharded_sampling(_PhiloxModel(), n_total, seed=9, cond=cond, y=y, device="cpu")
        q.put((rank, full.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(3, 7), (8, 16), (8, 11)])
def test_gloo_sharded_sampling_plumbing(world, n_total):
    """sharded_sampling over 3 (ragged) and 8 ranks: each rank's slice of the global cond / y, its sample offset into the noise
    stream and the gather reproduce the single-rank call"""
    q = mp.get_context("spawn").Queue()
    port = _free_port()
    res = _run_ranks(_sharded_worker, lambda r: (r, world, port, n_total, q), q, world=world)
    cond = torch.arange(n_total * 12, dtype=torch.float32).reshape(n_total, 3, 2, 2) * 1e-3
    ref = _PhiloxModel().sampling(n_total, cond=cond, y=torch.arange(n_total) * 10, rng="philox", seed=9).numpy()
    for r in range(world):
        assert np.array_equal(res[r], ref)


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from eo_diffusion_amd.training import allreduce_mean_
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)  # rank-dependent "gradients" in one flat bucket
        allreduce_mean_(flat)
        q.put((rank, flat.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_gloo_gradient_bucket_is_averaged(world):
    """data-parallel training (config 5): every rank ends with the MEAN of the ranks' flat gradient buckets"""
    q = mp.get_context("spawn").Queue()
    port = _free_port()
    res = _run_ranks(_grad_worker, lambda r: (r, world, port, q), q, world=world)
    ref = (torch.arange(1000, dtype=torch.float32) * float(sum(range(1, world + 1))) / world).numpy()
    for r in range(world):
        assert np.allclose(res[r], ref, rtol=1e-6, atol=0) and np.array_equal(res[r], res[0])


def test_allreduce_mean_is_a_noop_without_a_process_group():
    from eo_diffusion_amd.training import allreduce_mean_
    t = torch.ones(8)
    assert allreduce_mean_(t) is t and bool((t == 1).all())


def _run_bench(args, extra_env=None, timeout=240):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EOD_BENCH_STUB="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_gpus_n_starts_n_ranks():
    """`python bench.py --gpus 2` with no torchrun environment must itself start 2 ranks (here: the stub step over gloo, no GPU):
    rank 0 prints ONE line with n_gpus = 2 and the aggregate of both ranks' steps"""
    r, out = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1, r.stdout
    res = out[0]
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1 and res["stub"] is True
    assert res["config"]["global_batch"] == 2 * 16 and res["scaling"] == "weak"
    assert abs(res["value"] - 2 * 3 / (res["ms_per_step"] * 3e-3)) < 1e-6 * res["value"]  # whole-job rate = ranks x steps / max time


def test_bench_rejects_world_size_mismatch():
    """a launcher that started fewer ranks than --gpus says is an error, not a silent single-rank measurement"""
    r, out = _run_bench(["--gpus", "4", "--steps", "1", "--warmup", "0"], extra_env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not out
    assert "WORLD_SIZE=1" in (r.stderr + r.stdout)
