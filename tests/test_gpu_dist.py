"""GPU: the multi-GPU code path executed for real on ONE GPU -- a one-rank RCCL group ("nccl" backend) whose all_gather_into_tensor
runs on HBM tensors -- in FRESH child processes (a process that has initialised the GPU is never re-executed; the children are
started with subprocess and their one JSON line is read back).  SURVEY.md section 8e: batch shards, Philox keyed by the global
sample index, one all-gather of the final images."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _child_env(**extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.update(extra)
    return env


def _json_lines(out):
    return [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]


def test_bench_runs_its_rccl_all_gather_on_one_gpu():
    """bench.py with EOD_BENCH_FORCE_DIST=1: init_process_group("nccl"), the barriers, the all_gather_into_tensor of the final images
    (device memory) and the MAX all-reduce of the timing all execute; the JSON line reports the collective and that the gathered
    tensor holds this rank's shard bit for bit"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-secondary", "--size", "64", "--batch", "4"], env=_child_env(EOD_BENCH_FORCE_DIST="1"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_lines(r.stdout)
    assert len(out) == 1, r.stdout
    res = out[0]
    assert res["n_gpus"] == 1 and res["steps"] == 2 and res["outputs_finite"] is True and res["value"] > 0
    c = res["collective"]
    assert c["op"] == "all_gather_into_tensor" and c["backend"] == "nccl" and c["device"].startswith("cuda") and c["own_shard_bit_equal"] is True
    assert c["bytes_per_rank"] == 4 * 3 * 64 * 64 * 4


def test_sharded_sampling_through_a_one_rank_rccl_group():
    """dist.sharded_sampling(force_gather=True) under a one-rank nccl group == EODiffusion.sampling with the same Philox seeds"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_child.py")], env=_child_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_lines(r.stdout)
    assert len(out) == 1, r.stdout
    res = out[0]
    assert res == {"backend": "nccl", "world": 1, "sharded_equals_plain_bits": True, "finite": True, "on_gpu": True, "gather_is_copy": True}
