"""Batch-sharded sampling across the GPUs of one node.

The reference has no distributed code at all (single hard-coded device, train.py:46, inference.py:55).
Each sample's reverse chain is independent (GroupNorm is per-sample, no BatchNorm on the path), so the
path shards by samples with NO per-step communication: rank r of W runs samples [r*B/W, (r+1)*B/W),
draws its noise from the counter-based Philox generator keyed by the GLOBAL sample index (results are
therefore identical for every W, including W = 1), and the final images are concatenated with ONE
all-gather (RCCL over xGMI with backend "nccl"; gloo on CPU tensors in the tests).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_total, world, rank):
    """[lo, hi) of rank's samples; the first n_total % world ranks take one extra sample."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_samples(local, n_total, group=None, force_gather=False):
    """All-gather ragged per-rank shards [n_r, ...] into [n_total, ...] on every rank (one collective).  A single rank returns its
    shard as it is -- unless force_gather: then the collective runs even in a one-rank group (the RCCL all_gather_into_tensor on
    device memory, executed for real on a one-GPU box; tests/test_gpu_dist.py)."""
    initialised = dist.is_available() and dist.is_initialized()
    if force_gather and not initialised:
        raise RuntimeError("gather_samples(force_gather=True) needs an initialised process group")
    if not initialised or (dist.get_world_size(group) == 1 and not force_gather):
        assert local.shape[0] == n_total
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world)]
    assert local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    nmax = max(sizes)
    tail = local.shape[1:]
    if all(s == nmax for s in sizes):
        out = torch.empty((n_total,) + tuple(tail), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # ragged: pad to the largest shard, gather, strip
    pad = torch.zeros((nmax,) + tuple(tail), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty((world * nmax,) + tuple(tail), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * nmax: r * nmax + sizes[r]] for r in range(world)], 0)


@torch.no_grad()
def sharded_sampling(model, n_samples, *, seed=0, clipped_reverse_diffusion=True, cond=None, y=None, device=None,
                     group=None, progress=False, force_gather=False):
    """EODiffusion.sampling over all ranks of `group`: every rank returns the full [n_samples, C, H, W] tensor.
    cond / y are the GLOBAL tensors (each rank slices its own rows).  force_gather: see gather_samples."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(n_samples, world, rank)
    dev = device or f"cuda:{torch.cuda.current_device()}"
    c = cond[lo:hi] if cond is not None else None
    yy = y[lo:hi] if y is not None else None
    local = model.sampling(hi - lo, clipped_reverse_diffusion=clipped_reverse_diffusion, device=dev, cond=c, y=yy,
                           rng="philox", seed=seed, sample_offset=lo, progress=progress)
    return gather_samples(local, n_samples, group=group, force_gather=force_gather)
