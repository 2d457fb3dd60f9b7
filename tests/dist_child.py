"""Child process of tests/test_gpu_dist.py: a ONE-rank RCCL ("nccl") process group on the GPU, in a process that has not touched the
GPU before.  Runs dist.sharded_sampling with the all-gather forced (all_gather_into_tensor on HBM tensors), compares it bit for
bit with the plain EODiffusion.sampling of the same seeds, checks the ragged gather path, and prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    from eo_diffusion_amd.diffusion.model import EODiffusion
    from eo_diffusion_amd.dist import gather_samples, sharded_sampling
    from tests.synth import synth_state_dict

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1)
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    try:
        cfg = dict(image_size=16, in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=[2],
                   channel_mult=[1, 2], num_heads=4)
        u = UNetModel(**cfg).set_precision(os.environ.get("EOD_PRECISION", "fp32x3"))
        u.load_state_dict(synth_state_dict(unet_param_shapes(**cfg), 7))
        m = EODiffusion(u, timesteps=6, image_size=16, in_channels=3, device=str(dev)).to(dev).eval()
        full = sharded_sampling(m, 4, seed=7, device=str(dev), force_gather=True)
        ref = m.sampling(4, device=str(dev), rng="philox", seed=7, progress=False)
        torch.cuda.synchronize()
        res["sharded_equals_plain_bits"] = bool(torch.equal(full, ref))
        res["finite"] = bool(torch.isfinite(full).all())
        res["on_gpu"] = full.is_cuda
        # the collective on a bare HBM tensor too
        x = torch.arange(3 * 5, dtype=torch.float32, device=dev).reshape(3, 5)
        g = gather_samples(x, 3, force_gather=True)
        res["gather_is_copy"] = bool(torch.equal(g, x)) and g.data_ptr() != x.data_ptr()
    finally:
        dist.destroy_process_group()
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
