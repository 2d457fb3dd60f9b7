#!/usr/bin/env python3
"""The reference's training loop (train.py:45-150) on synthetic images, running on the HIP path end to end.

Same objects and calls as the reference script -- UNetModel(...), EODiffusion(...), model(image, noise), nn.MSELoss,
loss.backward(), AdamW.step(), ExponentialMovingAverage.update_parameters(), {"model", "model_ema"} checkpoints, sampling from
the EMA copy -- only the dataset is replaced by random tensors (no network access here).  `--fused` swaps torch's AdamW / the
AveragedModel-based EMA for the one-launch versions in eo_diffusion_amd.optim (the only optional change to the script).

    python examples/train_synthetic.py --steps 20 --image-size 64 --batch-size 16 [--fp16] [--fused] [--ckpt out.pt]
"""
import argparse
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from eo_diffusion_amd.backbones.unet_openai import UNetModel  # noqa: E402  (= `from backbones.unet_openai import *` via dropin/)
from eo_diffusion_amd.diffusion.model import EODiffusion  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--image-size", type=int, default=64)
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--model-ema-decay", type=float, default=0.995)
    ap.add_argument("--fp16", action="store_true", help="fp16 storage / MFMA with fp32 accumulation (default: exact-fp32 mode)")
    ap.add_argument("--fused", action="store_true", help="eo_diffusion_amd.optim.AdamW / ExponentialMovingAverage (one launch each)")
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--n-samples", type=int, default=4)
    args = ap.parse_args()
    device = "cuda:0"
    torch.manual_seed(0)
    # train.py:50-61
    unet = UNetModel(args.image_size, in_channels=3, model_channels=128, out_channels=3, channel_mult=[1, 2, 3, 4],
                     attention_resolutions=[], num_res_blocks=1, num_heads=1, use_fp16=args.fp16)
    model = EODiffusion(unet, timesteps=args.timesteps, image_size=args.image_size, in_channels=3).to(device)
    print(f"Diffusion with {sum(p.numel() for p in model.parameters() if p.requires_grad) / 1e6:.2f} M params")
    if args.fused:
        from eo_diffusion_amd.optim import AdamW, ExponentialMovingAverage
        optimizer = AdamW(model.parameters(), lr=args.lr)
        model_ema = ExponentialMovingAverage(model, device=device, decay=args.model_ema_decay)
    else:
        from torch.optim import AdamW
        from torch.optim.swa_utils import AveragedModel
        optimizer = AdamW(model.parameters(), lr=args.lr)
        d = args.model_ema_decay
        model_ema = AveragedModel(model, device, lambda avg, p, n: d * avg + (1 - d) * p, use_buffers=True)  # utils.py:56-67
    # train.py:76-85: cos warm-up from lr/100 over the first tenth of the run, then lr * exp(-3 * progress of the rest)
    from eo_diffusion_amd.train_utils import KeyframeLR
    max_steps, posmax = args.steps, max(1, args.steps // 10)
    scheduler = KeyframeLR(optimizer=optimizer, units="steps", end=max_steps, frames=[
        {"position": 0, "lr": args.lr / 100}, {"transition": "cos"}, {"position": posmax, "lr": args.lr},
        {"transition": lambda last_lr, sf, ef, pos, *_: args.lr * math.exp(-3 * (pos - posmax) / max(1, max_steps - posmax))}])
    loss_fn = nn.MSELoss(reduction="mean")
    g = torch.Generator(device=device).manual_seed(1)
    model.train()
    t0 = time.perf_counter()
    for step in range(args.steps):
        image = torch.rand((args.batch_size, 3, args.image_size, args.image_size), device=device, generator=g)  # data in [0, 1]
        noise = torch.randn_like(image)
        pred = model(image, noise)           # train.py:116
        loss = loss_fn(pred, noise)          # :117
        loss.backward()                      # :118
        optimizer.step()                     # :119
        optimizer.zero_grad()                # :120
        scheduler.step()                     # :121
        model_ema.update_parameters(model)   # :123
        if step % 5 == 0 or step == args.steps - 1:
            torch.cuda.synchronize()
            print(f"Step[{step + 1}/{args.steps}], loss:{loss.detach().item():.5f}, {(time.perf_counter() - t0) / (step + 1) * 1e3:.1f} ms/step")
    if args.ckpt:
        torch.save({"model": model.state_dict(), "model_ema": model_ema.state_dict()}, args.ckpt)  # train.py:137-138
        print("saved", args.ckpt)
    model_ema.eval()
    samples = model_ema.module.sampling(args.n_samples, clipped_reverse_diffusion=True, device=device)  # train.py:147
    print("sampled", tuple(samples.shape), "finite" if bool(torch.isfinite(samples).all()) else "NON-FINITE",
          f"grid {int(math.sqrt(args.n_samples))}x{int(math.sqrt(args.n_samples))}")


if __name__ == "__main__":
    main()
