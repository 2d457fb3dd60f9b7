import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_training import _setup, _oracle_grads
from tests.helpers import rel_l2
from eo_diffusion_amd.training import UNetTrainer
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
size, base, mults, nrb, N = 16, 32, (1, 2), 1, 2
m, sd, cfg, x, noise, t = _setup(prec, size, base, mults, nrb, N)
pred_ref, gref = _oracle_grads(sd, cfg, x, noise, t)
tr = UNetTrainer(m, N, size, size, "cuda:0", loss_scale=(256.0 if prec == "fp16" else 1.0))
pred = tr.forward(x.cuda(), t.cuda())
print("pred err", rel_l2(pred.cpu(), pred_ref))
tr.backward(2.0 * (pred - noise.cuda()) / pred.numel())
torch.cuda.synchronize()
for name, p in m.named_parameters():
    if name in gref:
        print(f"{name:50s} {rel_l2(p.grad.cpu(), gref[name]):10.3e}  |ref|={float(gref[name].norm()):.3e}")
