"""Fused optimizer side of the training step on the HIP path (train.py:70-75,119-124): `AdamW` and
`ExponentialMovingAverage` with the constructor signatures the reference uses, running as ONE kernel launch over flat fp32
buffers (csrc/train.hip: eod_adamw_step, eod_ema_update) instead of torch's per-tensor loops, and `mse_loss`.

    from eo_diffusion_amd.optim import AdamW, ExponentialMovingAverage        # instead of torch.optim / utils.py
    optimizer = AdamW(model.parameters(), lr=args.lr)
    model_ema = ExponentialMovingAverage(model, device=device, decay=1.0 - alpha)

`AdamW` moves the parameters into one flat buffer (each `p.data` becomes a view of it, values unchanged), so checkpoints,
`state_dict()` and the weight re-packing of the kernels keep working on the same Parameter objects."""
import copy

import torch

from . import _lib
from ._lib import check, ptr
from .engine import current_stream_ptr


def _flatten(tensors, device):
    """one flat fp32 buffer + views (16-byte aligned) holding `tensors`' values"""
    offs, off = [], 0
    for t in tensors:
        offs.append(off)
        off += (t.numel() + 3) // 4 * 4
    flat = torch.zeros((off,), dtype=torch.float32, device=device)
    views = []
    for t, o in zip(tensors, offs):
        v = flat[o:o + t.numel()].view(t.shape)
        v.copy_(t.detach())
        views.append(v)
    return flat, views


def mse_loss(pred, target, want_grad=True):
    """nn.MSELoss(reduction='mean') (train.py:86,117) on the GPU: returns (loss [1] fp32 tensor, dLoss/dpred or None)"""
    L = _lib.lib()
    if not pred.is_cuda:
        raise _lib.EodError("mse_loss: tensors must be on the HIP GPU")
    p, t = pred.detach().contiguous().float(), target.detach().contiguous().float()
    loss = torch.empty((1,), dtype=torch.float32, device=p.device)
    dp = torch.empty_like(p) if want_grad else None
    scratch = torch.empty((1024,), dtype=torch.float32, device=p.device)
    check(L.eod_mse_loss(ptr(p), ptr(t), p.numel(), ptr(loss), ptr(dp), ptr(scratch), 1024, current_stream_ptr(p.device)), "eod_mse_loss")
    return loss, dp


class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, no amsgrad), one fused launch per parameter group."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, *, skip_nonfinite=True):
        """skip_nonfinite (default on): a step whose gradients contain inf / NaN (fp16 training overflowing its static loss scale)
        is skipped on the device -- parameters and moments stay untouched; `skipped_steps()` reads the count (one host sync)."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.skip_nonfinite = bool(skip_nonfinite)
        self._flat = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                self._flat.append(None)
                continue
            dev = ps[0].device
            if dev.type != "cuda" or any(p.dtype != torch.float32 for p in ps):
                raise _lib.EodError("AdamW: parameters must be fp32 tensors on the HIP GPU")
            flat_p, views = _flatten(ps, dev)
            for p, v in zip(ps, views):
                p.data = v  # same values, now contiguous in one buffer
            st = dict(params=ps, p=flat_p, g=torch.zeros_like(flat_p), m=torch.zeros_like(flat_p), v=torch.zeros_like(flat_p), step=0,
                      gviews=[None] * len(ps), guard=torch.zeros((4,), dtype=torch.int32, device=dev),
                      scratch=torch.zeros((8192,), dtype=torch.int32, device=dev))
            off = 0
            for k, p in enumerate(ps):
                st["gviews"][k] = st["g"][off:off + p.numel()].view(p.shape)
                off += (p.numel() + 3) // 4 * 4
            self._flat.append(st)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        L = _lib.lib()
        for group, st in zip(self.param_groups, self._flat):
            if st is None:
                continue
            untouched = []  # torch.optim.AdamW skips parameters without a gradient (no decay, no state update): keep their values
            # Fast path: the gradients already lie in ONE flat buffer with this optimizer's own layout (UNetTrainer.flat_grad: same
            # parameter order, same 16-byte-aligned offsets) -> the kernel reads it in place, no per-tensor copies.
            gptr, base, off = ptr(st["g"]), None, 0
            in_place = True
            for p in st["params"]:
                if p.grad is not None:
                    b = p.grad.data_ptr() - 4 * off
                    if base is None:
                        base = b
                    in_place = in_place and b == base and p.grad.dtype == torch.float32 and p.grad.is_contiguous()
                off += (p.numel() + 3) // 4 * 4
            if in_place and base is not None and base % 16 == 0:
                owner = next(p.grad for p in st["params"] if p.grad is not None)
                in_place = owner.untyped_storage().data_ptr() <= base and \
                    base + 4 * st["p"].numel() <= owner.untyped_storage().data_ptr() + owner.untyped_storage().nbytes()
            else:
                in_place = False
            def keep(p, o):  # value AND moments of a parameter without a gradient: torch.optim.AdamW leaves all three alone
                sl = slice(o, o + p.numel())
                untouched.append((p, p.detach().clone(), sl, st["m"][sl].clone(), st["v"][sl].clone()))
            off = 0
            if in_place:
                gptr = base
                for p in st["params"]:
                    if p.grad is None:  # (its slice of the flat buffer is never written by the backward: zeros)
                        keep(p, off)
                    off += (p.numel() + 3) // 4 * 4
            else:
                for p, gv in zip(st["params"], st["gviews"]):
                    if p.grad is None:
                        gv.zero_()
                        keep(p, off)
                    elif p.grad.data_ptr() != gv.data_ptr():
                        gv.copy_(p.grad)
                    off += (p.numel() + 3) // 4 * 4
            st["step"] += 1
            b1, b2 = group["betas"]
            if self.skip_nonfinite:
                check(L.eod_adamw_step_guarded(ptr(st["p"]), gptr, ptr(st["m"]), ptr(st["v"]), st["p"].numel(), float(group["lr"]),
                                               float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), st["step"],
                                               ptr(st["guard"]), ptr(st["scratch"]), 8192, current_stream_ptr(st["p"].device)),
                      "eod_adamw_step_guarded")
            else:
                check(L.eod_adamw_step(ptr(st["p"]), gptr, ptr(st["m"]), ptr(st["v"]), st["p"].numel(), float(group["lr"]), float(b1),
                                       float(b2), float(group["eps"]), float(group["weight_decay"]), st["step"], current_stream_ptr(st["p"].device)),
                      "eod_adamw_step")
            for p, val, sl, m0, v0 in untouched:  # undo the weight decay and the moment decay a zero gradient went through
                p.detach().copy_(val)
                st["m"][sl].copy_(m0)
                st["v"][sl].copy_(v0)
            for p in st["params"]:  # written through the flat buffer: advance torch's version counters (the packed-weight
                torch.autograd.graph.increment_version(p)  # caches of the kernels key on them); no kernel is launched
        return loss

    def skipped_steps(self):
        """number of steps skipped because of non-finite gradients (host synchronisation)"""
        return sum(int(st["guard"][1]) for st in self._flat if st is not None)


class ExponentialMovingAverage(torch.nn.Module):
    """script_utils/utils.py:56-67 (torchvision-style EMA on AveragedModel, use_buffers=True): `module` is a deep copy of the
    model whose float parameters AND buffers follow avg = decay*avg + (1-decay)*value; the first update copies."""

    def __init__(self, model, decay, device="cpu"):
        super().__init__()
        self.module = copy.deepcopy(model)
        if device is not None:
            self.module = self.module.to(device)
        self.decay = float(decay)
        self.register_buffer("n_averaged", torch.tensor(0, dtype=torch.long, device=device))
        self._flat = None

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def _pairs(self, model):
        a = list(self.module.parameters()) + list(self.module.buffers())
        b = list(model.parameters()) + list(model.buffers())
        return [(x, y) for x, y in zip(a, b)]

    @torch.no_grad()
    def update_parameters(self, model):
        L = _lib.lib()
        pairs = self._pairs(model)
        if int(self.n_averaged) == 0:
            for a, b in pairs:
                a.copy_(b.detach().to(a.device))
        else:
            npar = len(list(self.module.parameters()))
            par = [(a, b) for a, b in pairs[:npar] if a.dtype == torch.float32 and a.is_cuda]
            rest = [(a, b) for a, b in pairs if not any(a is x for x, _ in par)]
            st = current_stream_ptr(par[0][0].device)
            if self._flat is None:  # the EMA parameters live in one flat buffer with the same layout as optim.AdamW's
                flat, views = _flatten([a for a, _ in par], par[0][0].device)
                for (a, _), v in zip(par, views):
                    a.data = v
                self._flat = flat
            src = [b for _, b in par]
            contiguous = all(src[k].data_ptr() + ((src[k].numel() + 3) // 4 * 4) * 4 == src[k + 1].data_ptr() for k in range(len(src) - 1))
            if contiguous:  # one launch over all parameters
                check(L.eod_ema_update(ptr(self._flat), src[0].data_ptr(), self._flat.numel(), self.decay, st), "eod_ema_update")
            else:
                for a, b in par:
                    check(L.eod_ema_update(ptr(a), ptr(b.detach().contiguous()), a.numel(), self.decay, st), "eod_ema_update")
            for a, b in rest:  # buffers (use_buffers=True): the schedule tables etc.
                if a.dtype == torch.float32 and a.is_cuda and b.is_cuda:
                    check(L.eod_ema_update(ptr(a), ptr(b.detach().contiguous()), a.numel(), self.decay, st), "eod_ema_update")
                else:
                    a.copy_(b.detach().to(a.device))
            # the kernels wrote through raw pointers: advance torch's version counters so that the EMA copy's cached launch
            # program (packed conv / qkv / proj weights, keyed on (data_ptr, _version)) is rebuilt before its next forward
            for a, _ in pairs:
                if a.is_cuda and a.dtype == torch.float32:
                    torch.autograd.graph.increment_version(a)
        self.n_averaged += 1
