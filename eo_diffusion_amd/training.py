"""Training step of the UNet on the HIP path (reference train.py:109-124:
`pred = model(image, noise)`, `loss = MSELoss(pred, noise)`, `loss.backward()`).

`UNetTrainer` builds, per input shape, (1) a forward Program that keeps every activation the backward needs and
(2) a static list of backward launches.  The backward reuses the forward's MFMA kernels (see csrc/train.hip):
backward-data = implicit-GEMM conv with flipped weights, backward-weights = batched NT GEMM over the pixel axis on
transposed operands, GroupNorm/SiLU/timestep-MLP backward = small HBM-bound kernels.  Parameter gradients are fp32 and are
assigned to `param.grad`, so `torch.optim.AdamW(model.parameters())` / `AveragedModel` of the reference script work
unchanged on them.

Scope of this first version: ResBlock (plain), AttentionBlock (legacy qkv order), Upsample / Downsample with conv,
first / head conv, the timestep MLP, class conditioning, and the factory variants use_scale_shift_norm (FiLM),
resblock_updown (incl. odd maps and the 3x3 -> 7x7 pad hack), use_new_attention_order, dropout (Philox mask, recomputed in the
backward), stride-2 convs of odd maps, Upsample(use_conv=True) of a 3x3 map, Up/Downsample without conv (conv_resample=False)."""
import ctypes as C
import math
import os

import torch
import torch.nn as nn

from ._lib import OP_DROPOUT, OP_TRANSPOSE, PACK_CHUNK, EodError, PackJob, check, ptr
from .engine import Act, Program, current_stream_ptr, round_up


def allreduce_mean_(flat, group=None):
    """in-place mean of one flat gradient bucket over the ranks of `group` (RCCL on GPU tensors, gloo on CPU tensors);
    no-op without an initialised process group or with a single rank"""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return flat
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    dist.all_reduce(flat, group=group)
    flat.div_(world)
    return flat


class _ConvRec:
    def __init__(self, srcs, conv, y, *, ksize=3, stride=1, upsample=False, res=None, emb=None, src_needs_grad=True, cout_rows=None):
        self.srcs, self.conv, self.y = srcs, conv, y
        self.ksize, self.stride, self.upsample, self.res, self.emb = ksize, stride, upsample, res, emb
        self.src_needs_grad = src_needs_grad
        self.cout_rows = cout_rows  # real output channels when y is channel-padded (head conv)


class _GNRec:
    def __init__(self, srcs, gn, ss, parts, y, silu, film=None):
        self.srcs, self.gn, self.ss, self.parts, self.y, self.silu = srcs, gn, ss, parts, y, silu
        self.film = film  # (view of the [N][J] emb_layers output at this block's columns, J, column offset) or None


class _DropRec:
    """nn.Dropout of ResBlock.out_layers (:339): y = x * mask / (1 - p), mask = Philox(seed; layer, step) -- recomputed, never stored"""

    def __init__(self, src, y, layer, p):
        self.src, self.y, self.layer, self.p = src, y, layer, p


class _PoolRec:
    """2x resampling without conv (resblock_updown, unet_openai.py:320-325): mode 0 = average pool (floor), 1 = nearest (with the
    3x3 -> 7x7 zero row / column of :237-239 when pad_tl)"""

    def __init__(self, src, y, mode, pad_tl=False):
        self.src, self.y, self.mode, self.pad_tl = src, y, mode, pad_tl


class _AttnRec:
    """softmax(q k^T / sqrt(d)) v on the natural [N][T][heads x (q|k|v) x d] layout of qkv (QKVAttentionLegacy, :465-481)"""

    def __init__(self, qkv, qkvT, ldT, P, a, nh, d, lay, lse=None):
        self.qkv, self.qkvT, self.ldT, self.P, self.a, self.nh, self.d = qkv, qkvT, ldT, P, a, nh, d
        self.lse = lse  # flash forward: P is NOT kept; the backward rebuilds it from the scores and this log-sum-exp
        self.lay = lay  # (q offset, k offset, v offset, head stride) in channels of the 3C-wide qkv tensor


class _VirtConv:
    """A 1x1 conv whose weight / bias are re-indexed copies of a real conv's: `weight` [Co'][Ci'][1][1] and `bias` [Co'] are fp32 tensors
    owned here, rows / columns of the real parameters scattered to `rows` / `cols` (everything else zero), rows scaled by `rscale`.
    Re-formed from the live parameters before every forward (UNetTrainer.virt); its gradients land in buffers of the same shape and are
    gathered back into the real parameters' gradients right after the conv's backward launches (`gather`).  Used by attention blocks
    whose head rows are not whole 16-byte chunks: every head is zero-padded to one (UNetTrainer._attention)."""

    def __init__(self, real, cout, cin, rows=None, cols=None, rscale=None, own_bias=True):
        dev = real.weight.device
        self.real, self.out_channels = real, cout
        self.rows, self.cols, self.rscale = rows, cols, rscale
        self.weight = torch.zeros((cout, cin, 1, 1), dtype=torch.float32, device=dev)
        self.bias = torch.zeros((cout,), dtype=torch.float32, device=dev) if own_bias else real.bias
        self.gw = torch.zeros_like(self.weight)
        self.gb = torch.zeros_like(self.bias) if own_bias else None

    def refresh(self):
        w = self.real.weight.detach().reshape(self.real.weight.shape[0], -1).float()
        b = self.real.bias.detach().float()
        if self.rscale is not None:
            w, b = w * self.rscale[:, None], b * self.rscale
        wp = self.weight.view(self.weight.shape[0], self.weight.shape[1])
        if self.rows is not None:
            wp.index_copy_(0, self.rows, w)
            self.bias.index_copy_(0, self.rows, b)
        else:
            wp.index_copy_(1, self.cols, w)

    def gather(self, dW, db):
        """padded gradients -> the real parameters' fp32 gradient views"""
        g = self.gw.view(self.gw.shape[0], self.gw.shape[1])
        if self.rows is not None:
            gw, gb = g.index_select(0, self.rows), self.gb.index_select(0, self.rows)
            if self.rscale is not None:
                gw, gb = gw * self.rscale[:, None], gb * self.rscale
            dW.view(gw.shape).copy_(gw)
            db.copy_(gb)
        else:
            dW.view(dW.shape[0], -1).copy_(g.index_select(1, self.cols))


class UNetTrainer:
    """Forward + backward of one UNetModel for a fixed input shape.  Usage:
        tr = UNetTrainer(unet, N, H, W, device, loss_scale=1024.)
        pred = tr.forward(x, t)                  # NCHW fp32, identical math to the inference program (unfused GroupNorm)
        tr.backward(dpred)                       # dpred NCHW fp32 = dLoss/dpred; fills p.grad for every parameter
    """

    def __init__(self, unet, N, H, W, device, *, cond_channels=0, loss_scale=1.0, dropout_seed=None):
        from .backbones import unet_openai as U
        self.U = U
        self.unet = unet
        self.device = torch.device(device)
        self.N, self.H, self.W = N, H, W
        self.loss_scale = float(loss_scale)
        self.inv_scale = 1.0 / self.loss_scale
        # (the split-fp16 inference mode "fp32x3" trains as exact "fp32": the backward kernels keep plain fp32 operands)
        prec = "fp32" if unet.precision == "fp32x3" else unet.precision
        self.prog = Program(self.device, prec)   # forward ops (native executor)
        self.bprog = Program(self.device, prec)  # backward ops that reuse executor op kinds
        self.L = self.prog.L
        self.dt = self.prog.dt
        self.es = self.prog.tdtype.itemsize
        self.bwd = []        # backward launches in execution order: ("op", Op) | ("call", fn, args)
        self.contrib = {}    # id(Act) -> [Act, ...] gradient contributions
        self.pgrad = {}      # Parameter -> fp32 gradient tensor
        self.repack = []     # closures refreshing derived tensors from the (updated) parameters (timestep-MLP concatenations)
        self.pack_jobs = []  # weight re-packs (forward and backward-data layouts of every conv): ONE launch per forward
        self.up4_sums = []   # (conv, fp32 class-kernel tensor) of the parity-class upsample convs: re-formed before the re-packs
        self.virt = []       # _VirtConv objects: re-formed from the live parameters before the re-packs
        self.vgrad = {}      # their weight / bias tensors -> fp32 gradient buffers
        self.recs = []
        self._keep = []
        self._scratch, self._scratch_all = {}, []
        self.dropout_seed = int(torch.initial_seed() if dropout_seed is None else dropout_seed) & (2**63 - 1)
        self._drop_ops = []  # forward executor ops whose `step` field is refreshed every forward
        self._alloc_flat_grad()
        # descriptors below bake raw device pointers of the LIVE parameters (biases, GroupNorm affine, timestep MLP): remember
        # where every parameter lives so that a later re-pointing (optim.AdamW's flat buffer, .to(), load_state_dict(assign=True))
        # is noticed instead of silently training on the old storage
        self._param_ptrs = self._ptr_fingerprint()
        self._build(cond_channels)

    # ------------------------------------------------------------------ small helpers
    def _ptr_fingerprint(self):
        return tuple(p.data_ptr() for p in self.unet.parameters())

    def stale(self):
        """True when a parameter's storage moved after this trainer was built (its descriptors point at the old storage)"""
        return self._ptr_fingerprint() != self._param_ptrs

    def _call(self, fn, *args):
        self.bwd.append(("call", fn, args))

    def _bop(self, build):
        n0 = len(self.bprog.ops)
        r = build()
        for op in self.bprog.ops[n0:]:
            self.bwd.append(("op", op))
        return r

    def _param_grad(self, p):
        """fp32 gradient of `p`: a view into ONE flat buffer (`flat_grad`), so that data-parallel training reduces all
        gradients with a single RCCL all-reduce (allreduce_grads) and a fused optimizer can walk them in one launch"""
        if p in self.vgrad:  # weight / bias of a _VirtConv: a buffer of its own, gathered into the real gradient afterwards
            return self.vgrad[p]
        if p not in self.pgrad:
            off = self._grad_offsets[p]
            self.pgrad[p] = self.flat_grad[off:off + p.numel()].view(p.shape)
        return self.pgrad[p]

    def _alloc_flat_grad(self):
        self._grad_offsets, off = {}, 0
        for p in self.unet.parameters():
            self._grad_offsets[p] = off
            off += (p.numel() + 3) // 4 * 4  # 16-byte aligned views
        self.flat_grad = torch.zeros((off,), dtype=torch.float32, device=self.device)

    def _plan_buckets(self, n_buckets=4):
        """gradient buckets for the OVERLAPPED data-parallel reduction: the flat buffer is cut into n_buckets contiguous ranges;
        a bucket is complete once the last backward launch that writes into it has been enqueued (found by scanning the launch
        list for pointers into the range) -- its all-reduce is issued right there and runs on RCCL's stream under the rest of the
        backward.  Parameters are laid out in forward order and the backward finishes them last-to-first, so the buckets
        complete from the back of the buffer to the front."""
        base, total = self.flat_grad.data_ptr(), self.flat_grad.numel()
        starts = sorted(self._grad_offsets.values())  # bucket boundaries sit ON parameter boundaries: no tensor straddles two buckets
        cuts = [0]
        te = [self._grad_offsets[p_] for p_ in self.unet.time_embed.parameters()]
        after_te = min(o for o in starts if o > max(te)) if te else 0
        if after_te:  # the timestep MLP's gradients are the last ones to be final: keep them in their own tiny bucket
            cuts.append(after_te)
        for k in range(1, n_buckets):
            c = min(starts, key=lambda o: abs(o - k * total // n_buckets))
            if c > cuts[-1]:
                cuts.append(c)
        cuts.append(total)
        bounds = list(zip(cuts[:-1], cuts[1:]))
        ready = [-1] * len(bounds)
        for i, item in enumerate(self.bwd):
            if item[0] != "call" and not (item[0] == "dyn" and len(item) > 2):  # (a "dyn" item may name the pointers it writes)
                continue
            for a in item[2]:
                if isinstance(a, int) and base <= a < base + total * 4:
                    off = (a - base) // 4
                    for k, (lo, hi) in enumerate(bounds):
                        if lo <= off < hi:
                            ready[k] = max(ready[k], i)
        self._buckets = sorted(zip(ready, bounds), key=lambda rb: rb[0])

    def allreduce_grads(self, group=None):
        """data-parallel step (config 5: one rank per GPU): average the gradients of all ranks, one bucket = everything
        (55-88 M fp32 values, a few ms over xGMI)"""
        allreduce_mean_(self.flat_grad, group)

    def _add_grad(self, act, g):
        self.contrib.setdefault(id(act), []).append(g)
        self._keep.append(act)

    def _pop_single(self, act):
        lst = self.contrib.get(id(act), [])
        if len(lst) == 1:
            return lst.pop()
        return None

    def _take_grad(self, act):
        lst = self.contrib.pop(id(act), [])
        if not lst:
            return None
        g = lst[0]
        for other in lst[1:]:
            out = self.bprog.act(g.N, g.H, g.W, g.C)
            self._call(self.L.eod_add, ptr(g.t), ptr(other.t), ptr(out.t), self.dt, g.t.numel())
            g = out
        return g

    def _shared(self, key, numel, dtype=None, zero=False):
        """scratch tensor shared by all launches that ask for `key`: everything runs in order on one stream, so
        temporaries whose lifetime ends inside one block (attention scores, their gradients, ...) can alias.  A request that is
        larger than what was handed out before gets a new buffer (pointers already captured stay valid)."""
        dtype = dtype or self.prog.tdtype
        cur = self._scratch.get((key, dtype))
        if cur is None or cur.numel() < numel:
            cur = (torch.zeros if zero else torch.empty)((numel,), dtype=dtype, device=self.device)
            self._scratch[(key, dtype)] = cur
            self._scratch_all.append(cur)
        return cur[:numel]

    # ------------------------------------------------------------------ forward emission (training form)
    def _conv_fwd(self, srcs, conv, *, ksize=3, stride=1, upsample=False, res=None, emb=None, stats=True, src_needs_grad=True, skip=None):
        """skip = (block inputs, the 1x1 skip_connection conv): its product rides in this conv's accumulators (eod_conv_desc.skip_x); the
        backward is unchanged -- the skip conv gets a record of its own whose output is a placeholder that only ever carries dY"""
        prog = self.prog
        fwd_ups = upsample
        if (upsample and len(srcs) == 1 and res is None and emb is None and ksize == 3 and stride == 1
                and prog.conv_up4_ok(srcs[0], conv.out_channels)):
            # forward of a conv over a nearest-2x upsampling in its parity-class form (4/9 of the MACs, engine.pack_conv_up4): the class
            # kernels are re-formed from the live weight every step (eod_conv_up4_weights), then packed like any weight; the backward
            # keeps the nine-tap formulation on the original weight (rec.upsample)
            cout, cin = conv.weight.shape[0], conv.weight.shape[1]
            wc = prog.empty((4 * cout, cin, 3, 3), torch.float32)
            self.up4_sums.append((conv, wc))
            w = prog.empty((9, 4 * cout, cin))
            self._add_pack_job(0, conv, w, src=wc)
            fwd_ups = wc
        else:
            w = prog.pack_conv(conv.weight)
            self._add_pack_job(0, conv, w)
        kw = {}
        if emb is not None:
            kw = dict(cbias=emb[0], cbias_stride=emb[1])
        up4_wc = fwd_ups if isinstance(fwd_ups, torch.Tensor) else None
        if skip is not None:
            ssrcs, sconv = skip
            assert res is None and len(srcs) == 1
            sw = prog.empty((1, conv.out_channels, sum(a.C for a in ssrcs)))
            self._add_pack_job(0, sconv, sw)
            res = Act(None, srcs[0].N, srcs[0].H, srcs[0].W, conv.out_channels)  # placeholder of skip_connection(x): never materialised
            self.recs.append(_ConvRec(ssrcs, sconv, res, ksize=1))
            kw["skip"] = (ssrcs, ("packed", sw), sconv.bias)
        y, _ = prog.conv(srcs[0], w, prog.f32(conv.bias), conv.out_channels, x2=srcs[1] if len(srcs) > 1 else None,
                         ksize=ksize, stride=stride, pad=ksize // 2, upsample=("up4" if up4_wc is not None else fwd_ups),
                         res=None if skip is not None else res, stats=stats, **kw)
        rec = _ConvRec(srcs, conv, y, ksize=ksize, stride=stride, upsample=upsample, res=res, emb=emb, src_needs_grad=src_needs_grad)
        rec.up4_wc = up4_wc  # fp32 class-kernel tensor of the parity-class form (its backward-data can use it too)
        self.recs.append(rec)
        return y

    def _add_pack_job(self, kind, conv, dst, cpad=None, ci0=0, nci=0, src=None):
        """kind 0: forward packing [tap][Cout][cpad = cin_pad]; kind 1: backward-data packing of input channels [ci0, ci0 + nci)
        [taps-1-tap][ci][cpad = cout_pad] (eod_pack_jobs).  src: an fp32 OIHW tensor to pack instead of conv.weight"""
        w = conv.weight.detach() if src is None else src
        cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
        j = PackJob()
        j.w, j.dst, j.kind, j.Cout, j.Cin, j.taps = ptr(w), ptr(dst), kind, cout, cin, ks * ks
        j.ci0, j.nci, j.cpad = ci0, nci, (cpad or (cin if kind == 0 else cout))
        self.pack_jobs.append(j)

    def _finish_pack_jobs(self):
        n = len(self.pack_jobs)
        arr = (PackJob * n)(*self.pack_jobs)
        blk_job, blk_first = [], []
        for k, j in enumerate(self.pack_jobs):
            total = j.taps * (j.nci * j.cpad if j.kind else j.Cout * j.cpad)
            for first in range(0, total, PACK_CHUNK):
                blk_job.append(k)
                blk_first.append(first)
        self._pack_tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        self._pack_blk_job = torch.tensor(blk_job, dtype=torch.int32, device=self.device)
        self._pack_blk_first = torch.tensor(blk_first, dtype=torch.int64, device=self.device)
        self._pack_nblocks = len(blk_job)

    def _run_pack_jobs(self):
        check(self.L.eod_pack_jobs(ptr(self._pack_tab), ptr(self._pack_blk_job), ptr(self._pack_blk_first), self._pack_nblocks, self.dt,
                                   current_stream_ptr(self.device)), "eod_pack_jobs")

    def _gn_fwd(self, srcs, gn, silu=True, film=None):
        prog = self.prog
        kw = dict(film=film[0], film_stride=film[1]) if film is not None else {}
        ss = prog.gn_stats(srcs, prog.f32(gn.weight), prog.f32(gn.bias), eps=gn.eps, **kw)
        parts = prog.last_gn_parts
        y = prog.gn_apply(srcs, ss, silu=silu)
        self.recs.append(_GNRec(srcs, gn, ss, parts, y, silu, film))
        return y

    def _dropout_fwd(self, src, p):
        prog = self.prog
        layer = len(self._drop_ops)
        y = prog.act(src.N, src.H, src.W, src.C)
        idx = prog._small(OP_DROPOUT, p=(ptr(src.t), ptr(y.t)), l=(src.t.numel(), self.dropout_seed), i=(self.dt, layer, 0), f=(float(p),))
        self._drop_ops.append(idx)
        self.recs.append(_DropRec(src, y, layer, float(p)))
        return y

    def _dropout_bwd(self, rec):
        dy = self._take_grad(rec.y)
        if dy is None:
            raise EodError("training: a dropout output has no gradient (graph bug)")
        dx = self.bprog.act(dy.N, dy.H, dy.W, dy.C)
        # same key as the forward of THIS step (step_id is read when the backward runs)
        self.bwd.append(("dyn", lambda st, a=ptr(dy.t), b=ptr(dx.t), n=dy.t.numel(), r=rec: check(
            self.L.eod_dropout(a, b, self.dt, n, r.p, self.dropout_seed, r.layer, self.step_id, st), "eod_dropout")))
        self._add_grad(rec.src, dx)

    def dropout_mask(self, layer, like):
        """mask / (1 - p) of dropout layer `layer` for the most recent forward (test / inspection helper)"""
        rec = [r for r in self.recs if isinstance(r, _DropRec)][layer]
        ones = torch.ones_like(like)
        out = torch.empty_like(like)
        check(self.L.eod_dropout(ptr(ones), ptr(out), self.dt, ones.numel(), rec.p, self.dropout_seed, rec.layer, self.step_id,
                                 current_stream_ptr(self.device)), "eod_dropout")
        return out

    def _pool_fwd(self, src, mode):
        pad_tl = mode == 1 and src.H == 3 and src.W == 3  # Upsample's 3x3 -> 7x7 hack
        y = self.prog.resample2x(src, mode, pad_tl)
        self.recs.append(_PoolRec(src, y, mode, pad_tl))
        return y

    def _resblock(self, blk, h):
        U = self.U
        srcs = list(h) if isinstance(h, tuple) else [h]
        gn1, conv1 = blk.in_layers[0], blk.in_layers[2]
        gn2, conv2 = blk.out_layers[0], blk.out_layers[3]
        a1 = self._gn_fwd(srcs, gn1)
        if blk.updown:  # resblock_updown (:366-371): resample the normalised branch and the skip input, no conv
            if len(srcs) != 1 or blk.h_upd.use_conv:
                raise EodError("training: this resblock_updown variant is not built yet")
            mode = 1 if isinstance(blk.h_upd, U.Upsample) else 0
            a1 = self._pool_fwd(a1, mode)
            srcs = [self._pool_fwd(srcs[0], mode)]
        off = self.ctx.offsets[id(blk)]
        emb_view = self.ctx.out[:, off:]
        if blk.use_scale_shift_norm:  # FiLM (:377-381): emb_out = [scale | shift] modulates the second GroupNorm
            h1 = self._conv_fwd([a1], conv1)
            a2 = self._gn_fwd([h1], gn2, film=(emb_view, self.ctx.J, off, blk.emb_layers[1]))
        else:
            h1 = self._conv_fwd([a1], conv1, emb=(emb_view, self.ctx.J, off, blk.emb_layers[1]))
            a2 = self._gn_fwd([h1], gn2)
        if blk.dropout > 0 and self.unet.training:  # out_layers = [GroupNorm, SiLU, Dropout(p), conv]
            a2 = self._dropout_fwd(a2, blk.dropout)
        if isinstance(blk.skip_connection, nn.Identity):
            if len(srcs) != 1:
                raise EodError("identity skip over a virtual concat is not supported")
            skip = srcs[0]
        else:
            k = blk.skip_connection.kernel_size[0]
            if k == 1 and self.prog.conv_skip_ok(a2, conv2.out_channels, srcs):
                return self._conv_fwd([a2], conv2, skip=(srcs, blk.skip_connection))  # `skip_connection(x) + h` in conv2's launch
            skip = self._conv_fwd(srcs, blk.skip_connection, ksize=k, stats=False)
        return self._conv_fwd([a2], conv2, res=skip)

    def _layer(self, layer, h):
        U = self.U
        if isinstance(layer, U.ResBlock):
            return self._resblock(layer, h)
        if isinstance(layer, U.Upsample):
            if isinstance(h, tuple):
                raise EodError("training: Upsample over a virtual concat is not supported")
            if not layer.use_conv:           # conv_resample=False: plain nearest 2x (unet_openai.py:229-241 without :241)
                return self._pool_fwd(h, 1)
            if h.H == 3 and h.W == 3:        # the 3x3 -> 7x7 zero row / column hack (:237-239): materialise it, then a plain conv
                return self._conv_fwd([self._pool_fwd(h, 1)], layer.conv)
            return self._conv_fwd([h], layer.conv, upsample=True)
        if isinstance(layer, U.Downsample):
            if not layer.use_conv:           # conv_resample=False: 2x2 average pool (:266-267)
                return self._pool_fwd(h, 0)
            return self._conv_fwd([h], layer.op, stride=2)  # odd maps: the backward-data runs on the even grid and is cropped
        if isinstance(layer, U.AttentionBlock):
            return self._attention(layer, h)
        raise EodError(f"training: unsupported layer {type(layer).__name__}")

    def _attention(self, blk, x):
        """x + proj_out(attention(qkv(GN(x)))), unet_openai.py:427-433, in a form whose every piece has a backward here:
        qkv / proj_out as 1x1 convs, scores and P.V as batched NT GEMMs on qkv and its transpose, P kept for the backward.
        Sequence lengths that are not a multiple of one 16-byte chunk (T = 196 at 14x14) are pitched to Tp >= T: the pad
        columns of P / dS and of the transposed operands are zeros, so they drop out of every contraction."""
        prog = self.prog
        if isinstance(x, tuple):
            raise EodError("training: AttentionBlock over a virtual concat is not supported")
        C0, nh = blk.channels, blk.num_heads
        d0 = C0 // nh
        N, T = x.N, x.H * x.W
        qkv_conv, proj_conv, d = blk.qkv, blk.proj_out, d0
        if d0 % prog.epc:
            # head rows that are not whole 16-byte chunks (96 channels in 8 heads: d = 12): every head is zero-padded to d' = the next
            # multiple of one chunk.  qkv / proj_out run as _VirtConv copies whose extra rows / columns are zero -- the pad channels of
            # q, k, v, a and of every gradient are exact zeros, so they drop out of each contraction -- and the q rows carry
            # sqrt(d' / d): the kernels below scale the scores by 1 / sqrt(d'), which then is the reference's 1 / sqrt(d)
            # (unet_openai.py:475-478, 507-512).  The gradients of the padded copies are gathered back into the real parameters'.
            d = round_up(d0, prog.epc)
            new = blk.attention.new_order
            ridx = [(w * nh * d + h * d + j) if new else (h * 3 * d + w * d + j)
                    for r in range(3 * C0)
                    for (w, h, j) in [((r // C0, (r % C0) // d0, r % d0) if new else ((r % (3 * d0)) // d0, r // (3 * d0), r % d0))]]
            rsc = [math.sqrt(d / d0) if ((r // C0) if new else ((r % (3 * d0)) // d0)) == 0 else 1.0 for r in range(3 * C0)]
            cidx = [(c // d0) * d + c % d0 for c in range(C0)]
            dev = self.device
            qkv_conv = _VirtConv(blk.qkv, 3 * nh * d, C0, rows=torch.tensor(ridx, device=dev),
                                 rscale=torch.tensor(rsc, dtype=torch.float32, device=dev))
            proj_conv = _VirtConv(blk.proj_out, C0, nh * d, cols=torch.tensor(cidx, device=dev), own_bias=False)
            for vc in (qkv_conv, proj_conv):
                vc.refresh()
                self.virt.append(vc)
                self.vgrad[vc.weight] = vc.gw
                if vc.gb is not None:
                    self.vgrad[vc.bias] = vc.gb
        Cc = nh * d
        Tp = round_up(T, prog.epc)
        # channel layout of qkv: legacy [h][q|k|v][d] (unet_openai.py:474), new order [q|k|v][h][d] (:506-514)
        qo, ko, vo, hs = (0, Cc, 2 * Cc, d) if blk.attention.new_order else (0, d, 2 * d, 3 * d)
        xn = self._gn_fwd([x], blk.norm, silu=False)
        qkv = self._conv_fwd([xn], qkv_conv, ksize=1, stats=False)          # [N][T][3C]
        BK = 128 // self.es
        ldT = round_up(N * Tp, BK)
        qkvT = prog.empty((3 * Cc * ldT,), zero=True)                        # [3C][n*Tp + t]
        prog._small(OP_TRANSPOSE, p=(ptr(qkv.t), ptr(qkvT)), l=(ldT, 0, 0, 0), i=(self.dt, N, 1, T, 3 * Cc, 1, Tp, 1, 0, 0))
        flash = (prog.precision == "fp16" and d % 8 == 0 and d <= 64 and os.environ.get("EOD_ATTN_TRAIN", "flash") != "gemm")
        if flash:
            # forward = the inference path's fused kernel on the natural qkv layout (T x T never materialised); it also returns the
            # log-sum-exp of every score row, from which the backward rebuilds P
            lse = prog.empty((N, nh, T), torch.float32)
            a = prog.act(N, x.H, x.W, Cc)
            prog.attention_nat(qkv.t, a.t, N, T, Cc, nh, d, qo, ko, vo, hs, lse=lse)
            self.recs.append(_AttnRec(qkv, qkvT, ldT, None, a, nh, d, (qo, ko, vo, hs), lse))
            return self._conv_fwd([a], proj_conv, ksize=1, res=x, stats=True)
        S = self._shared("attn_S", N * nh * T * Tp, torch.float32)  # only P is kept for the backward
        prog.gemm(qkv.t, qkv.t, S, T, T, d, 3 * Cc, 3 * Cc, Tp, alpha=1.0 / math.sqrt(d), c_f32=True, nb0=N, nb1=nh,
                  sa=(T * 3 * Cc, hs), sb=(T * 3 * Cc, hs), sc=(nh * T * Tp, T * Tp), a_off=qo, b_off=ko)
        P = prog.empty((N * nh, T, Tp), zero=True)
        prog.softmax_rows(S, Tp, P, Tp, N * nh * T, T)
        a = prog.act(N, x.H, x.W, Cc)
        # a[n][t][h*d + j] = sum_s P[n,h][t][s] * v[n][s][h][j]  with v^T rows taken from qkvT
        prog.gemm(P, qkvT, a.t, T, d, Tp, Tp, ldT, Cc, nb0=N, nb1=nh, sa=(nh * T * Tp, T * Tp), sb=(Tp, hs * ldT),
                  sc=(T * Cc, d), b_off=vo * ldT)
        self.recs.append(_AttnRec(qkv, qkvT, ldT, P, a, nh, d, (qo, ko, vo, hs)))
        return self._conv_fwd([a], proj_conv, ksize=1, res=x, stats=True)

    def _attn_bwd(self, rec):
        L, bp, dt, es = self.L, self.bprog, self.dt, self.es
        da = self._take_grad(rec.a)
        if da is None:
            raise EodError("training: attention output has no gradient (graph bug)")
        qkv, qkvT, ldT, P, nh, d = rec.qkv, rec.qkvT, rec.ldT, rec.P, rec.nh, rec.d
        qo, ko, vo, hs = rec.lay
        N, T, Cc = qkv.N, qkv.H * qkv.W, rec.a.C
        Tp = round_up(T, self.prog.epc)
        B = N * nh
        alpha = 1.0 / math.sqrt(d)
        BK = 128 // es
        if P is None and os.environ.get("EOD_ATTN_BWD", "flash") == "flash":
            # flash-style backward: P is rebuilt tile by tile in registers from q, k and the forward's log-sum-exp
            D = self._shared("attn_D", B * T, torch.float32)
            self._call(L.eod_rowdot, ptr(da.t), ptr(rec.a.t), dt, N, nh, T, T * Cc, d, Cc, d, ptr(D))
            dqkv = bp.act(qkv.N, qkv.H, qkv.W, 3 * Cc)
            self._call(L.eod_attention_bwd, ptr(qkv.t), ptr(da.t), ptr(rec.lse), ptr(D), ptr(dqkv.t), dt, N, T, Cc, nh, d, qo, ko, vo, hs)
            self._add_grad(qkv, dqkv)
            return
        if P is None:  # flash forward: P = exp(q k^T / sqrt(d) - lse), one GEMM with the exp in its epilogue (bias_mode 4)
            P = self._shared("attn_P", B * T * Tp)
            self._bop(lambda: bp.gemm(qkv.t, qkv.t, P, T, T, d, 3 * Cc, 3 * Cc, Tp, alpha=alpha, bias=rec.lse, bias_mode=4, nb0=N, nb1=nh,
                                      sa=(T * 3 * Cc, hs), sb=(T * 3 * Cc, hs), sc=(nh * T * Tp, T * Tp), a_off=qo, b_off=ko))
        dS = self._shared("attn_dS", B * T * Tp)
        if Tp == T and os.environ.get("EOD_ATTN_BWD", "flash") != "nt":
            # dS = P * (dP - D) with dP = da v^T formed in the GEMM's accumulators only: D[n][h][t] = sum_j da*a (= rowsum(dP*P))
            # first, then the GEMM epilogue (bias_mode 3) subtracts D and multiplies by P -- the fp32 T x T dP never exists
            # dS ~ (1/T) |dP - D| is a SUBNORMAL fp16 number at T in the thousands: it is kept on a 2^12 scale (alpha of the GEMM, D
            # scaled to match), which the dq / dk products below divide out again (same device as csrc/attn_bwd.hip: AB_DS_SCALE)
            ds_scale = 4096.0 if self.prog.precision == "fp16" else 1.0
            D = self._shared("attn_D", B * T, torch.float32)
            self._call(L.eod_rowdot, ptr(da.t), ptr(rec.a.t), dt, N, nh, T, T * Cc, d, Cc, d, ptr(D))
            if ds_scale != 1.0:
                self._call(L.eod_scale_f32, ptr(D), B * T, ds_scale)
            self._bop(lambda: bp.gemm(da.t, qkv.t, dS, T, T, d, Cc, 3 * Cc, Tp, alpha=ds_scale, bias=D, bias_mode=3, res=P, nb0=N, nb1=nh,
                                      sa=(T * Cc, d), sb=(T * 3 * Cc, hs), sc=(nh * T * Tp, T * Tp), b_off=vo))
            alpha = alpha / ds_scale
        else:
            # dP[b][t][s] = sum_j da[n][t][h*d+j] * v[n][s][h][j], then the row-wise softmax backward
            dP = self._shared("attn_dP", B * T * Tp, torch.float32)
            self._bop(lambda: bp.gemm(da.t, qkv.t, dP, T, T, d, Cc, 3 * Cc, Tp, c_f32=True, nb0=N, nb1=nh, sa=(T * Cc, d),
                                      sb=(T * 3 * Cc, hs), sc=(nh * T * Tp, T * Tp), b_off=vo))
            self._call(L.eod_softmax_bwd_rows, ptr(P), Tp, ptr(dP), Tp, ptr(dS), dt, B * T, T)
        dqkv = bp.act(qkv.N, qkv.H, qkv.W, 3 * Cc)
        # dq[n][t][h][j] = alpha * sum_s dS[b][t][s] * k[n][s][h][j]      (k^T rows from qkvT)
        self._bop(lambda: bp.gemm(dS, qkvT, dqkv.t, T, d, Tp, Tp, ldT, 3 * Cc, alpha=alpha, nb0=N, nb1=nh, sa=(nh * T * Tp, T * Tp),
                                  sb=(Tp, hs * ldT), sc=(T * 3 * Cc, hs), b_off=ko * ldT, c_off=qo))
        if self.prog.precision == "fp16" and T % 8 == 0 and os.environ.get("EOD_ATTN_BWD", "flash") != "nt":
            # dk = alpha * dS^T q and dv = P^T da: both operands are query-major as stored -> gemm_tn_kernel (transposed LDS
            # reads), the T x T matrices are not transposed in HBM
            self._call(L.eod_gemm_tn, ptr(dS), Tp, ptr(qkv.t) + qo * es, 3 * Cc, ptr(dqkv.t) + ko * es, 3 * Cc, dt, T, d, T, alpha,
                       N, nh, nh * T * Tp, T * Tp, T * 3 * Cc, hs, T * 3 * Cc, hs)
            self._call(L.eod_gemm_tn, ptr(P), Tp, ptr(da.t), Cc, ptr(dqkv.t) + vo * es, 3 * Cc, dt, T, d, T, 1.0,
                       N, nh, nh * T * Tp, T * Tp, T * Cc, d, T * 3 * Cc, hs)
        else:
            ldB = round_up(B * Tp, BK)
            dST = self._shared("attn_dST", Tp * ldB)   # [s][b*Tp + t]; the transposes write every column up to ldB
            PT = self._shared("attn_PT", Tp * ldB)
            self._call(L.eod_transpose_gather, ptr(dS), dt, B, 1, T, Tp, ptr(dST), ldB, 1, Tp, 1, 0, 0, 0, 0, 0)
            self._call(L.eod_transpose_gather, ptr(P), dt, B, 1, T, Tp, ptr(PT), ldB, 1, Tp, 1, 0, 0, 0, 0, 0)
            daT = self._shared("attn_daT", Cc * ldT)  # [h*d + j][n*Tp + t]
            self._call(L.eod_transpose_gather, ptr(da.t), dt, N, 1, T, Cc, ptr(daT), ldT, 1, Tp, 1, 0, 0, 0, 0, 0)
            # dk[n][s][h][j] = alpha * sum_t dS[b][t][s] * q[n][t][h][j]      (dS^T and q^T)
            self._bop(lambda: bp.gemm(dST, qkvT, dqkv.t, T, d, Tp, ldB, ldT, 3 * Cc, alpha=alpha, nb0=N, nb1=nh, sa=(nh * Tp, Tp),
                                      sb=(Tp, hs * ldT), sc=(T * 3 * Cc, hs), b_off=qo * ldT, c_off=ko))
            # dv[n][s][h][j] = sum_t P[b][t][s] * da[n][t][h*d+j]
            self._bop(lambda: bp.gemm(PT, daT, dqkv.t, T, d, Tp, ldB, ldT, 3 * Cc, nb0=N, nb1=nh, sa=(nh * Tp, Tp),
                                      sb=(Tp, d * ldT), sc=(T * 3 * Cc, hs), c_off=vo))
        self._add_grad(qkv, dqkv)

    def _seq(self, seq, h):
        for layer in seq:
            h = self._layer(layer, h)
        return h

    def _build(self, ccond):
        U, unet, prog, N, H, W = self.U, self.unet, self.prog, self.N, self.H, self.W
        self.with_y = unet.num_classes is not None
        cx = unet.in_channels - ccond
        c_pad = round_up(unet.in_channels, prog.epc)
        a0, self.i_in = prog.to_nhwc(N, cx, ccond, H, W, c_pad)
        # ---- timestep embedding (same single descriptor as the inference program) ----
        ctx = self.ctx = U._EmbCtx()
        for m in unet.modules():
            if isinstance(m, U.ResBlock):
                ctx.register(m)
        te1, te2 = unet.time_embed[0], unet.time_embed[2]
        self.E, self.D = te1.out_features, te1.in_features
        self.wcat = prog.empty((ctx.J, self.E), torch.float32)
        self.bcat = prog.empty((ctx.J,), torch.float32)
        self.repack.append(self._refresh_cat)
        self.freqs = prog.own(U.timestep_frequencies(self.D).to(self.device))
        ctx.out = prog.empty((N, ctx.J), torch.float32)
        self.h1 = prog.empty((N, self.E), torch.float32)
        self.emb = prog.empty((N, self.E), torch.float32)
        self.i_t = prog.temb(dict(
            t=0, freqs=ptr(self.freqs), w1=ptr(prog.f32(te1.weight)), b1=ptr(prog.f32(te1.bias)),
            w2=ptr(prog.f32(te2.weight)), b2=ptr(prog.f32(te2.bias)),
            label_emb=ptr(prog.f32(unet.label_emb.weight)) if self.with_y else 0, y=0,
            wcat=ptr(self.wcat), bcat=ptr(self.bcat), h1=ptr(self.h1), emb=ptr(self.emb), out=ptr(ctx.out),
            N=N, D=self.D, E=self.E, J=ctx.J))
        # ---- encoder / middle / decoder ----
        conv0 = unet.input_blocks[0][0]
        w0 = prog.pack_conv(conv0.weight, cin_pad=c_pad)
        self._add_pack_job(0, conv0, w0, cpad=c_pad)
        h, _ = prog.conv(a0, w0, prog.f32(conv0.bias), conv0.out_channels, stats=True)
        self.recs.append(_ConvRec([a0], conv0, h, src_needs_grad=False))
        hs = [h]
        for blk in list(unet.input_blocks)[1:]:
            h = self._seq(blk, h)
            hs.append(h)
        h = self._seq(unet.middle_block, h)
        for blk in unet.output_blocks:
            h = self._seq(blk, (h, hs.pop()))
        # ---- head: the NCHW fp32 output is produced from a channel-padded NHWC conv output ----
        gn, conv = unet.out[0], unet.out[2]
        a = self._gn_fwd([h], gn)
        self.cout = unet.out_channels
        self.cout_pad = round_up(self.cout, prog.epc)
        wh = prog.pack_conv(conv.weight)
        self._add_pack_job(0, conv, wh)
        self.pred = torch.empty((N, self.cout, H, W), dtype=torch.float32, device=self.device)
        _, i_out = prog.conv(a, wh, prog.f32(conv.bias), self.cout, out_nchw_f32=True)
        prog.ops[i_out].u.conv.y = self.pred.data_ptr()
        self.head = (a, conv)
        prog.finalize()
        self._build_backward()
        self._finish_pack_jobs()

    def _refresh_cat(self):
        torch.cat([b.emb_layers[1].weight.detach().float() for b in self.ctx.blocks], 0, out=self.wcat)
        torch.cat([b.emb_layers[1].bias.detach().float() for b in self.ctx.blocks], 0, out=self.bcat)

    # ------------------------------------------------------------------ backward emission
    def _build_backward(self):
        N, H, W = self.N, self.H, self.W
        bp = self.bprog
        # dLoss/dpred arrives NCHW fp32 -> NHWC storage dtype, channels padded to one 16-byte chunk, times loss_scale
        self.dpred = torch.zeros((N, self.cout, H, W), dtype=torch.float32, device=self.device)
        gpred, _ = self._bop(lambda: bp.to_nhwc(N, self.cout, 0, H, W, self.cout_pad))
        self.bwd[-1][1].u.small.p[0] = self.dpred.data_ptr()
        self.dout_cat = bp.empty((N, self.ctx.J), torch.float32)
        a_head, conv_head = self.head
        head_y = Act(None, N, H, W, self.cout_pad)
        self._add_grad(head_y, gpred)
        self._conv_bwd(_ConvRec([a_head], conv_head, head_y, cout_rows=self.cout))
        for rec in reversed(self.recs):
            if isinstance(rec, _ConvRec):
                self._conv_bwd(rec)
            elif isinstance(rec, _AttnRec):
                self._attn_bwd(rec)
            elif isinstance(rec, _PoolRec):
                self._pool_bwd(rec)
            elif isinstance(rec, _DropRec):
                self._dropout_bwd(rec)
            else:
                self._gn_bwd(rec)
        self._temb_bwd()
        bp.finalize()

    def _wgrad(self, rec, dy, dYt, ld, Kper, S, rp, cout, shift_dy=None, Wp=None):
        """dW of one conv from the transposed output gradient dYt [rows][ld] (see csrc/train.hip).
        shift_dy = (buffer, per, margin): 3x3 / stride-1 convs whose inputs are wider than their output shift the OUTPUT gradient
        instead of the input (three dx-shifted copies of dY, ONE copy of every input source; the dy taps are the -+W offsets
        of the dY copies)."""
        L, bp, dt, es = self.L, self.bprog, self.dt, self.es
        N, Ho, Wo = dy.N, dy.H, (Wp or dy.W)  # Wo = pitch of a pixel row in the transposed operands
        ks, stride = rec.ksize, rec.stride
        taps = ks * ks
        conv = rec.conv
        cin_total = conv.weight.shape[1]
        dW = self._param_grad(conv.weight)
        ci0 = 0
        for xs in rec.srcs:
            cs = xs.C
            cs_real = min(cs, cin_total - ci0)
            ldp = round_up(cs, 4)
            s1 = ks == 3 and stride == 1
            ncopy = (1 if shift_dy else 3) if s1 else taps
            margin = round_up(Wo, 8) if (s1 and not shift_dy) else 0
            xt = bp.empty((ncopy * (cs * ld + 2 * margin),))
            xt.zero_()
            per = cs * ld + 2 * margin
            for k in range(ncopy):
                if s1:
                    gdy, gdx, pad = 1, (1 if shift_dy else k), 1
                elif ks == 3:
                    gdy, gdx, pad = k // 3, k % 3, 1
                else:
                    gdy, gdx, pad = 0, 0, 0
                self._call(L.eod_transpose_gather, ptr(xs.t), dt, xs.N, xs.H, xs.W, cs, ptr(xt) + (k * per + margin) * es, ld, Ho, Wo,
                           stride, pad, gdy, gdx, int(bool(rec.upsample)), rp)
            partial = bp.empty((S * taps * cout * ldp,), torch.float32)
            if s1 and shift_dy:
                ybuf, yper, ymargin = shift_dy
                for kx in range(3):  # A = dY shifted by -(kx-1) pixels, read at -(ky-1)*W; B = the single copy of X
                    self._bop(lambda kx=kx: bp.gemm(ybuf, xt, partial, cout, cs, Kper, ld, ld, ldp, c_f32=True, nb0=S, nb1=3,
                                                    sa=(Kper, -Wo), sb=(Kper, 0), sc=(taps * cout * ldp, 3 * cout * ldp),
                                                    a_off=kx * yper + ymargin + Wo, c_off=kx * cout * ldp))
            elif s1:
                for kx in range(3):
                    self._bop(lambda kx=kx: bp.gemm(dYt, xt, partial, cout, cs, Kper, ld, ld, ldp, c_f32=True, nb0=S, nb1=3,
                                                    sa=(Kper, 0), sb=(Kper, Wo), sc=(taps * cout * ldp, 3 * cout * ldp),
                                                    b_off=kx * per + margin - Wo, c_off=kx * cout * ldp))
            else:
                self._bop(lambda: bp.gemm(dYt, xt, partial, cout, cs, Kper, ld, ld, ldp, c_f32=True, nb0=S, nb1=taps,
                                          sa=(Kper, 0), sb=(Kper, per), sc=(taps * cout * ldp, cout * ldp), b_off=margin))
            self._call(L.eod_wgrad_reduce, ptr(partial), S, ks, cout, cs_real, ldp, ci0, cin_total, self.inv_scale, ptr(dW))
            ci0 += cs_real

    @staticmethod
    def _wgrad_splits(strips, tiles):
        """pixel-range splits of a backward-weights launch: tiles x S workgroups must FIT the chip's 512 co-resident workgroups (2 per
        CU).  Rounding S up put 513 workgroups on 512 slots for the 128 -> 128 convs (3 row taps x 171 splits): the one left over ran
        alone after all the others, a second full round (EOD_WGRAD_ROUND=up restores that for A/B)"""
        want = int(os.environ.get("EOD_WGRAD_WGS", "512"))
        s = (want + tiles - 1) // tiles if os.environ.get("EOD_WGRAD_ROUND", "down") == "up" else want // tiles
        return max(1, min(strips, s))

    def _wgrad_direct(self, rec, dy, cout):
        """3x3 / stride-1 and 1x1 backward-weights straight from the NHWC tensors: conv3x3_wgrad_kernel / gemm_tn_kernel
        (pixel-major staging + transposed LDS reads), split over pixel ranges into fp32 partial tiles"""
        L, bp, dt = self.L, self.bprog, self.dt
        conv = rec.conv
        cin_total = conv.weight.shape[1]
        dW = self._param_grad(conv.weight)
        ks = rec.ksize
        npix = dy.N * dy.H * dy.W
        strips = dy.N * (dy.H * dy.W // 64) if ks == 3 else (npix + 63) // 64
        ci0 = 0
        for xs in rec.srcs:
            cs = xs.C
            cs_real = min(cs, cin_total - ci0)
            ldp = round_up(cs, 4)
            up4 = (ks == 3 and getattr(rec, "up4_wc", None) is not None and os.environ.get("EOD_UP4", "1") != "0"
                   and (xs.W % 64 == 0 or (xs.W in (16, 32) and (xs.H * xs.W) % 64 == 0)))
            if up4:
                # parity-class form (csrc/train.hip: conv3x3_wgrad_kernel<WS, CLS>): 16 class / tap correlations of the stride-2 views of dY
                # with X at its stored resolution -- 4/9 of the MACs of the nine taps over the 2H x 2W gradient -- folded back into dW
                tiles = ((cout + 127) // 128) * ((cs + 127) // 128) * 8
                strips4 = xs.N * (xs.H * xs.W // 64)
                S = self._wgrad_splits(strips4, tiles)
                partial = bp.empty((S * 16 * cout * ldp,), torch.float32)
                t16 = bp.empty((cout * cin_total * 16,), torch.float32)
                self._call(L.eod_conv3x3_wgrad, ptr(dy.t), ptr(xs.t), dt, dy.N, xs.H, xs.W, cs, dy.H, dy.W, dy.C, cout, 2, ptr(partial), ldp, S)
                self._call(L.eod_wgrad_reduce, ptr(partial), S, 4, cout, cs_real, ldp, ci0, cin_total, self.inv_scale, ptr(t16))
                self._call(L.eod_wgrad_up4_map, ptr(t16), cout, cin_total, ptr(dW))
                ci0 += cs_real
                continue
            tiles = ((cout + 127) // 128) * ((cs + 127) // 128) * (6 if rec.stride == 2 else ks)
            S = self._wgrad_splits(strips, tiles)
            partial = bp.empty((S * ks * ks * cout * ldp,), torch.float32)
            if ks == 3:  # (ups 3 = stride-2 conv: X gathered at pixel stride 2)
                self._call(L.eod_conv3x3_wgrad, ptr(dy.t), ptr(xs.t), dt, dy.N, xs.H, xs.W, cs, dy.H, dy.W, dy.C, cout,
                           3 if rec.stride == 2 else int(bool(rec.upsample)), ptr(partial), ldp, S)
            else:
                self._call(L.eod_conv1x1_wgrad, ptr(dy.t), ptr(xs.t), dt, npix, cs, dy.C, cout, ptr(partial), ldp, S)
            self._call(L.eod_wgrad_reduce, ptr(partial), S, ks, cout, cs_real, ldp, ci0, cin_total, self.inv_scale, ptr(dW))
            ci0 += cs_real

    def _conv_bwd(self, rec):
        L, bp, dt, es = self.L, self.bprog, self.dt, self.es
        dy = self._take_grad(rec.y)
        if dy is None:
            raise EodError("training: a conv output has no gradient (graph bug)")
        N, Ho, Wo = dy.N, dy.H, dy.W
        conv = rec.conv
        cout = rec.cout_rows or conv.out_channels
        ks, stride = rec.ksize, rec.stride
        s1 = ks == 3 and stride == 1
        rp = 1 if s1 else 0
        # GEMM path: pixel rows of the transposed operands are pitched to whole 16-byte chunks (Wp >= Wo); the pad columns
        # are zero in the un-shifted operand of every product, so they never contribute
        Wp = round_up(Wo, 16 // es)
        K = N * (Ho + 2 * rp) * Wp
        BK = 128 // es
        steps = (K + BK - 1) // BK
        cin_max = max(s.C for s in rec.srcs)
        tiles = ((cout + 127) // 128) * ((cin_max + 127) // 128) * ks * ks
        S = max(1, min(steps, (768 + tiles - 1) // tiles))
        per_steps = (steps + S - 1) // S
        S = (steps + per_steps - 1) // per_steps
        Kper = per_steps * BK
        ld = S * Kper
        shift_dy = None
        # dedicated backward-weights kernel (no transposed copies) where it applies; the GEMM path otherwise
        strip_ok = Wo % 64 == 0 or (Wo in (16, 32) and (Ho * Wo) % 64 == 0)
        s2 = (ks == 3 and stride == 2 and not rec.upsample and all(x.H == 2 * Ho and x.W == 2 * Wo for x in rec.srcs))  # even maps
        geom = ((s1 or s2) and strip_ok) or (ks == 1 and stride == 1 and not rec.upsample)
        direct = (geom and self.prog.precision == "fp16"
                  and dy.C % 8 == 0 and all(x.C % 8 == 0 for x in rec.srcs)
                  and dy.t.numel() * es < 2**31 and all(x.t.numel() * es < 2**31 for x in rec.srcs)
                  and os.environ.get("EOD_WGRAD", "direct") != "gemm")
        if direct:
            # bias / timestep-projection gradients = per-channel sums of dY, taken from the NHWC tensor in place
            HW = Ho * Wo
            if getattr(dy, "csum", None) is not None:  # the gradient came straight out of a GroupNorm backward, which summed its channels on the way
                csum, Pn = dy.csum
            else:
                Pn = max(1, min(256, HW // 64))
                csum = bp.empty((N, Pn, dy.C, 2), torch.float32)
                self._call(L.eod_gn_partial, ptr(dy.t), dt, N, HW, dy.C, ptr(csum), Pn, dy.C, 0)
            if conv.bias is not None or rec.emb is not None:
                self._call(L.eod_channel_sums_finish, ptr(csum), N, Pn, dy.C, cout, self.inv_scale,
                           ptr(self._param_grad(conv.bias)) if conv.bias is not None else 0,
                           (ptr(self.dout_cat) + rec.emb[2] * 4) if rec.emb is not None else 0, self.ctx.J,
                           ptr(bp.empty((N, cout), torch.float32)))
            if rec.emb is not None:
                self._emb_layer_wgrad(rec.emb[2], rec.emb[3])
            self._wgrad_direct(rec, dy, cout)
        elif s1 and dy.C < sum(x.C for x in rec.srcs) and os.environ.get("EOD_WGRAD_SHIFT", "auto") != "x":
            # inputs wider than the output: three dx-shifted copies of dY (pad rows, +-W margins) instead of three of each input
            ymargin = round_up(Wp, 8)
            yper = dy.C * ld + 2 * ymargin
            ybuf = bp.empty((3 * yper,), zero=True)
            for kx in range(3):  # dYs_kx[co][(n, hp, w)] = dY[n][hp-1][w - (kx-1)]
                self._call(L.eod_transpose_gather, ptr(dy.t), dt, N, Ho, Wo, dy.C, ptr(ybuf) + (kx * yper + ymargin) * es, ld, Ho, Wp,
                           1, 1, 1, 2 - kx, 0, rp)
            shift_dy = (ybuf, yper, ymargin)
            dYt = ybuf[yper + ymargin:]  # the un-shifted copy (kx = 1) doubles as the plain transpose for the row sums
        else:
            dYt = bp.empty((dy.C * ld + 16,))
            self._call(L.eod_transpose_gather, ptr(dy.t), dt, N, Ho, Wo, dy.C, ptr(dYt), ld, Ho, Wp, 1, 0, 0, 0, 0, rp)
        if not direct and conv.bias is not None:  # bias gradient = row sums of dYt, in two levels (enough blocks to fill the chip)
            units = ld // BK
            nseg = max(dv for dv in range(1, min(units, 128) + 1) if units % dv == 0)
            tmp = bp.empty((nseg, cout), torch.float32)
            self._call(L.eod_rowsum_segments, ptr(dYt), dt, cout, ld, nseg, ld // nseg, self.inv_scale, ptr(tmp), cout)
            self._call(L.eod_colsum, ptr(tmp), nseg, cout, ptr(self._param_grad(conv.bias)))
        if not direct and rec.emb is not None:  # timestep-embedding projection: per-image sums of the same gradient (kept loss-scaled)
            off = rec.emb[2]
            self._call(L.eod_rowsum_segments, ptr(dYt), dt, cout, ld, N, (Ho + 2 * rp) * Wp, 1.0, ptr(self.dout_cat) + off * 4, self.ctx.J)
            self._emb_layer_wgrad(off, rec.emb[3])
        if not direct:
            self._wgrad(rec, dy, dYt, ld, Kper, S, rp, cout, shift_dy, Wp)
        if isinstance(conv, _VirtConv):  # padded copies of an attention block's projections: gradients back to the real parameters
            dWr, dbr = self._param_grad(conv.real.weight), self._param_grad(conv.real.bias)
            self.bwd.append(("dyn", lambda st, c=conv, a=dWr, b=dbr: c.gather(a, b), (dWr.data_ptr(), dbr.data_ptr())))
        if rec.res is not None:
            self._add_grad(rec.res, dy)
        if not rec.src_needs_grad:
            return
        ci0 = 0
        cin_total = conv.weight.shape[1]
        for xs in rec.srcs:
            cs = xs.C
            if getattr(rec, "up4_wc", None) is not None and bp.conv_up4_bwd_ok(dy, cs):
                # parity-class backward-data: per output parity a 2x2-tap conv of the stride-2 view of dY with the transposed class kernels,
                # accumulated in one (H x W) tile -- no (2H x 2W) dX, no 2x2 sum pool (csrc/igemm.hip: conv_up4_halo_kernel<BWD>)
                wd = bp.empty((9, cs, 4 * dy.C))
                self._add_pack_job(1, conv, wd, cpad=4 * dy.C, ci0=0, nci=cs, src=rec.up4_wc)
                g, _ = self._bop(lambda: bp.conv(dy, wd, None, cs, ksize=3, stride=1, pad=1, upsample="up4b", res=self._pop_single(xs)))
                self._add_grad(xs, g)
                ci0 += cs
                continue
            wd = bp.empty((ks * ks, cs, dy.C))
            self._add_pack_job(1, conv, wd, cpad=dy.C, ci0=ci0, nci=cs)
            odd = stride == 2 and ((xs.H % 2) or (xs.W % 2))  # stride-2 conv of an odd map (unet_openai.py:262-264, e.g. 7 -> 4)
            prev = None if (rec.upsample or odd) else self._pop_single(xs)
            g, _ = self._bop(lambda: bp.conv(dy, wd, None, cs, ksize=ks, stride=1, pad=ks // 2, upsample=(2 if stride == 2 else False),
                                             res=prev))
            if rec.upsample:
                g = self._bop(lambda: bp.resample2x(g, 2))
            if odd:  # dX lives on the (2 Ho) x (2 Wo) grid of the zero-inserted gradient: the input's last row / column does not exist
                g = self._bop(lambda: bp.resample2x(g, 4, (xs.H % 2) | ((xs.W % 2) << 1)))
            if (g.H, g.W) != (xs.H, xs.W):
                raise EodError(f"training: backward-data shape {g.H}x{g.W} != input {xs.H}x{xs.W}")
            self._add_grad(xs, g)
            ci0 += cs

    def _gn_bwd(self, rec):
        L, bp, dt = self.L, self.bprog, self.dt
        dy = self._take_grad(rec.y)
        if dy is None:
            raise EodError("training: a GroupNorm output has no gradient (graph bug)")
        gn = rec.gn
        N, HW, ctot = dy.N, dy.H * dy.W, dy.C
        groups = 32
        parts = rec.parts
        p1 = parts[1] if len(parts) == 2 else (None, 0, 0)
        mr = bp.empty((N, groups, 2), torch.float32)
        self._call(L.eod_gn_mean_rstd, ptr(parts[0][0]), parts[0][1], parts[0][2], ptr(p1[0]), p1[1], p1[2], N, HW, groups, gn.eps, ptr(mr))
        P = max(1, min(256, HW // 64))
        part = bp.empty((N, P, ctot, 2), torch.float32)
        coef = bp.empty((N, ctot, 3), torch.float32)
        gb = bp.empty((N, ctot, 2), torch.float32)
        coff = 0
        for s in rec.srcs:
            self._call(L.eod_gn_bwd_partial, ptr(s.t), ptr(dy.t), ptr(rec.ss), dt, N, HW, s.C, ptr(part), P, ctot, coff, int(rec.silu))
            coff += s.C
        if rec.film is not None:  # d[scale | shift] of the FiLM go to this block's columns of the emb_layers gradient (loss-scaled)
            fv, fj, foff = rec.film[:3]
            self._call(L.eod_gn_bwd_finalize, ptr(part), P, ctot, N, HW, groups, ptr(mr), ptr(self.prog.f32(gn.weight)), ptr(self.prog.f32(gn.bias)),
                       ptr(fv), fj, ptr(self.dout_cat) + foff * 4, self.ctx.J, ptr(coef), ptr(gb))
            self._emb_layer_wgrad(foff, rec.film[3])
        else:
            self._call(L.eod_gn_bwd_finalize, ptr(part), P, ctot, N, HW, groups, ptr(mr), ptr(self.prog.f32(gn.weight)), 0, 0, 0, 0, 0,
                       ptr(coef), ptr(gb))
        self._call(L.eod_gn_bwd_params, ptr(gb), N, ctot, self.inv_scale, ptr(self._param_grad(gn.weight)), ptr(self._param_grad(gn.bias)))
        coff = 0
        for s in rec.srcs:
            dx = bp.act(s.N, s.H, s.W, s.C)
            prev = self._pop_single(s)
            # per-channel sums of dx on the way: if dx turns out to be the whole output gradient of the conv that produced s, that conv's
            # bias / timestep-projection gradients come from them (no pass of their own over dY).  Only the direct backward-weights path
            # (fp16 storage) reads them: elsewhere the kernel skips the reduction.  (dx.csum, not Act.stats: the forward engine reads that
            # field as {sum, sum of squares} slots.)
            cs, Ps = None, 0
            if self.prog.precision == "fp16":
                Ps = L.eod_gn_bwd_apply_slabs(dt, N, HW, s.C)
                cs = bp.empty((N, Ps, s.C, 2), torch.float32, zero=True)
            self._call(L.eod_gn_bwd_apply, ptr(s.t), ptr(dy.t), ptr(rec.ss), ptr(coef), ptr(prev.t) if prev is not None else 0, dt,
                       N, HW, s.C, ctot, coff, int(rec.silu), ptr(dx.t), ptr(cs))
            dx.csum = (cs, Ps) if cs is not None else None
            self._add_grad(s, dx)
            coff += s.C

    def _emb_layer_wgrad(self, off, lin):
        """weight / bias gradient of one ResBlock's emb_layers Linear, issued as soon as the block's columns of the concatenated
        gradient exist (so that its parameters are final early and their bucket can be all-reduced under the rest of the backward)"""
        n = lin.out_features
        self._call(self.L.eod_linear_bwd_small, ptr(self.dout_cat) + off * 4, self.ctx.J, ptr(self.emb), 0, 0, ptr(self.wcat) + off * self.E * 4, 0,
                   self.N, self.E, n, 1, self.inv_scale, ptr(self._param_grad(lin.weight)), ptr(self._param_grad(lin.bias)), 0, 0)

    def _pool_bwd(self, rec):
        dy = self._take_grad(rec.y)
        if dy is None:
            raise EodError("training: a resampled tensor has no gradient (graph bug)")
        if rec.mode == 1:   # nearest 2x (+ zero row / column): sum the 2x2 blocks behind the pad
            g = self._bop(lambda: self.bprog.resample2x(dy, 2, rec.pad_tl))
        else:               # average pool with floor: the dropped last row / column of an odd map gets a zero gradient
            g = self._bop(lambda: self.bprog.resample2x(dy, 3, rec.src.H % 2 == 1))
        if (g.H, g.W) != (rec.src.H, rec.src.W):
            raise EodError("training: resample backward shape mismatch")
        self._add_grad(rec.src, g)

    def _temb_bwd(self):
        L, bp, N, E, D, J = self.L, self.bprog, self.N, self.E, self.D, self.ctx.J
        te1, te2 = self.unet.time_embed[0], self.unet.time_embed[2]
        inv = self.inv_scale
        demb = bp.empty((N, E), torch.float32)
        dpre1 = bp.empty((N, E), torch.float32)
        pre1 = bp.empty((N, E), torch.float32)
        self._t_slot = torch.zeros((N,), dtype=torch.int64, device=self.device)
        tp = self._t_slot.data_ptr()
        # emb_layers of every ResBlock: the weight / bias gradients were issued per block (_emb_layer_wgrad); here the gradient
        # w.r.t. the shared embedding, over all blocks' columns at once
        scratch = bp.empty((32, N, E), torch.float32)
        self._call(L.eod_linear_bwd_small, ptr(self.dout_cat), J, ptr(self.emb), 0, 0, ptr(self.wcat), ptr(self.emb), N, E, J, 1, inv,
                   0, 0, ptr(demb), ptr(scratch))
        self._y_slot = torch.zeros((N,), dtype=torch.int64, device=self.device)
        if self.with_y:  # emb = time_embed(...) + label_emb(y): the embedding rows get the same gradient rows
            lw = self.unet.label_emb.weight
            self._call(L.eod_embedding_bwd, ptr(demb), self._y_slot.data_ptr(), N, E, lw.shape[0], inv, ptr(self._param_grad(lw)))
        # time_embed[2]: emb = W2 h1 + b2, h1 = SiLU(pre1)
        self._call(L.eod_temb_pre1, tp, ptr(self.freqs), ptr(self.prog.f32(te1.weight)), ptr(self.prog.f32(te1.bias)), N, D, E, ptr(pre1))
        self._call(L.eod_linear_bwd_small, ptr(demb), E, ptr(self.h1), 0, 0, ptr(self.prog.f32(te2.weight)), ptr(pre1), N, E, E, 0, inv,
                   ptr(self._param_grad(te2.weight)), ptr(self._param_grad(te2.bias)), ptr(dpre1), 0)
        # time_embed[0]: pre1 = W1 sinusoid(t) + b1
        self._call(L.eod_linear_bwd_small, ptr(dpre1), E, 0, tp, ptr(self.freqs), ptr(self.prog.f32(te1.weight)), 0, N, D, E, 2, inv,
                   ptr(self._param_grad(te1.weight)), ptr(self._param_grad(te1.bias)), 0, 0)

    # ------------------------------------------------------------------ execution
    step_id = 0

    def forward(self, x, timesteps, cond=None, y=None):
        """x NCHW fp32 on the GPU, timesteps int64 [N] -> prediction NCHW fp32 (a buffer owned by the trainer)"""
        assert (y is not None) == self.with_y, "must specify y if and only if the model is class-conditional"
        if self.stale():
            raise EodError("UNetTrainer: the UNet's parameter storage moved after this trainer was built (optimizer flat buffer, "
                           ".to(), load_state_dict(assign=True)); build the optimizer first or create a new UNetTrainer")
        self.step_id += 1
        for conv, wc in self.up4_sums:  # class kernels of the parity-class upsample convs from the live weights
            w = conv.weight.detach()
            check(self.L.eod_conv_up4_weights(ptr(w), ptr(wc), w.shape[0], w.shape[1], current_stream_ptr(self.device)), "eod_conv_up4_weights")
        for vc in self.virt:
            vc.refresh()
        self._run_pack_jobs()
        for fn in self.repack:
            fn()
        self._x = x.contiguous().float()
        if timesteps.is_floating_point() and bool((timesteps != timesteps.round()).any()):
            # (inference takes fractional timesteps, eod_temb_desc.t_f32; the backward's recomputation of the sinusoid reads int64)
            raise EodError("training: fractional timesteps are not supported (train.py / model.py:40 draw integers)")
        self._t = timesteps.to(device=self.device, dtype=torch.int64).contiguous()
        self._t_slot.copy_(self._t)
        self.prog.ops[self.i_in].u.small.p[0] = self._x.data_ptr()
        if cond is not None:
            self._c = cond.contiguous().float()
            self.prog.ops[self.i_in].u.small.p[1] = self._c.data_ptr()
        self.prog.ops[self.i_t].u.temb.t = self._t_slot.data_ptr()
        for idx in self._drop_ops:
            self.prog.ops[idx].u.small.i[2] = self.step_id & 0x7fffffff
        if y is not None:
            self._y_slot.copy_(y.to(torch.int64))
            self.prog.ops[self.i_t].u.temb.y = self._y_slot.data_ptr()
        self.prog._arr = None
        self.prog.run()
        return self.pred

    def backward(self, dpred, assign=True, allreduce=False, group=None):
        """dpred = dLoss/dpred (NCHW fp32).  assign=True: sets `param.grad` (fp32) of every UNet parameter;
        assign=False: returns the gradients in `unet.parameters()` order (clones, for torch.autograd accumulation).
        allreduce=True (data-parallel training): the gradient buckets are averaged over the ranks of `group`, each bucket's
        all-reduce is issued as soon as the bucket is complete and overlaps the remaining backward launches."""
        st = current_stream_ptr(self.device)
        torch.mul(dpred, self.loss_scale, out=self.dpred)
        L = self.L
        pending, world, works = [], 1, []
        if allreduce:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                world = dist.get_world_size(group)
            if world > 1 or os.environ.get("EOD_FORCE_ALLREDUCE") == "1":
                if not hasattr(self, "_buckets"):
                    self._plan_buckets()
                pending = list(self._buckets)
        for i, item in enumerate(self.bwd):
            if item[0] == "op":
                check(L.eod_program_run(C.byref(item[1]), 1, st), "backward op")
            elif item[0] == "dyn":
                item[1](st)
            else:
                check(item[1](*item[2], st), item[1].__name__)
            while pending and pending[0][0] <= i:
                _, (lo, hi) = pending.pop(0)
                works.append(dist.all_reduce(self.flat_grad[lo:hi], group=group, async_op=True))
        for _, (lo, hi) in pending:  # buckets nobody writes (cannot happen for a UNet, kept for safety)
            works.append(dist.all_reduce(self.flat_grad[lo:hi], group=group, async_op=True))
        for w in works:
            w.wait()
        if works and world > 1:
            self.flat_grad.div_(world)
        # parameters the forward never uses (the reference's dead `nout` / `conv_out` head, unet_openai.py:744) get NO gradient,
        # exactly like autograd: torch.optim.AdamW then skips them (no weight decay), and so does optim.AdamW
        if not assign:
            return [self.pgrad[p].to(p.dtype).clone() if p in self.pgrad else None for p in self.unet.parameters()]
        for p, g in self.pgrad.items():
            p.grad = g
        return None


class _UNetTrainFn(torch.autograd.Function):
    """autograd bridge: the UNet parameters are inputs of the node, so `loss.backward()` accumulates the HIP-computed
    gradients into `param.grad` exactly like the reference's autograd graph does (train.py:118)."""

    @staticmethod
    def forward(ctx, trainer, x, timesteps, cond, y, *params):
        ctx.trainer = trainer
        ctx.n_params = len(params)
        pred = trainer.forward(x, timesteps, cond, y)
        ctx.step_id = trainer.step_id
        return pred.clone()  # the trainer's own buffer is overwritten by the next forward

    @staticmethod
    def backward(ctx, dpred):
        tr = ctx.trainer
        if ctx.step_id != tr.step_id:
            raise EodError("training: backward called after another forward of the same shape (saved activations were overwritten)")
        grads = tr.backward(dpred.contiguous().float(), assign=False)
        return (None, None, None, None, None) + tuple(grads)


def unet_train_forward(unet, x, timesteps, cond=None, y=None):
    """UNetModel.forward in training mode (called when autograd is enabled and parameters require grad)."""
    N, cx, H, W = x.shape
    ccond = 0 if cond is None else cond.shape[1]
    cache = unet.__dict__.setdefault("_eod_trainers", {})
    key = (N, cx, ccond, H, W, str(x.device), unet.precision, bool(unet.training))  # (train / eval differ by the dropout ops)
    tr = cache.get(key)
    if tr is not None and tr.stale():  # parameter storage moved (optimizer flat buffer, .to(), ...): its pointers are dead
        del cache[key]
        tr = None
    if tr is None:
        import os
        scale = float(os.environ.get("EOD_LOSS_SCALE", "1024" if unet.precision == "fp16" else "1"))
        while len(cache) >= 2:  # a trainer owns every saved activation of its shape (tens of GiB at 256x256x16): keep two shapes
            cache.pop(next(iter(cache)))
        tr = cache[key] = UNetTrainer(unet, N, H, W, x.device, cond_channels=ccond, loss_scale=scale)
    params = [p for p in unet.parameters()]
    return _UNetTrainFn.apply(tr, x.detach(), timesteps, None if cond is None else cond.detach(), y, *params)
