"""Program builder / executor for the HIP path.

A `Program` is the host-side view of one static launch plan: a flat list of C descriptors
(`_lib.Op`), the device buffers they point into (activations are channels-last, see include/eodiff.h)
and a few named *bindings* -- descriptor fields that are re-pointed at caller tensors on every run
(input image, timesteps, labels, output).  `Program.run()` is ONE FFI call: `eod_program_run` walks
the descriptor array natively and enqueues every kernel on the current HIP stream.

torch is used here only for device memory (torch.empty / zeros), stream handles and dtype plumbing.
"""
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import (OP_ACT_BOUND, OP_BOUND_AFFINE, OP_ATTN, OP_ATTN_NAT, OP_CONV, OP_DROPOUT, OP_GEMM, OP_GN_APPLY, OP_GN_FINALIZE, OP_GN_PARTIAL, OP_POOL, OP_SOFTMAX,
                   OP_TEMB, OP_TO_NCHW, OP_TO_NHWC, ConvDesc, GemmDesc, Op, TembDesc, check, ptr)

AB = 32  # entries per image of an activation bound table (csrc/common.h: EOD_AB)

# precision mode -> storage dtype of activations.  "fp32x3": fp32 storage, the 3x3 halo convs compute every product as three fp16
# MFMAs on split operands (csrc/igemm.hip, eod_conv_desc.w_split); everything else runs exactly as in "fp32".
PRECISIONS = {"fp32": torch.float32, "fp16": torch.float16, "fp32x3": torch.float32}


def current_stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise _lib.EodError(
            f"{what}: tensor is on '{getattr(t, 'device', None)}'. eo_diffusion_amd runs on MI355X (HIP) only; "
            "there is no CPU / eager fallback.")


class Act:
    """Channels-last activation handle: tensor of shape [N, H, W, C] in the storage dtype."""
    __slots__ = ("t", "N", "H", "W", "C", "stats", "bound", "presplit", "csum")

    def __init__(self, t, N, H, W, C, stats=None):
        self.t, self.N, self.H, self.W, self.C = t, N, H, W, C
        self.stats = stats  # (fp32 tensor [N][P][C][2], P): GroupNorm partial sums emitted by the producing conv
        self.csum = None   # training: (fp32 [N][P][C][2], P) per-channel sums a GroupNorm backward emitted with this gradient tensor
        self.bound = None   # fp32x3 programs: bound table [N][32] of this tensor once something has produced one (Program.bound_of)
        self.presplit = False  # fp32x3: the tensor holds [8 x fp16 hi | 8 x fp16 lo] groups of s_n * x (s_n from `bound`), for ONE
        #                        split-fp16 consumer on the generic conv kernel (eod_conv_desc.x_presplit); nothing else can read it

    @property
    def HW(self):
        return self.H * self.W


class _LazyConvW:
    """Conv weight of an fp32x3 program: its packed format (plain fp32 or split fp16 pairs + device scale) is decided by the conv
    that consumes it (Program.conv asks eod_conv_split_ok for its geometry); each format is packed at most once."""

    def __init__(self, prog, weight, cin_pad):
        self.prog, self.weight, self.cin_pad = prog, weight, cin_pad
        self._plain = self._split = None

    def plain(self):
        if self._plain is None:
            self._plain = self.prog._pack_conv_plain(self.weight, self.cin_pad)
        return self._plain

    def split(self):
        if self._split is None:
            prog = self.prog
            w = prog.f32(self.weight)
            cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
            cin_pad = self.cin_pad or cin
            dst = prog.empty((k * k, cout, cin_pad), torch.float32)  # 4 bytes per element: [8 x fp16 hi | 8 x fp16 lo] per 8 channels
            scale = prog.empty((_lib.WSCALE_ROWS + cout,), torch.float32)  # {s, 1/(16 s), -, -} + one int32 row exponent per output row
            check(prog.L.eod_pack_conv_weight_split(ptr(w), ptr(dst), ptr(scale), cout, cin, k, cin_pad,
                                                    current_stream_ptr(prog.device)), "pack_conv_weight_split")
            self._split = (dst, scale)
        return self._split

    def split_pair(self, weight2):
        """this 3x3 weight and a 1x1 weight [Cout][Cin2] that feeds the same accumulator (the ResBlock's fused skip conv): both in the
        split format under ONE scale -> (dst, scale, dst2)"""
        prog = self.prog
        w, w2 = prog.f32(self.weight), prog.f32(weight2)
        cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
        cin2 = w2.shape[1]
        assert w2.shape[0] == cout and w2.numel() == cout * cin2, "skip weight must be [Cout][Cin2][1][1]"
        cin_pad = self.cin_pad or cin
        dst = prog.empty((k * k, cout, cin_pad), torch.float32)
        dst2 = prog.empty((1, cout, cin2), torch.float32)
        scale = prog.empty((_lib.WSCALE_ROWS + cout,), torch.float32)
        check(prog.L.eod_pack_conv_weight_split_pair(ptr(w), ptr(dst), ptr(w2), ptr(dst2), ptr(scale), cout, cin, k, cin_pad, cin2,
                                                     current_stream_ptr(prog.device)), "pack_conv_weight_split_pair")
        return dst, scale, dst2


class Program:
    def __init__(self, device, precision):
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {list(PRECISIONS)}")
        self.device = torch.device(device)
        self.precision = precision
        self.tdtype = PRECISIONS[precision]
        self.split = precision == "fp32x3"
        self.dt = _lib.dtype_id(self.tdtype)
        self.epc = 16 // self.tdtype.itemsize  # elements per 16-byte chunk
        self.ops = []
        self.keep = []       # tensors owned by the program
        self.bindings = {}   # name -> list[(op_index, setter)]
        self._arr = None
        self._timer = None
        self.L = _lib.lib()
        self.nbytes = 0
        self._bounds = {}        # fp32x3: (data_ptr of every source of a virtual concat) -> bound table [N][32] of the raw concat
        self.drop_ops = []       # indices of OP_DROPOUT descriptors (their `step` field advances every run)
        self.drop_seed = int(torch.initial_seed()) & (2**63 - 1)
        self.drop_step = 0

    # ------------------------------------------------------------------ memory
    def empty(self, shape, dtype=None, zero=False):
        dtype = dtype or self.tdtype
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
        self.keep.append(t)
        self.nbytes += t.numel() * t.element_size()
        return t

    def act(self, N, H, W, C, zero=False):
        return Act(self.empty((N, H, W, C), zero=zero), N, H, W, C)

    def own(self, t):
        self.keep.append(t)
        return t

    def f32(self, param):
        """fp32 contiguous device view of a parameter (biases, GN affine, MLP weights stay fp32)."""
        t = param.detach()
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
        return self.own(t)

    # ------------------------------------------------------------------ weight packing (one-off, at build time)
    def pack_conv(self, weight, cin_pad=None):
        """Conv2d/Conv1d weight (OIHW / OIK fp32) -> [tap][Cout][cin_pad] in the storage dtype (fp32x3 programs: a handle whose
        format the consuming conv chooses)."""
        if self.split and weight.dim() == 4:
            return _LazyConvW(self, weight, cin_pad)
        return self._pack_conv_plain(weight, cin_pad)

    def pack_conv_up4(self, weight):
        """3x3 weight (OIHW fp32) of a conv over the nearest-2x upsampling of its input -> the packed [4*Cout][Cin][3][3] tensor of the
        parity-class form (eod_conv_up4_ok): output pixel (2i+p, 2j+q) reads stored rows {i-1+p, i+p} only, so class (p, q) is a 2x2-tap
        conv whose taps are sums of the original ones; row block 2p+q holds that kernel in tap slots dy' in {p, p+1}, dx' in {q, q+1}."""
        w = self.f32(weight)
        cout, cin = w.shape[0], w.shape[1]
        wc = torch.empty((4 * cout, cin, 3, 3), dtype=torch.float32, device=self.device)  # classes (0,0), (0,1), (1,0), (1,1)
        check(self.L.eod_conv_up4_weights(ptr(w), ptr(wc), cout, cin, current_stream_ptr(self.device)), "conv_up4_weights")
        return self.pack_conv(self.own(wc))

    def _pack_conv_plain(self, weight, cin_pad=None):
        w = self.f32(weight)
        cout, cin = w.shape[0], w.shape[1]
        k = w.shape[2] if w.dim() >= 3 else 1
        cin_pad = cin_pad or cin
        dst = self.empty((k * k if w.dim() == 4 else 1, cout, cin_pad))
        ks = k if w.dim() == 4 else 1
        check(self.L.eod_pack_conv_weight(ptr(w), ptr(dst), self.dt, cout, cin, ks, cin_pad,
                                          current_stream_ptr(self.device)), "pack_conv_weight")
        return dst

    def pack_conv_tapmajor(self, weight, cin_pad):
        """3x3 Conv2d weight of a thin input (OIHW fp32) -> [Cout][ldk], k = tap*cin_pad + c (conv(..., w_tapmajor=True))."""
        w = self.f32(weight)
        cout, cin = w.shape[0], w.shape[1]
        assert w.dim() == 4 and w.shape[2] == 3 and w.shape[3] == 3
        ldk = self.L.eod_conv_tapmajor_ldk(cin_pad, self.dt)
        if self.split:  # fp32x3 program: split-fp16 pairs + device scale (conv() sets w_split)
            dst, scale = self.empty((cout, ldk), torch.float32), self.empty((_lib.WSCALE_ROWS + cout,), torch.float32)
            check(self.L.eod_pack_conv_weight_tapmajor_split(ptr(w), ptr(dst), ptr(scale), cout, cin, cin_pad,
                                                             current_stream_ptr(self.device)), "pack_conv_weight_tapmajor_split")
            return ("split", dst, scale)
        dst = self.empty((cout, ldk))
        check(self.L.eod_pack_conv_weight_tapmajor(ptr(w), ptr(dst), self.dt, cout, cin, cin_pad,
                                                   current_stream_ptr(self.device)), "pack_conv_weight_tapmajor")
        return dst

    def pack_rows(self, weight2d, row_index=None):
        """rows of an fp32 [R][K] matrix (optionally gathered by row_index) -> storage dtype [r][K]."""
        w = self.f32(weight2d)
        rows = w.shape[0] if row_index is None else len(row_index)
        cols = w.shape[1]
        rm = None
        if row_index is not None:
            rm = self.own(torch.tensor(row_index, dtype=torch.int32, device=self.device))
        dst = self.empty((rows, cols))
        check(self.L.eod_pack_rows(ptr(w), w.stride(0), ptr(rm), ptr(dst), cols, self.dt, rows, cols,
                                   current_stream_ptr(self.device)), "pack_rows")
        return dst

    # ------------------------------------------------------------------ op emission
    def _push(self, kind):
        op = Op()
        op.kind = kind
        self.ops.append(op)
        self._arr = None
        return op, len(self.ops) - 1

    def bind(self, name, op_index, setter):
        self.bindings.setdefault(name, []).append((op_index, setter))

    # ------------------------------------------------------------------ activation bound tables (fp32x3 programs, csrc/common.h)
    def bound_of(self, srcs):
        """Bound table [N][32] (fp32, device) of the virtual concat of `srcs` (Acts) as it is stored -- every split-fp16 consumer of a
        tensor that is NOT behind a fused GroupNorm derives its per-image operand scale from it.  Taken, in this order, from: a GroupNorm
        finalize that already covered exactly these tensors (gn_stats), the tensor's own table (a producer that knows a bound: the
        normalising pass, resampling, attention), the partial sums of the producing convs' epilogues (one tiny launch, no pass over the
        data), a direct max|x| pass.  None outside fp32x3 programs."""
        if not self.split:
            return None
        key = tuple(s.t.data_ptr() for s in srcs)
        hit = self._bounds.get(key)
        if hit is not None:
            return hit
        if len(srcs) == 1 and srcs[0].bound is not None:
            return srcs[0].bound
        ab = self.empty((srcs[0].N, AB), torch.float32, zero=True)
        if all(s.stats is not None for s in srcs) and len(srcs) <= 2:
            s0, s1 = srcs[0], (srcs[1] if len(srcs) > 1 else None)
            self._small(OP_ACT_BOUND, p=(0, ptr(s0.stats[0]), ptr(s1.stats[0]) if s1 else 0, ptr(ab)), l=(0,),
                        i=(self.dt, s0.N, s0.stats[1], s0.C, s1.stats[1] if s1 else 0, s1.C if s1 else 0))
        elif len(srcs) == 1:
            x = srcs[0]
            self._small(OP_ACT_BOUND, p=(ptr(x.t), 0, 0, ptr(ab)), l=(x.t.numel() // x.N,), i=(self.dt, x.N, 0, 0, 0, 0))
        else:  # several sources without epilogue statistics: one direct pass each, accumulated entry-wise into the same table
            for k, x in enumerate(srcs):
                self._small(OP_ACT_BOUND, p=(ptr(x.t), 0, 0, ptr(ab)), l=(x.t.numel() // x.N,), i=(self.dt, x.N, 0, 0, 0, 0, int(k > 0)))
        self._bounds[key] = ab
        if len(srcs) == 1:
            srcs[0].bound = ab
        return ab

    def bound_of_tensor(self, t, n):
        """direct max|x| bound table of a plain tensor whose first axis splits into n images (operands of the x3 GEMMs)"""
        if not self.split:
            return None
        ab = self.empty((n, AB), torch.float32, zero=True)
        self._small(OP_ACT_BOUND, p=(ptr(t), 0, 0, ptr(ab)), l=(t.numel() // n,), i=(self.dt, n, 0, 0, 0, 0))
        return ab

    def linear_bound(self, ab_in, weight2d, bias):
        """A-PRIORI bound table of y = W x + b from the table of x: |y| <= (max row L1 norm of W) * max|x| + max|b|.  The two scalars
        are condensed from the parameters once per plan (eod_weight_l1max, build time), the table itself is one tiny op per run.  A
        producer that writes its output pre-split (the qkv projection in front of the fused attention) scales by it."""
        w = self.f32(weight2d)
        coef = torch.zeros((2,), dtype=torch.float32, device=self.device)
        check(self.L.eod_weight_l1max(ptr(w), w.shape[0], w.numel() // w.shape[0], ptr(bias), ptr(coef), current_stream_ptr(self.device)),
              "weight_l1max")
        self.own(coef)
        n = ab_in.shape[0]
        ab = self.empty((n, AB), torch.float32, zero=True)
        self._small(OP_BOUND_AFFINE, p=(ptr(ab_in), ptr(coef), ptr(ab)), i=(n,))
        return ab

    def conv_up4_ok(self, x, cout):
        """True if the library has the parity-class form of `3x3 conv over the nearest-2x upsampling of x` for this geometry"""
        if os.environ.get("EOD_UP4", "1") == "0" or self.precision == "fp32":
            return False
        d = ConvDesc()
        d.dtype, d.N, d.H, d.W, d.C0, d.C1, d.Cout = self.dt, x.N, x.H, x.W, x.C, 0, cout
        d.ksize, d.stride, d.pad, d.upsample, d.pad_tl = 3, 1, 1, 3, 0
        d.w_split = int(self.split)
        return bool(self.L.eod_conv_up4_ok(C.byref(d)))

    def conv_up4_bwd_ok(self, dy, cout):
        """True if the library has the parity-class backward-data (dX at half the resolution of dy, `cout` channels) for this geometry"""
        if os.environ.get("EOD_UP4", "1") == "0" or self.precision != "fp16" or dy.H % 2 or dy.W % 2:
            return False
        d = ConvDesc()
        d.dtype, d.N, d.H, d.W, d.C0, d.C1, d.Cout = self.dt, dy.N, dy.H, dy.W, dy.C, 0, cout
        d.ksize, d.stride, d.pad, d.upsample, d.pad_tl, d.Ho, d.Wo = 3, 1, 1, 4, 0, dy.H // 2, dy.W // 2
        return bool(self.L.eod_conv_up4_bwd_ok(C.byref(d)))

    def conv_skip_ok(self, x, cout, srcs):
        """True if the library can run `conv3x3(x) + conv1x1(cat(srcs))` (a ResBlock's out_layers conv + skip_connection,
        unet_openai.py:385) as ONE launch for this geometry (eod_conv_desc.skip_x)"""
        if len(srcs) > 2 or any((s.N, s.H, s.W) != (x.N, x.H, x.W) for s in srcs) or self.precision == "fp32":
            return False
        d = ConvDesc()
        d.dtype, d.N, d.H, d.W, d.C0, d.C1, d.Cout = self.dt, x.N, x.H, x.W, x.C, 0, cout
        d.ksize, d.stride, d.pad, d.Ho, d.Wo = 3, 1, 1, x.H, x.W
        d.w_split = int(self.split)
        d.skip_C0, d.skip_C1 = srcs[0].C, (srcs[1].C if len(srcs) > 1 else 0)
        return bool(self.L.eod_conv_skip_ok(C.byref(d)))

    def conv(self, x, w_packed, bias, cout, *, x2=None, ksize=3, stride=1, pad=1, upsample=False, pad_tl=False,
             cbias=None, cbias_stride=0, res=None, out_nchw_f32=False, out=None, stats=False, gn=None, w_tapmajor=False, skip=None,
             y_presplit_bound=None):
        """gn = (scale_shift tensor from gn_stats(), silu): GroupNorm(+SiLU) of the conv INPUT.  Fused into the conv's
        patch staging when the library can (eod_conv_gn_fusable), otherwise applied by a separate pass first.
        skip = (srcs, weight, bias): add conv1x1(cat(srcs)) with that OI11 weight / bias in the same launch (only where
        conv_skip_ok(x, cout, srcs); w_packed must be pack_conv(weight3x3) of this program)."""
        if gn is not None:
            probe = ConvDesc()
            probe.dtype, probe.N, probe.H, probe.W = self.dt, x.N, x.H, x.W
            probe.C0, probe.C1, probe.Cout = x.C, (x2.C if x2 is not None else 0), cout
            probe.ksize, probe.stride, probe.pad, probe.upsample, probe.pad_tl = ksize, stride, pad, int(upsample), int(pad_tl)
            probe.out_nchw_f32 = int(out_nchw_f32)
            probe.w_split = int(self.split and isinstance(w_packed, _LazyConvW))  # (the fusion threshold depends on the product form)
            if not self.L.eod_conv_gn_fusable(C.byref(probe)):
                x = self.gn_apply([x] + ([x2] if x2 is not None else []), gn[0], silu=gn[1])  # (carries the normalised tensor's bound table)
                x2, gn = None, None
        if upsample == "up4":  # w_packed = pack_conv_up4(weight): the parity-class form of the nearest-2x conv
            upsample = 3
        if upsample == "up4b":  # its backward-data: x = dY on the (2Ho x 2Wo) grid, w_packed = dgrad packing of the class kernels
            upsample = 4
        ups = 2 if upsample else 1
        heff, weff = x.H * ups + int(pad_tl), x.W * ups + int(pad_tl)
        ho = (heff + 2 * pad - ksize) // stride + 1
        wo = (weff + 2 * pad - ksize) // stride + 1
        if upsample == 4:
            ho, wo = x.H // 2, x.W // 2
        ab = ab_skip = None
        if self.split:
            # fp32x3: the kernel derives its per-image operand scale from the bound table of what it splits -- the normalised tensor
            # behind a fused GroupNorm (its finalize wrote that table), the stored tensor otherwise; likewise for the fused skip conv's
            # input.  (Looked up BEFORE the conv is pushed: a table that does not exist yet is one more op in front of it.)
            ab = getattr(gn[0], "eod_bound_norm", None) if gn is not None else self.bound_of([x] + ([x2] if x2 is not None else []))
            if skip is not None:
                ab_skip = self.bound_of(list(skip[0]))
        op, idx = self._push(OP_CONV)
        d = op.u.conv
        d.x, d.x2 = ptr(x.t), ptr(x2.t) if x2 is not None else 0
        d.bias, d.cbias, d.cbias_stride = ptr(bias), ptr(cbias), cbias_stride
        d.res = ptr(res.t) if res is not None else 0
        d.dtype, d.N, d.H, d.W = self.dt, x.N, x.H, x.W
        d.C0, d.C1, d.Cout = x.C, (x2.C if x2 is not None else 0), cout
        d.ksize, d.stride, d.pad, d.upsample, d.pad_tl = ksize, stride, pad, int(upsample), int(pad_tl)
        d.Ho, d.Wo, d.out_nchw_f32, d.alpha = ho, wo, int(out_nchw_f32), 1.0
        d.w_tapmajor = int(w_tapmajor)
        if x.presplit or (x2 is not None and x2.presplit):
            assert self.split and gn is None and skip is None and (x2 is None or x2.presplit == x.presplit), "pre-split input: plain split conv only"
            d.x_presplit = 1
        if skip is not None:
            ssrc, sw, sb = skip
            assert res is None and x2 is None and self.conv_skip_ok(x, cout, ssrc), "fused skip conv: ask conv_skip_ok first"
            d.skip_x, d.skip_C0 = ptr(ssrc[0].t), ssrc[0].C
            if len(ssrc) > 1:
                d.skip_x2, d.skip_C1 = ptr(ssrc[1].t), ssrc[1].C
            op._skip_c = d.skip_C0 + d.skip_C1
            if self.split:
                wt, wscale, wt2 = w_packed.split_pair(sw)
                d.w, d.w_split, d.w_scale, d.skip_w = ptr(wt), 1, ptr(wscale), ptr(wt2)
            elif isinstance(sw, tuple) and sw[0] == "packed":  # [1][Cout][skip channels] that the caller (re)packs itself (training step)
                d.w, d.skip_w = ptr(w_packed), ptr(sw[1])
            else:
                d.w, d.skip_w = ptr(w_packed), ptr(self._pack_conv_plain(sw))
            if sb is not None:  # the skip conv's bias: the per-sample bias slot with stride 0 (both vectors stay live parameters)
                assert cbias is None, "fused skip conv: the per-sample bias slot carries the skip bias"
                d.cbias, d.cbias_stride = ptr(self.f32(sb)), 0
        elif isinstance(w_packed, tuple) and w_packed[0] == "split":  # pre-split weights (thin-input first conv of an fp32x3 program)
            d.w, d.w_split, d.w_scale = ptr(w_packed[1]), 1, ptr(w_packed[2])
        elif isinstance(w_packed, _LazyConvW):  # fp32x3: split-fp16 product where the library has it for this geometry
            if self.L.eod_conv_split_ok(C.byref(d)):
                wt, wscale = w_packed.split()
                d.w, d.w_split, d.w_scale = ptr(wt), 1, ptr(wscale)
            else:
                d.w = ptr(w_packed.plain())
        else:
            d.w = ptr(w_packed)
        if gn is not None:
            d.gn_scale_shift, d.gn_silu = ptr(gn[0]), int(gn[1])
        if d.w_split:
            assert ab is not None and (skip is None or ab_skip is not None), "fp32x3: no bound table for this conv input"
            d.a_bound, d.skip_bound = ptr(ab), ptr(ab_skip)
        if y_presplit_bound is not None:  # the output is written pre-split, scaled from this a-priori table (linear_bound)
            assert d.w_split and not stats and not out_nchw_f32, "pre-split output: split-fp16 conv on the generic kernel, no statistics"
            d.y_presplit_bound = ptr(y_presplit_bound)
        assert not d.x_presplit or d.w_split, "a pre-split tensor needs the split-fp16 conv (channel counts that are multiples of 8)"
        wsz = self.L.eod_conv_workspace_size(C.byref(d))
        if wsz > 0:  # split-K partial tiles (small maps)
            ws = self.empty((wsz // 4,), torch.float32)
            d.workspace, d.workspace_bytes = ptr(ws), wsz
        if out_nchw_f32:
            y = None  # bound by the caller (external NCHW fp32 tensor)
        else:
            y = out if out is not None else self.act(x.N, ho, wo, cout)
            d.y = ptr(y.t)
            if y_presplit_bound is not None:
                y.presplit, y.bound = True, y_presplit_bound
            if stats:
                slots = self.L.eod_conv_stats_slots(C.byref(d))
                if slots > 0:  # the epilogue emits the next GroupNorm's per-channel partial sums for free
                    st = self.empty((x.N, slots, cout, 2), torch.float32)
                    d.stats, d.stats_slots = ptr(st), slots
                    y.stats = (st, slots)
        if x2 is not None:
            assert (x2.N, x2.H, x2.W) == (x.N, x.H, x.W)
        if res is not None:
            assert (res.N, res.H, res.W, res.C) == (x.N, ho, wo, cout), "residual shape"
        return y, idx

    def gemm(self, a, b, c, M, N, K, lda, ldb, ldc, *, bias=None, bias_mode=1, res=None, alpha=1.0, c_f32=False,
             nb0=1, nb1=1, sa=(0, 0), sb=(0, 0), sc=(0, 0), a_off=0, b_off=0, c_off=0, a_bound=None, b_bound=None):
        """C = alpha*A.B^T (+bias)(+res); a/b/c are tensors, *_off element offsets into them.  fp32x3 programs run it as the split-fp16
        product (x3) when K % 8 == 0: a_bound / b_bound = bound table [nb0][32] of that operand (bound_of / bound_of_tensor /
        weight_bound; indexed by the outer batch index = image), None = the operand is known to stay below 4094 in magnitude (softmax
        weights).  Activations stacked along M (nb0 == 1 over several images) have no per-image scale here: run those as 1x1 convs."""
        op, idx = self._push(OP_GEMM)
        d = op.u.gemm
        es = self.tdtype.itemsize
        d.a = ptr(a) + a_off * es
        d.b = ptr(b) + b_off * es
        d.c = ptr(c) + c_off * (4 if c_f32 else es)
        d.bias = ptr(bias)
        d.res = ptr(res)
        d.lda, d.ldb, d.ldc = lda, ldb, ldc
        d.sa0, d.sa1, d.sb0, d.sb1, d.sc0, d.sc1 = sa[0], sa[1], sb[0], sb[1], sc[0], sc[1]
        d.dtype, d.M, d.N, d.K, d.nb0, d.nb1 = self.dt, M, N, K, nb0, nb1
        d.bias_mode, d.c_f32, d.alpha = (bias_mode if bias is not None else 0), int(c_f32), alpha
        d.x3 = int(self.split and K % 8 == 0)  # fp32x3 mode: split-fp16 products
        if d.x3:
            for t in (a_bound, b_bound):
                assert t is None or tuple(t.shape) == (nb0, AB), "gemm: bound tables are [nb0][32]"
            d.a_bound, d.b_bound = ptr(a_bound), ptr(b_bound)
        return idx

    def weight_bound(self, w, n):
        """bound table [n][32] of a parameter tensor that every image shares (an operand of an x3 GEMM): one max|w| pass at BUILD time
        (parameters are constants of a plan), the row repeated per image"""
        if not self.split:
            return None
        one = torch.zeros((1, AB), dtype=torch.float32, device=self.device)
        check(self.L.eod_act_bound(ptr(w), _lib.dtype_id(w.dtype), 1, w.numel(), 0, 0, 0, 0, 0, 0, ptr(one), 0, current_stream_ptr(self.device)),
              "act_bound")
        return self.own(one.expand(n, AB).contiguous())

    def _small(self, kind, p=(), l=(), i=(), f=()):
        op, idx = self._push(kind)
        s = op.u.small
        for k, v in enumerate(p):
            s.p[k] = v
        for k, v in enumerate(l):
            s.l[k] = v
        for k, v in enumerate(i):
            s.i[k] = v
        for k, v in enumerate(f):
            s.f[k] = v
        return idx

    def gn_stats(self, srcs, gamma, beta, *, eps=1e-5, groups=32, film=None, film_stride=0):
        """GroupNorm32 [+FiLM] statistics over the virtual channel-concat of `srcs` (1 or 2 Acts) folded with the
        affine parameters: returns the fp32 {scale, shift} table [N][sum(C)][2] (y = x*scale + shift)."""
        x0 = srcs[0]
        N, H, W = x0.N, x0.H, x0.W
        HW = H * W
        ctot = sum(s.C for s in srcs)
        if ctot % groups:
            raise ValueError(f"GroupNorm: {ctot} channels not divisible by {groups} groups")
        parts = []
        for s in srcs:
            if s.stats is not None:       # emitted by the producing conv's epilogue
                parts.append((s.stats[0], s.stats[1], s.C))
            else:
                P = max(1, min(256, HW // 64))
                part = self.empty((N, P, s.C, 2), torch.float32)
                self._small(OP_GN_PARTIAL, p=(ptr(s.t), ptr(part)), i=(self.dt, N, HW, s.C, P, s.C, 0))
                parts.append((part, P, s.C))
        if len(parts) > 2:
            raise ValueError("GroupNorm over more than two concatenated sources is not supported")
        self.last_gn_parts = parts  # (partial-sum tensor, P, C) per source: the training path derives mean / rstd from them
        ss = self.empty((N, ctot, 2), torch.float32)
        p1 = parts[1] if len(parts) == 2 else (None, 0, 0)
        ab_raw = ab_norm = None
        if self.split and groups <= AB:
            # fp32x3: the finalize also writes the bound tables its consumers scale their split operands by (csrc/common.h): of the
            # normalised tensor (the conv behind this GroupNorm) and of the stored one (skip / resampling convs over the same tensors)
            ab_raw, ab_norm = self.empty((N, AB), torch.float32, zero=True), self.empty((N, AB), torch.float32, zero=True)
            self._bounds.setdefault(tuple(s.t.data_ptr() for s in srcs), ab_raw)
            if len(srcs) == 1 and srcs[0].bound is None:
                srcs[0].bound = ab_raw
        self._small(OP_GN_FINALIZE,
                    p=(ptr(parts[0][0]), ptr(gamma), ptr(beta), ptr(film) if film is not None else 0, ptr(ss), ptr(p1[0]), ptr(ab_raw), ptr(ab_norm)),
                    l=(HW, film_stride), i=(N, parts[0][1], parts[0][2], groups, p1[1], p1[2]), f=(eps,))
        ss.eod_bound_norm = ab_norm
        return ss

    def gn_apply(self, srcs, ss, *, silu, split_out=False):
        """y = act(x*scale + shift) as a separate pass; materialises the (normalised) concat of `srcs`.  split_out (fp32x3): y is
        written PRE-SPLIT for its one consumer, a split-fp16 conv on the generic kernel (the qkv projection behind AttentionBlock.norm):
        that conv then DMAs finished [hi | lo] rows into LDS instead of re-splitting its pixel rows every K-step."""
        x0 = srcs[0]
        N, H, W = x0.N, x0.H, x0.W
        ctot = sum(s.C for s in srcs)
        y = self.act(N, H, W, ctot)
        y.bound = getattr(ss, "eod_bound_norm", None)  # (|SiLU(v)| <= |v|: the table holds with and without the activation)
        split_out = bool(split_out and self.split and y.bound is not None and ctot % 8 == 0 and all(s.C % 8 == 0 for s in srcs))
        y.presplit = split_out
        coff = 0
        for s in srcs:
            self._small(OP_GN_APPLY, p=(ptr(s.t), ptr(ss), ptr(y.t), ptr(y.bound) if split_out else 0),
                        i=(self.dt, N, H * W, s.C, ctot, coff, int(silu)))
            coff += s.C
        return y

    def group_norm(self, srcs, gamma, beta, *, silu, eps=1e-5, groups=32, film=None, film_stride=0, split_out=False):
        """GroupNorm32 [+FiLM] [+SiLU] as stats + separate apply pass (returns the normalised Act; split_out: see gn_apply)."""
        ss = self.gn_stats(srcs, gamma, beta, eps=eps, groups=groups, film=film, film_stride=film_stride)
        return self.gn_apply(srcs, ss, silu=silu, split_out=split_out)

    def attention(self, qk, vT, out, N, T, Cc, heads, d, dpad, ld_qk, ldt, k_off, lse=None):
        """fused flash-style attention (fp16, head dim <= 64): eod_attention_fwd; lse: optional fp32 [N][heads][T] output"""
        op, idx = self._push(OP_ATTN)
        a = op.u.attn
        a.qk, a.vT, a.out, a.lse = ptr(qk), ptr(vT), ptr(out), ptr(lse)
        a.ld_qk, a.ldt = ld_qk, ldt
        a.dtype, a.N, a.T, a.C, a.heads, a.d, a.dpad, a.k_off = self.dt, N, T, Cc, heads, d, dpad, k_off
        return idx

    def attention_nat(self, qkv, out, N, T, Cc, heads, d, q_off, k_off, v_off, head_stride, lse=None, qkv_bound=None, out_presplit=False,
                      in_presplit=False):
        """fused attention on the natural qkv layout [N][T][3C] (head dim % 8 == 0 and <= 512 -- <= 64 in the exact fp32 mode --, any T): eod_attention_fwd_nat.  fp32 storage
        (fp32x3): qkv_bound = bound table [N][32] of qkv (None: |q|, |k|, |v| < 4094 guaranteed by the caller); out_presplit: `out` is
        written pre-split (scale from qkv_bound) for the proj_out conv"""
        flags = (_lib.ATTN_OUT_PRESPLIT if out_presplit else 0) | (_lib.ATTN_IN_PRESPLIT if in_presplit else 0) | \
            (_lib.ATTN_EXACT_F32 if self.precision == "fp32" else 0)  # (the exact mode: IEEE fp32 products, csrc/attn_f32.hip)
        return self._small(OP_ATTN_NAT, p=(ptr(qkv), ptr(out), ptr(lse), ptr(qkv_bound)), l=(flags,),
                           i=(self.dt, N, T, Cc, heads, d, q_off, k_off, v_off, head_stride))

    def softmax_rows(self, s_f32, lds, p_out, ldp, rows, n):
        return self._small(OP_SOFTMAX, p=(ptr(s_f32), ptr(p_out)), l=(lds, ldp, rows), i=(self.dt, n))

    def to_nhwc(self, N, C0, C1, H, W, c_pad):
        y = self.act(N, H, W, c_pad)
        idx = self._small(OP_TO_NHWC, p=(0, 0, ptr(y.t)), i=(C0, C1, self.dt, N, H, W, c_pad))
        return y, idx

    def to_nchw(self, x):
        idx = self._small(OP_TO_NCHW, p=(ptr(x.t), 0), i=(self.dt, x.N, x.H, x.W, x.C))
        return idx

    def resample2x(self, x, mode, pad_tl=False):
        if mode == 4:  # crop: pad_tl = bit 0 drop the last row, bit 1 drop the last column
            y = self.act(x.N, x.H - (int(pad_tl) & 1), x.W - ((int(pad_tl) >> 1) & 1), x.C)
            self._small(OP_POOL, p=(ptr(x.t), ptr(y.t)), i=(self.dt, x.N, x.H, x.W, x.C, 4, int(pad_tl)))
            return y
        if mode in (1, 3):
            ho, wo = 2 * x.H + int(pad_tl), 2 * x.W + int(pad_tl)
        else:
            o = int(pad_tl) if mode == 2 else 0
            ho, wo = (x.H - o) // 2, (x.W - o) // 2
        y = self.act(x.N, ho, wo, x.C)
        if mode in (0, 1, 3):  # averages and copies stay inside the input's bound
            y.bound = x.bound
        self._small(OP_POOL, p=(ptr(x.t), ptr(y.t)), i=(self.dt, x.N, x.H, x.W, x.C, mode, int(pad_tl)))
        return y

    def dropout(self, x, p):
        """nn.Dropout in train mode: y = x * keep / (1 - p), keep = Philox4x32-10(seed; element / 4, layer, step) (eod_dropout)"""
        y = self.act(x.N, x.H, x.W, x.C)
        layer = len(self.drop_ops)
        idx = self._small(OP_DROPOUT, p=(ptr(x.t), ptr(y.t)), l=(x.t.numel(), self.drop_seed), i=(self.dt, layer, 0), f=(float(p),))
        self.drop_ops.append(idx)
        return y

    def next_dropout_step(self):
        if not self.drop_ops:
            return
        self.drop_step += 1
        if self._arr is None:
            self.finalize()
        for idx in self.drop_ops:
            self._arr[idx].u.small.i[2] = self.drop_step & 0x7fffffff

    def temb(self, desc_fields):
        op, idx = self._push(OP_TEMB)
        for k, v in desc_fields.items():
            setattr(op.u.temb, k, v)
        return idx

    # ------------------------------------------------------------------ execution
    def finalize(self):
        arr = (Op * len(self.ops))()
        for k, o in enumerate(self.ops):
            C.memmove(C.byref(arr, k * C.sizeof(Op)), C.byref(o), C.sizeof(Op))
        self._arr = arr
        return self

    def set_binding(self, name, value_ptr):
        if self._arr is None:
            self.finalize()
        for op_index, setter in self.bindings[name]:
            setter(self._arr[op_index], value_ptr)

    def run(self, stream=None):
        if self._arr is None:
            self.finalize()
        st = stream if stream is not None else current_stream_ptr(self.device)
        if self._timer:
            check(self.L.eod_program_run_timed(self._arr, len(self.ops), st, self._timer), "eod_program_run_timed")
        else:
            check(self.L.eod_program_run(self._arr, len(self.ops), st), "eod_program_run")

    # ------------------------------------------------------------------ measurement (bench.py)
    def enable_timing(self, max_runs, only=None):
        """bracket ops with HIP events on the launch stream for the next `max_runs` runs; `only` = iterable of op
        indices to time (default all; every event pair costs a few microseconds of stream idle time)"""
        self.disable_timing()
        self._timer = self.L.eod_timer_create(len(self.ops), max_runs)
        if not self._timer:
            raise _lib.EodError("eod_timer_create failed")
        if only is not None:
            mask = (C.c_ubyte * len(self.ops))()
            for k in only:
                mask[k] = 1
            check(self.L.eod_timer_set_mask(self._timer, mask, len(self.ops)), "eod_timer_set_mask")

    def read_timing(self):
        """(runs, [ms summed over runs] per op); call after synchronising the stream"""
        buf = (C.c_float * len(self.ops))()
        runs = self.L.eod_timer_read(self._timer, buf)
        if runs < 0:
            check(runs, "eod_timer_read")
        return runs, list(buf)

    def disable_timing(self):
        if getattr(self, "_timer", None):
            self.L.eod_timer_destroy(self._timer)
        self._timer = None

    def op_stats(self):
        """algorithmic work per op: list of dicts(kind, flops, bytes, label) from the descriptors"""
        es = self.tdtype.itemsize
        out = []
        for op in self.ops:
            k = op.kind
            if k == OP_CONV:
                d = op.u.conv
                cin = d.C0 + d.C1
                m = d.N * d.Ho * d.Wo
                cin_alg = getattr(op, "_cin_alg", cin)
                sc = getattr(op, "_skip_c", 0)  # fused 1x1 skip conv over `sc` more input channels (center tap only)
                fl = 2.0 * m * d.Cout * (cin_alg * d.ksize * d.ksize + sc)
                by = es * (d.N * d.H * d.W * (cin + sc) + d.ksize * d.ksize * d.Cout * cin + d.Cout * sc + (0 if d.out_nchw_f32 else m * d.Cout)) \
                    + (4 * m * d.Cout if d.out_nchw_f32 else 0) + (es * m * d.Cout if d.res else 0)
                geo = d.ksize == 3 and d.stride == 1 and d.pad == 1 and not d.pad_tl and (d.Wo % 16 == 0 or (d.Wo == 8 and d.Cout > 64 and not d.upsample)) and d.Ho % 8 == 0
                halo = geo and d.Cout > 64 and not d.out_nchw_f32 and not d.w_tapmajor  # mirrors conv_uses_halo() in csrc/igemm.hip
                head = geo and d.Cout <= 32 and d.out_nchw_f32 and not d.upsample  # 32-column instance (HBM-bound head conv)
                headk = head and bool(d.gn_scale_shift) and d.Cout <= 16 and d.C1 == 0 and cin <= 384 and self.precision != "fp32" \
                    and self.L.eod_get_option(b"head") != 0  # mirrors conv_head_ok()
                first = geo and bool(d.w_tapmajor) and bool(d.w_split) and d.Wo % 16 == 0 and d.Cout > 64 and d.C0 in (4, 8) and not d.out_nchw_f32 \
                    and self.L.eod_get_option(b"first") != 0  # mirrors conv_first_ok()
                up4 = d.upsample == 3  # parity-class form of the nearest-2x conv: the algorithm's 9 taps are executed as 4 (pre-summed)
                out.append(dict(kind="conv", flops=fl, bytes=by, exec_flops=fl * (4.0 / 9.0 if up4 else 1.0),
                                kernel="conv_up4_halo_kernel" if up4 else "conv3x3_halo_kernel" if halo else
                                       "conv_head_kernel" if headk else "conv3x3_halo_kernel<BN=32>" if head else
                                       "conv_first_x3_kernel" if first else "igemm_kernel",
                                label=f"conv{d.ksize}x{d.ksize}s{d.stride}{'u4' if up4 else 'u' if d.upsample else ''} {d.H}x{d.W} {cin}->{d.Cout}"
                                      + (f" +skip1x1 {sc}" if sc else "")))
            elif k == OP_GEMM:
                d = op.u.gemm
                nb = d.nb0 * d.nb1
                out.append(dict(kind="gemm", flops=2.0 * nb * d.M * d.N * d.K,
                                bytes=es * nb * (d.M * d.K + d.N * d.K) + (4 if d.c_f32 else es) * nb * d.M * d.N,
                                label=f"gemm {nb}x[{d.M}x{d.N}x{d.K}]"))
            elif k in (OP_GN_PARTIAL, OP_GN_APPLY):
                s = op.u.small
                n, hw, c = (s.i[1], s.i[2], s.i[3])
                by = es * n * hw * c * (1 if k == OP_GN_PARTIAL else 2)
                out.append(dict(kind="gn_partial" if k == OP_GN_PARTIAL else "gn_apply", flops=0.0, bytes=by,
                                label=f"gn {hw}px {c}ch"))
            elif k == OP_ATTN:
                a = op.u.attn
                out.append(dict(kind="attention", flops=4.0 * a.N * a.T * a.T * a.C,
                                bytes=es * a.N * a.T * a.C * 4, label=f"attn T={a.T} h={a.heads} d={a.d}"))
            elif k == OP_ATTN_NAT:
                s = op.u.small
                n_, t_, c_ = s.i[1], s.i[2], s.i[3]
                out.append(dict(kind="attention", flops=4.0 * n_ * t_ * t_ * c_, bytes=es * n_ * t_ * c_ * 4,
                                label=f"attn(nat) T={t_} h={s.i[4]} d={s.i[5]}"))
            elif k == OP_SOFTMAX:
                s = op.u.small
                out.append(dict(kind="softmax", flops=0.0, bytes=s.l[2] * (4 * s.i[1] + es * s.l[1]), label="softmax"))
            else:
                out.append(dict(kind={OP_GN_FINALIZE: "gn_finalize", OP_TEMB: "temb", OP_TO_NHWC: "to_nhwc",
                                      OP_TO_NCHW: "to_nchw", OP_POOL: "resample", OP_ACT_BOUND: "act_bound", OP_BOUND_AFFINE: "act_bound"}.get(k, str(k)), flops=0.0, bytes=0.0,
                                label=""))
        return out


def round_up(v, m):
    return (v + m - 1) // m * m
