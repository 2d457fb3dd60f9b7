"""MI355X-native guided-diffusion UNet behind the reference's Python API.

Drop-in for furio1999/EO_Diffusion `backbones/unet_openai.py`: same class names, constructor
signatures (`UNetModel(...)` :553-575), `forward(x, timesteps, cond=None, y=None)` (:746) and
`state_dict()` key layout (incl. the unused `nout` / `conv_out` head of :744), so checkpoints and the
reference's train.py / inference.py work unchanged.  Nothing here calls torch arithmetic: modules are
parameter containers + *emitters* that append descriptors to an `engine.Program`; the arithmetic runs
in libeodiff.so (implicit-GEMM MFMA convolutions, fused GroupNorm+SiLU, attention GEMMs, ...).

Precision modes (storage dtype of activations / packed weights; accumulation is always fp32):
  "fp32":   exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), rel-L2 <= 1e-5 vs the fp32 CPU oracle
  "fp32x3": fp32 storage; the 3x3 convs compute each product as three fp16 MFMAs on operands split into hi + lo halves
            (~2^-22 relative per product, same 1e-5 gate as "fp32", ~2.5x faster); everything else as "fp32"
  "fp16":   fp16 storage + v_mfma_f32_32x32x16_f16 (the reference's `use_fp16=True` intent, :568,592), <= 5e-3
Select with UNetModel(..., use_fp16=True), `model.set_precision(...)` or EOD_PRECISION=...
"""
import math
import os
from abc import abstractmethod

import torch
import torch as th
import torch.nn as nn

from .. import _lib
from ..engine import Act, Program, current_stream_ptr, require_gpu, round_up

__all__ = [
    "GroupNorm32", "conv_nd", "linear", "avg_pool_nd", "update_ema", "zero_module", "normalization",
    "timestep_embedding", "checkpoint", "TimestepBlock", "TimestepEmbedSequential", "Upsample", "Downsample",
    "ResBlock", "AttentionBlock", "QKVAttentionLegacy", "QKVAttention", "UNetModel", "UNetBig", "UNet", "UNetSmall",
    "unet_param_shapes", "th", "nn", "math",
]


def _not_on_path(self, *a, **k):
    raise _lib.EodError(
        f"{type(self).__name__} is a parameter container on the HIP path and has no standalone forward; "
        "call the enclosing ResBlock / AttentionBlock / Upsample / Downsample / UNetModel instead.")


class GroupNorm32(nn.GroupNorm):
    """Parameters of GroupNorm(32, C) (unet_openai.py:11-13); computed by eod_gn_* inside blocks."""
    forward = _not_on_path


class _Conv2dP(nn.Conv2d):
    forward = _not_on_path


class _Conv1dP(nn.Conv1d):
    forward = _not_on_path


class _LinearP(nn.Linear):
    forward = _not_on_path


class _SiLUMark(nn.SiLU):
    forward = _not_on_path


class _DropoutMark(nn.Dropout):
    forward = _not_on_path


def conv_nd(dims, *args, **kwargs):
    if dims == 1:
        return _Conv1dP(*args, **kwargs)
    if dims == 2:
        return _Conv2dP(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


def linear(*args, **kwargs):
    return _LinearP(*args, **kwargs)


def avg_pool_nd(dims, *args, **kwargs):
    if dims == 2:
        return _AvgPool2x(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


class _AvgPool2x(nn.AvgPool2d):
    forward = _not_on_path


def update_ema(target_params, source_params, rate=0.99):
    for targ, src in zip(target_params, source_params):
        targ.detach().mul_(rate).add_(src, alpha=1 - rate)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def normalization(channels):
    return GroupNorm32(32, channels)


def timestep_frequencies(dim, max_period=10000):
    """fp32 frequency table of timestep_embedding (unet_openai.py:91-94); host-side, init-time."""
    half = dim // 2
    return th.exp(-math.log(max_period) * th.arange(start=0, end=half, dtype=th.float32) / half)


def timestep_embedding(timesteps, dim, max_period=10000):
    """Sinusoidal embedding [N, dim] of a 1-D tensor of (possibly fractional) timesteps (unet_openai.py:81-99).  Inside
    UNetModel.forward the sinusoid is fused into the first Linear of time_embed (eod_time_embed); this standalone form is one
    launch of eod_timestep_embedding."""
    require_gpu(timesteps, "timestep_embedding")
    t = timesteps.detach().to(th.float32).contiguous()
    assert t.dim() == 1, "timesteps must be a 1-D tensor"
    freqs = timestep_frequencies(dim, max_period).to(t.device)
    out = th.empty((t.shape[0], dim), dtype=th.float32, device=t.device)
    _lib.check(_lib.lib().eod_timestep_embedding(t.data_ptr(), freqs.data_ptr() if dim >= 2 else 0, out.data_ptr(), t.shape[0], dim,
                                                 current_stream_ptr(t.device)), "eod_timestep_embedding")
    return out


def device_timesteps(timesteps, device):
    """(tensor, is_fp32): timesteps as the timestep-MLP kernel reads them -- int64 [N] for every integer dtype (what each caller on the path
    passes: model.py:40, 52; ddim.py:143), fp32 [N] for a floating tensor: the reference's timestep_embedding forms
    `timesteps[:, None].float() * freqs` (unet_openai.py:95) and so also takes fractional values; eod_temb_desc.t_f32 selects the type"""
    if timesteps.is_floating_point():
        return timesteps.to(device=device, dtype=th.float32).contiguous(), 1
    return timesteps.to(device=device, dtype=th.int64).contiguous(), 0


def checkpoint(func, inputs, params, flag):
    """Activation checkpointing is a training-memory device (unet_openai.py:102-148); inference path
    simply evaluates the function."""
    return func(*inputs)


class CheckpointFunction(th.autograd.Function):
    """Name kept for scripts that import it (unet_openai.py:120-148).  The reference re-runs `run_function` in its backward to save
    activation memory; here the training step keeps what its own backward needs and the flash backward recomputes P (training.py), so
    this node only evaluates the function -- under autograd, UNetModel.forward goes through training._UNetTrainFn instead."""

    @staticmethod
    def forward(ctx, run_function, length, *args):
        with th.no_grad():
            return run_function(*args[:length])

    @staticmethod
    def backward(ctx, *output_grads):
        raise NotImplementedError("CheckpointFunction: gradients of the HIP path come from UNetModel's own training step (training.py)")


def count_flops_attn(model, _x, y):
    """`thop` hook (unet_openai.py:436-455): the two batched matmuls of an attention op on an output of shape [b, c, *spatial] cost
    b * T^2 * c multiply-adds each, T = prod(spatial); used as thop.profile(..., custom_ops={QKVAttention: QKVAttention.count_flops})"""
    b, c, *spatial = y[0].shape
    T = 1
    for d in spatial:
        T *= int(d)
    model.total_ops += th.DoubleTensor([2 * b * T * T * c])


# ------------------------------------------------------------------------------------------------
# standalone execution helper: NCHW fp32 in -> program -> NCHW fp32 out
# ------------------------------------------------------------------------------------------------
def default_precision():
    return os.environ.get("EOD_PRECISION", "fp32")


class _Emitter(nn.Module):
    """Mixin: run a single block through the HIP path (used by per-module parity tests)."""

    _precision = None

    def _standalone(self, x, emit, extra_key=()):
        require_gpu(x, type(self).__name__)
        if th.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError(f"{type(self).__name__} called standalone runs inference only: wrap the call in torch.no_grad() "
                                      "(training goes through UNetModel.forward, which has the HIP backward)")
        prec = self._precision or default_precision()
        N, Cc, H, W = x.shape
        prog = Program(x.device, prec)
        c_pad = round_up(Cc, prog.epc)
        a, idx = prog.to_nhwc(N, Cc, 0, H, W, c_pad)
        xin = x.detach().contiguous().float()
        prog.ops[idx].u.small.p[0] = xin.data_ptr()
        if c_pad != Cc:
            raise _lib.EodError(f"{type(self).__name__}: channels ({Cc}) must be a multiple of {prog.epc}")
        y = emit(prog, a)
        out = th.empty((y.N, y.C, y.H, y.W), dtype=th.float32, device=x.device)
        i2 = prog.to_nchw(y)
        prog.ops[i2].u.small.p[1] = out.data_ptr()
        prog.next_dropout_step()
        prog.run()
        th.cuda.current_stream(x.device).synchronize()  # program-owned buffers die with `prog`
        return out


class TimestepBlock(_Emitter):
    @abstractmethod
    def forward(self, x, emb):
        """Apply the module to `x` given `emb` timestep embeddings."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Children that are TimestepBlocks receive the embedding (unet_openai.py:195-208)."""

    def _emit(self, prog, h, ctx):
        for layer in self:
            h = layer._emit(prog, h, ctx)
        return h

    def forward(self, x, emb):
        for layer in self:
            x = layer(x, emb) if isinstance(layer, TimestepBlock) else layer(x)
        return x


class Upsample(_Emitter):
    """nearest 2x (+ optional 3x3 conv), unet_openai.py:211-242.  The 2x is never materialised when a
    conv follows: the implicit-GEMM gather indexes (h>>1, w>>1)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if use_conv:
            self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=1)

    def _emit(self, prog, h, ctx=None):
        if isinstance(h, tuple):
            raise _lib.EodError("Upsample over a virtual concat is not supported")
        assert h.C == self.channels
        pad_tl = h.H == 3 and h.W == 3  # the 3x3 -> 7x7 hack of unet_openai.py:237-239
        if not self.use_conv:
            return prog.resample2x(h, 1, pad_tl)
        if not pad_tl and prog.conv_up4_ok(h, self.out_channels):
            # parity-class form (engine.pack_conv_up4): every output parity is a 2x2-tap conv of the stored map -> 4/9 of the MACs
            y, _ = prog.conv(h, prog.pack_conv_up4(self.conv.weight), prog.f32(self.conv.bias), self.out_channels,
                             ksize=3, stride=1, pad=1, upsample="up4", stats=True)
            return y
        y, _ = prog.conv(h, prog.pack_conv(self.conv.weight), prog.f32(self.conv.bias), self.out_channels,
                         ksize=3, stride=1, pad=1, upsample=True, pad_tl=pad_tl, stats=True)
        return y

    def forward(self, x):
        assert x.shape[1] == self.channels
        return self._standalone(x, self._emit)


class Downsample(_Emitter):
    """stride-2 3x3 conv or 2x2 average pool, unet_openai.py:245-271."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if use_conv:
            self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=1)
        else:
            assert self.channels == self.out_channels
            self.op = avg_pool_nd(dims, kernel_size=2, stride=2)

    def _emit(self, prog, h, ctx=None):
        assert h.C == self.channels
        if not self.use_conv:
            return prog.resample2x(h, 0)
        y, _ = prog.conv(h, prog.pack_conv(self.op.weight), prog.f32(self.op.bias), self.out_channels,
                         ksize=3, stride=2, pad=1, stats=True)
        return y

    def forward(self, x):
        assert x.shape[1] == self.channels
        return self._standalone(x, self._emit)


class _EmbCtx:
    """Where each ResBlock finds its slice of the batched emb_layers GEMV output [N][J]."""

    def __init__(self):
        self.blocks = []   # ResBlocks in emission order
        self.offsets = {}  # id(block) -> column offset
        self.J = 0
        self.out = None    # fp32 tensor [N][J]

    def register(self, blk):
        self.offsets[id(blk)] = self.J
        self.blocks.append(blk)
        self.J += blk.emb_layers[1].out_features


class ResBlock(TimestepBlock):
    """GN+SiLU -> conv3x3 (+timestep bias) -> GN+SiLU -> conv3x3 (+skip), unet_openai.py:274-385."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False,
                 dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        self.channels = channels
        self.emb_channels = emb_channels
        self.dropout = dropout
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.use_checkpoint = use_checkpoint
        self.use_scale_shift_norm = use_scale_shift_norm
        self.in_layers = nn.Sequential(normalization(channels), _SiLUMark(),
                                       conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.updown = up or down
        if up:
            self.h_upd = Upsample(channels, False, dims)
            self.x_upd = Upsample(channels, False, dims)
        elif down:
            self.h_upd = Downsample(channels, False, dims)
            self.x_upd = Downsample(channels, False, dims)
        else:
            self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(
            _SiLUMark(), linear(emb_channels, 2 * self.out_channels if use_scale_shift_norm else self.out_channels))
        self.out_layers = nn.Sequential(
            normalization(self.out_channels), _SiLUMark(), _DropoutMark(p=dropout),
            zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    def _emit(self, prog, h, ctx):
        """h: Act, or (Act, Act) = virtual th.cat([h, skip], 1) of unet_openai.py:773."""
        srcs = list(h) if isinstance(h, tuple) else [h]
        cin = sum(s.C for s in srcs)
        assert cin == self.channels, (cin, self.channels)
        cout = self.out_channels
        gn1, conv1 = self.in_layers[0], self.in_layers[2]
        gn2, conv2 = self.out_layers[0], self.out_layers[3]
        # GroupNorm+SiLU of the block input: statistics here, the normalisation itself is applied inside the conv
        # (fused into the halo-patch staging) whenever the conv kernel can; resblock_updown needs it materialised.
        ss1 = prog.gn_stats(srcs, prog.f32(gn1.weight), prog.f32(gn1.bias), eps=gn1.eps)
        off = ctx.offsets[id(self)]
        emb_ptr_off = ctx.out[:, off:]  # view: pointer to column `off`, row stride J
        temb = {} if self.use_scale_shift_norm else dict(cbias=emb_ptr_off, cbias_stride=ctx.J)
        if self.updown:
            if len(srcs) != 1:
                raise _lib.EodError("resblock_updown over a virtual concat is not supported")
            hn = self.h_upd._emit(prog, prog.gn_apply(srcs, ss1, silu=True))
            srcs = [self.x_upd._emit(prog, srcs[0])]
            h1, _ = prog.conv(hn, prog.pack_conv(conv1.weight), prog.f32(conv1.bias), cout, stats=True, **temb)
        else:
            h1, _ = prog.conv(srcs[0], prog.pack_conv(conv1.weight), prog.f32(conv1.bias), cout,
                              x2=srcs[1] if len(srcs) > 1 else None, gn=(ss1, True), stats=True, **temb)
        if self.use_scale_shift_norm:
            # FiLM (unet_openai.py:377-381): emb_out[:, :cout] = scale, [:, cout:] = shift, folded into the GN table
            ss2 = prog.gn_stats([h1], prog.f32(gn2.weight), prog.f32(gn2.bias), eps=gn2.eps, film=emb_ptr_off, film_stride=ctx.J)
        else:
            ss2 = prog.gn_stats([h1], prog.f32(gn2.weight), prog.f32(gn2.bias), eps=gn2.eps)
        live_dropout = self.training and self.dropout > 0
        if isinstance(self.skip_connection, nn.Identity):
            if len(srcs) != 1:
                raise _lib.EodError("identity skip over a virtual concat is not supported")
            skip = srcs[0]
        elif self.skip_connection.kernel_size[0] == 1 and not live_dropout and prog.conv_skip_ok(h1, cout, srcs):
            # the 1x1 skip_connection rides in conv2's accumulators (its K loop continues over the block input's channels): the skip
            # tensor is never written or read back
            sc = self.skip_connection
            out, _ = prog.conv(h1, prog.pack_conv(conv2.weight), prog.f32(conv2.bias), cout, stats=True, gn=(ss2, True),
                               skip=(srcs, sc.weight, sc.bias))
            return out
        else:
            sc = self.skip_connection
            k = sc.kernel_size[0]
            skip, _ = prog.conv(srcs[0], prog.pack_conv(sc.weight), prog.f32(sc.bias), cout,
                                x2=srcs[1] if len(srcs) > 1 else None, ksize=k, stride=1, pad=k // 2)
        if live_dropout:
            # train-mode forward without autograd (the reference samples previews from modules left in train mode): nn.Dropout of
            # out_layers[2] (unet_openai.py:339) is live.  GroupNorm+SiLU is materialised, then y = x * keep / (1 - p) with the
            # Philox mask of eod_dropout keyed by (seed, layer, forward counter) -- a fresh mask every forward
            hn = prog.gn_apply([h1], ss2, silu=True)
            hd = prog.dropout(hn, self.dropout)
            out, _ = prog.conv(hd, prog.pack_conv(conv2.weight), prog.f32(conv2.bias), cout, res=skip, stats=True)
            return out
        # every block output feeds a GroupNorm next (in_layers / attention norm / out head): emit its partial sums here
        out, _ = prog.conv(h1, prog.pack_conv(conv2.weight), prog.f32(conv2.bias), cout, res=skip, stats=True, gn=(ss2, True))
        return out

    def forward(self, x, emb):
        """Standalone call with an explicit embedding [N, emb_channels] (fp32, on the GPU)."""
        require_gpu(emb, "ResBlock emb")

        def emit(prog, a):
            ctx = _EmbCtx()
            ctx.register(self)
            lin = self.emb_layers[1]
            n, e = emb.shape
            ctx.out = prog.empty((n, ctx.J), th.float32)
            embc = prog.own(emb.detach().contiguous().float())
            # emb_layers = Linear(SiLU(emb)): the last stage of eod_time_embed with `emb` supplied (w1 = NULL)
            prog.temb(dict(t=0, freqs=0, w1=0, b1=0, w2=0, b2=0, label_emb=0, y=0,
                           wcat=_lib.ptr(prog.f32(lin.weight)), bcat=_lib.ptr(prog.f32(lin.bias)), h1=0,
                           emb=_lib.ptr(embc), out=_lib.ptr(ctx.out), N=n, D=0, E=e, J=ctx.J))
            return self._emit(prog, a, ctx)

        return self._standalone(x, emit)


def _qkv_attention_standalone(mod, qkv):
    """QKVAttention(Legacy).forward(qkv) (unet_openai.py:465-481 / 497-515) as its own call: qkv [N, 3*H*d, T] -> [N, H*d, T].
    One layout pass to the kernels' [N][T][3C] form, the fused attention kernel (eod_attention_fwd_nat; the T x T weights never exist),
    one pass back.  Precision = EOD_PRECISION (a bare QKVAttention module has no enclosing model to ask): fp16 -> fp16 storage and
    MFMA; fp32x3 -> fp32 in and out, fp32 softmax, both contractions as split-fp16 products (~2^-22 per product, safe at any magnitude:
    the operand scale comes from a max|x| pass over qkv); fp32 -> the same with exact fp32 MFMA products (csrc/attn_f32.hip)."""
    require_gpu(qkv, type(mod).__name__)
    bs, width, length = qkv.shape
    nh = mod.n_heads
    assert width % (3 * nh) == 0
    d = width // (3 * nh)
    Cc = nh * d
    if d % 8 or d > (64 if default_precision() == "fp32" else 512):
        raise _lib.EodError(f"{type(mod).__name__} standalone: head dim {d} must be a multiple of 8 and <= 512 (<= 64 in the exact fp32 "
                            "mode; inside AttentionBlock the other head sizes run through the GEMM path)")
    prog = Program(qkv.device, default_precision())
    a0, i_in = prog.to_nhwc(bs, width, 0, 1, length, width)
    xin = qkv.detach().contiguous().float()
    prog.ops[i_in].u.small.p[0] = xin.data_ptr()
    qo, ko, vo, hs = (0, Cc, 2 * Cc, d) if mod.new_order else (0, d, 2 * d, 3 * d)
    a = prog.act(bs, 1, length, Cc)
    prog.attention_nat(a0.t, a.t, bs, length, Cc, nh, d, qo, ko, vo, hs, qkv_bound=prog.bound_of([a0]))
    out = th.empty((bs, Cc, 1, length), dtype=th.float32, device=qkv.device)
    i_out = prog.to_nchw(a)
    prog.ops[i_out].u.small.p[1] = out.data_ptr()
    prog.run()
    th.cuda.current_stream(qkv.device).synchronize()  # program-owned buffers die with `prog`
    return out.view(bs, Cc, length).type(qkv.dtype)


class QKVAttentionLegacy(nn.Module):
    """Head-interleaved qkv layout [h][q|k|v][d] (unet_openai.py:456-481)."""
    new_order = False

    def __init__(self, n_heads):
        super().__init__()
        self.n_heads = n_heads

    def forward(self, qkv):
        return _qkv_attention_standalone(self, qkv)

    @staticmethod
    def count_flops(model, _x, y):
        return count_flops_attn(model, _x, y)


class QKVAttention(nn.Module):
    """qkv layout [q|k|v][h][d] (unet_openai.py:488-515)."""
    new_order = True

    def __init__(self, n_heads):
        super().__init__()
        self.n_heads = n_heads

    def forward(self, qkv):
        return _qkv_attention_standalone(self, qkv)

    @staticmethod
    def count_flops(model, _x, y):
        return count_flops_attn(model, _x, y)


class AttentionPool2d(nn.Module):
    """CLIP-style attention pooling (unet_openai.py:151-181): defined upstream, instantiated by nothing on the EODiffusion path (only
    backbones/unet.py's EncoderUNetModel uses its twin).  Kept as a parameter container with the reference's constructor and state_dict
    keys (`positional_embedding`, `qkv_proj.*`, `c_proj.*`) so that the name imports and checkpoints holding one load; calling it
    raises -- there is no HIP path for dead code."""

    def __init__(self, spacial_dim: int, embed_dim: int, num_heads_channels: int, output_dim: int = None):
        super().__init__()
        self.positional_embedding = nn.Parameter(th.randn(embed_dim, spacial_dim ** 2 + 1) / embed_dim ** 0.5)
        self.qkv_proj = conv_nd(1, embed_dim, 3 * embed_dim, 1)
        self.c_proj = conv_nd(1, embed_dim, output_dim or embed_dim, 1)
        self.num_heads = embed_dim // num_heads_channels
        self.attention = QKVAttention(self.num_heads)

    def forward(self, x):
        raise NotImplementedError("AttentionPool2d is not on the EODiffusion path (SURVEY.md section 8: out of scope); no HIP kernel path exists for it")


class AttentionBlock(_Emitter):
    """x + proj_out(attention(qkv(GN(x)))) on [N, C, T], unet_openai.py:388-433."""

    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False, use_new_attention_order=False):
        super().__init__()
        self.channels = channels
        if num_head_channels == -1:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0, \
                f"q,k,v channels {channels} is not divisible by num_head_channels {num_head_channels}"
            self.num_heads = channels // num_head_channels
        self.use_checkpoint = use_checkpoint
        self.norm = normalization(channels)
        self.qkv = conv_nd(1, channels, channels * 3, 1)
        self.attention = QKVAttention(self.num_heads) if use_new_attention_order else QKVAttentionLegacy(self.num_heads)
        self.proj_out = zero_module(conv_nd(1, channels, channels, 1))

    def _row_maps(self, epc):
        """rows of qkv.weight producing q|k (layout [q: heads x dpad | k: heads x dpad], each head zero-padded
        to a multiple of one 16-byte chunk so that it can be the K dimension of the score GEMM) and v (head-major)."""
        Cc, nh = self.channels, self.num_heads
        assert (3 * Cc) % (3 * nh) == 0
        d = Cc // nh
        dpad = round_up(d, epc)
        if self.attention.new_order:   # [q|k|v][h][d]  (unet_openai.py:506-514)
            base = lambda which, h: which * Cc + h * d
        else:                          # [h][q|k|v][d]  (unet_openai.py:474)
            base = lambda which, h: h * 3 * d + which * d
        qk = []
        for which in (0, 1):
            for h in range(nh):
                qk += [base(which, h) + j for j in range(d)] + [-1] * (dpad - d)
        v = [base(2, h) + j for h in range(nh) for j in range(d)]
        return qk, v, d, dpad

    def _emit(self, prog, x, ctx=None):
        if isinstance(x, tuple):
            raise _lib.EodError("AttentionBlock over a virtual concat is not supported")
        Cc, nh = self.channels, self.num_heads
        assert x.C == Cc
        N, T = x.N, x.HW
        d_nat = Cc // nh
        # wide heads (the ONE 512-channel head of the train.py:50 middle block, :675-681): the same natural-layout call, served by
        # csrc/attn_wide.hip (fp16 and split-fp16 products; the exact fp32 mode keeps the materialised path below) -- for SHORT
        # sequences, where the five launches of the materialised path are the cost (T = 64 / 256: 0.048 / 0.068 ms against 0.077 /
        # 0.087 in fp16, 0.053 against 0.108 ms at T = 256 in fp32x3).  Its workgroups own 32 queries, so K and V are re-staged T / 32
        # times: at T = 1024 (256 x 256 inputs) the materialised path is the faster one (0.26 against 0.35 ms) and stays.  The choice
        # depends on the sequence length only (never on the batch: a sample's bits do not depend on its neighbours).
        wide = 64 < d_nat <= 512 and prog.precision != "fp32" and T <= 512
        if d_nat % 8 == 0 and (d_nat <= 64 or wide) and os.environ.get("EOD_ATTN", "nat") == "nat":
            # fused attention straight on the qkv projection's natural channel layout (legacy [h][q|k|v][d], new [q|k|v][h][d]):
            # one projection GEMM, no packed q|k / transposed v operands, T x T never materialised in ANY precision mode
            # (eod_attention_fwd_nat; fp16: fp16 MFMA; fp32x3: fp32 in / out with split-fp16 products, the projections run as 1x1 convs
            # of the same product type; fp32: exact fp32 MFMA, csrc/attn_f32.hip)
            qo, ko, vo, hs = (0, Cc, 2 * Cc, d_nat) if self.attention.new_order else (0, d_nat, 2 * d_nat, 3 * d_nat)
            xn = prog.group_norm([x], prog.f32(self.norm.weight), prog.f32(self.norm.bias), silu=False, eps=self.norm.eps,
                                 split_out=prog.split)  # (fp32x3: written pre-split for the qkv conv, its only consumer)
            if prog.split:
                # The whole block runs on PRE-SPLIT operands where the geometry allows (whole 128-row conv tiles per image): norm writes
                # xn pre-split; the qkv conv's epilogue writes q / k / v pre-split, scaled from an A-PRIORI bound of its output
                # (|W xn + b| <= max row L1 norm * max|xn| + max|b|: prog.linear_bound); the attention kernel then loads Q fragments and
                # LDS-DMAs K / V tiles as they are -- no split arithmetic -- and writes a pre-split for proj_out (rows of a are convex
                # combinations of v rows: the table of qkv bounds them).  Otherwise the table of qkv comes from the conv epilogue's sums
                # of squares (no pass over qkv) and the kernels split in LDS / registers.
                bq = prog.f32(self.qkv.bias)
                ps = xn.presplit and T % 128 == 0 and T >= 256 and Cc % 8 == 0 and not wide  # (whole conv tiles per image, no split-K shapes)
                out_ps = Cc % 8 == 0 and not wide
                qkv_bound = prog.linear_bound(xn.bound, self.qkv.weight.view(3 * Cc, Cc), bq) if ps else None
                qkv, _ = prog.conv(xn, prog.pack_conv(self.qkv.weight.view(3 * Cc, Cc, 1, 1)), bq, 3 * Cc, ksize=1, stride=1, pad=0,
                                   stats=not ps, y_presplit_bound=qkv_bound)
                a = prog.act(N, x.H, x.W, Cc)
                if not ps:
                    qkv_bound = prog.bound_of([qkv])
                prog.attention_nat(qkv.t, a.t, N, T, Cc, nh, d_nat, qo, ko, vo, hs, qkv_bound=qkv_bound, out_presplit=out_ps, in_presplit=ps)
                a.bound, a.presplit = qkv_bound, out_ps
                out, _ = prog.conv(a, prog.pack_conv(self.proj_out.weight.view(Cc, Cc, 1, 1)), prog.f32(self.proj_out.bias), Cc,
                                   ksize=1, stride=1, pad=0, res=x, stats=True)
                return out
            wqkv = prog.pack_rows(self.qkv.weight.view(3 * Cc, Cc))
            qkv = prog.empty((N * T, 3 * Cc))
            prog.gemm(xn.t, wqkv, qkv, N * T, 3 * Cc, Cc, Cc, Cc, 3 * Cc, bias=prog.f32(self.qkv.bias), bias_mode=1)
            a = prog.empty((N * T, Cc))
            prog.attention_nat(qkv, a, N, T, Cc, nh, d_nat, qo, ko, vo, hs)
            out = prog.act(N, x.H, x.W, Cc)
            prog.gemm(a, prog.pack_rows(self.proj_out.weight.view(Cc, Cc)), out.t, N * T, Cc, Cc, Cc, Cc, Cc, bias=prog.f32(self.proj_out.bias),
                      bias_mode=1, res=x.t)
            return out
        qk_rows, v_rows, d, dpad = self._row_maps(prog.epc)
        Cq = nh * dpad
        w2d = self.qkv.weight.view(3 * Cc, Cc)
        wqk = prog.pack_rows(w2d, qk_rows)
        wv = prog.pack_rows(w2d, v_rows)
        bq = prog.f32(self.qkv.bias)
        bq0 = th.cat([bq, bq.new_zeros(1)])  # index -1 -> 0 bias for the padding rows
        bqk = prog.own(bq0[th.tensor(qk_rows, device=bq.device)].contiguous())
        bv = prog.own(bq[th.tensor(v_rows, device=bq.device)].contiguous())
        wproj = prog.pack_rows(self.proj_out.weight.view(Cc, Cc))
        bproj = prog.f32(self.proj_out.bias)

        xn = prog.group_norm([x], prog.f32(self.norm.weight), prog.f32(self.norm.bias), silu=False, eps=self.norm.eps)
        # fp32x3 (wide or odd head dims): the split-fp16 GEMMs scale each operand by ITS image's bound table (engine.gemm), so the
        # projections whose rows stack every image along M run as 1x1 convs (per-image scale inside the conv kernel), and the batched
        # GEMMs take the tables of qk (from the conv epilogue's statistics), xn (the GroupNorm finalize), vT (one max|x| pass).
        x3 = prog.split and Cc % 8 == 0
        qk_bound = vT_bound = None
        # q|k projection: [N*T][2*Cq]
        if x3:
            w0 = th.cat([w2d.detach(), w2d.new_zeros(1, Cc)])  # index -1 -> a zero row (head padding), like bq0
            w_qk = prog.own(w0[th.tensor(qk_rows, device=w0.device)].contiguous().view(2 * Cq, Cc, 1, 1))
            qk_act, _ = prog.conv(xn, prog.pack_conv(w_qk), bqk, 2 * Cq, ksize=1, stride=1, pad=0, stats=True)
            qk = qk_act.t.view(N * T, 2 * Cq)
            qk_bound = prog.bound_of([qk_act])
        else:
            qk = prog.empty((N * T, 2 * Cq))
            prog.gemm(xn.t, wqk, qk, N * T, 2 * Cq, Cc, Cc, Cc, 2 * Cq, bias=bqk, bias_mode=1)
        # v projection, produced TRANSPOSED ([N][C][ldt], keys contiguous) by swapping the GEMM operands:
        # vT[n][c][t] = sum_k Wv[c][k] * xn[n][t][k] + bv[c]
        ldt = round_up(T, prog.epc)
        vT = prog.empty((N, Cc, ldt), zero=True)
        prog.gemm(wv, xn.t, vT, Cc, T, Cc, Cc, Cc, ldt, bias=bv, bias_mode=2, nb0=N, sa=(0, 0), sb=(T * Cc, 0),
                  sc=(Cc * ldt, 0), a_bound=prog.weight_bound(wv, N) if x3 else None, b_bound=xn.bound if x3 else None)
        if x3:
            vT_bound = prog.bound_of_tensor(vT, N)
        a = prog.empty((N * T, Cc))
        fused = prog.precision == "fp16" and dpad <= 64 and d % 4 == 0 and os.environ.get("EOD_ATTN", "nat") != "gemm"
        if fused:
            # flash-style fused kernel: online softmax, the T x T matrix never exists
            prog.attention(qk, vT, a, N, T, Cc, nh, d, dpad, 2 * Cq, ldt, Cq)
        else:
            # materialised path (fp32 parity mode, head dims > 64): S = (q*s)(k*s)^T with s = d^-1/4 (unet_openai.py:475-478)
            # -> alpha = 1/sqrt(d); fp32 scores [N*nh][T][ldt], row softmax, then a = P.V
            S = prog.empty((N * nh, T, ldt), th.float32)
            prog.gemm(qk, qk, S, T, T, dpad, 2 * Cq, 2 * Cq, ldt, alpha=1.0 / math.sqrt(d), c_f32=True, nb0=N, nb1=nh,
                      sa=(T * 2 * Cq, dpad), sb=(T * 2 * Cq, dpad), sc=(nh * T * ldt, T * ldt), b_off=Cq, a_bound=qk_bound, b_bound=qk_bound)
            P = prog.empty((N * nh, T, ldt))
            prog.softmax_rows(S, ldt, P, ldt, N * nh * T, T)
            # a[n][t][h*d + j] = sum_s P[n,h][t][s] * vT[n][h*d + j][s]     (P in [0, 1]: no table needed)
            prog.gemm(P, vT, a, T, d, ldt, ldt, ldt, Cc, nb0=N, nb1=nh, sa=(nh * T * ldt, T * ldt),
                      sb=(Cc * ldt, d * ldt), sc=(T * Cc, d), a_bound=None, b_bound=vT_bound)
        if x3:
            a_act = Act(a.view(N, x.H, x.W, Cc), N, x.H, x.W, Cc)
            a_act.bound = vT_bound  # rows of a = convex combinations of v rows
            out, _ = prog.conv(a_act, prog.pack_conv(self.proj_out.weight.view(Cc, Cc, 1, 1)), bproj, Cc, ksize=1, stride=1, pad=0, res=x,
                               stats=True)
            return out
        out = prog.act(N, x.H, x.W, Cc)
        prog.gemm(a, wproj, out.t, N * T, Cc, Cc, Cc, Cc, Cc, bias=bproj, bias_mode=1, res=x.t)
        return out

    def forward(self, x):
        return self._standalone(x, self._emit)


# ------------------------------------------------------------------------------------------------
class UNetModel(_Emitter):
    """The full UNet (unet_openai.py:522-780).  Module graph and parameter names reproduce :597-744
    key for key; `forward` builds (once per input signature) and replays a native launch program."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 time_emb_factor=4, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False):
        super().__init__()
        if dims != 2:
            raise ValueError(f"unsupported dimensions: {dims}")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.dtype = th.float16 if use_fp16 else th.float32
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads_upsample
        self._precision = "fp16" if use_fp16 else None
        self._use_graph = None  # None: follow EOD_GRAPH (default off)

        ted = model_channels * time_emb_factor
        self.time_embed = nn.Sequential(linear(model_channels, ted), _SiLUMark(), linear(ted, ted))
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, ted)

        def res(cin, cout, **kw):
            return ResBlock(cin, ted, dropout, out_channels=cout, dims=dims, use_checkpoint=use_checkpoint,
                            use_scale_shift_norm=use_scale_shift_norm, **kw)

        def attn(c, heads):
            return AttentionBlock(c, use_checkpoint=use_checkpoint, num_heads=heads, num_head_channels=num_head_channels,
                                  use_new_attention_order=use_new_attention_order)

        ch = input_ch = int(channel_mult[0] * model_channels)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, ch, 3, padding=1))])
        self._feature_size = ch
        skip_chans = [ch]
        ds = 1
        last = len(channel_mult) - 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                stage = [res(ch, int(mult * model_channels))]
                ch = int(mult * model_channels)
                if ds in attention_resolutions:
                    stage.append(attn(ch, num_heads))
                self.input_blocks.append(TimestepEmbedSequential(*stage))
                self._feature_size += ch
                skip_chans.append(ch)
            if level != last:
                down = res(ch, ch, down=True) if resblock_updown else Downsample(ch, conv_resample, dims=dims, out_channels=ch)
                self.input_blocks.append(TimestepEmbedSequential(down))
                skip_chans.append(ch)
                ds *= 2
                self._feature_size += ch

        self.middle_block = TimestepEmbedSequential(res(ch, ch), attn(ch, num_heads), res(ch, ch))
        self._feature_size += ch

        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = skip_chans.pop()
                stage = [res(ch + ich, int(model_channels * mult))]
                ch = int(model_channels * mult)
                if ds in attention_resolutions:
                    stage.append(attn(ch, num_heads_upsample))
                if level and i == num_res_blocks:
                    stage.append(res(ch, ch, up=True) if resblock_updown
                                 else Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*stage))
                self._feature_size += ch

        self.out = nn.Sequential(normalization(ch), _SiLUMark(),
                                 zero_module(conv_nd(dims, input_ch, out_channels, 3, padding=1)))
        # the reference also registers an unused duplicate head (unet_openai.py:744); its keys are part
        # of every checkpoint, so they are kept
        self.nout, self.act, self.conv_out = (normalization(ch), _SiLUMark(),
                                              zero_module(conv_nd(dims, input_ch, out_channels, 3, padding=1)))

    # ---- program cache is process-local state, never copied / pickled / saved -------------------
    def __getstate__(self):
        st = dict(self.__dict__)
        st.pop("_eod_cache", None)
        st.pop("_eod_trainers", None)
        return st

    def enable_graph(self, on=True):
        """Replay the UNet launch program as ONE hipGraph launch (HIP stream capture of eod_program_run).  Inputs are
        copied into static buffers and the returned tensor is the program's static output buffer: it is overwritten by
        the next forward (the samplers consume it immediately).  Pays off when the step is launch-bound (small maps)."""
        self._use_graph = bool(on)
        self.__dict__.pop("_eod_cache", None)
        return self

    def set_precision(self, precision):
        if precision not in ("fp32", "fp16", "fp32x3"):
            raise ValueError(precision)
        self._precision = precision
        self.__dict__.pop("_eod_cache", None)
        return self

    @property
    def precision(self):
        return self._precision or default_precision()

    def _fingerprint(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    # ---- emission ---------------------------------------------------------------------------------
    def _build(self, N, cx, ccond, H, W, device, with_y):
        prog = Program(device, self.precision)
        if cx + ccond != self.in_channels:
            raise _lib.EodError(f"UNetModel: got {cx}+{ccond} input channels, model has in_channels={self.in_channels}")
        c_pad = round_up(cx + ccond, prog.epc)
        a0, i_in = prog.to_nhwc(N, cx, ccond, H, W, c_pad)
        prog.bind("x", i_in, lambda op, v: op.u.small.p.__setitem__(0, v))
        if ccond:
            prog.bind("cond", i_in, lambda op, v: op.u.small.p.__setitem__(1, v))

        # ---- timestep embedding: one descriptor for the MLP and ALL ResBlock emb_layers (k6) ----
        ctx = _EmbCtx()
        for m in self.modules():
            if isinstance(m, ResBlock):
                ctx.register(m)
        te1, te2 = self.time_embed[0], self.time_embed[2]
        E, D = te1.out_features, te1.in_features
        wcat = prog.own(th.cat([prog.f32(b.emb_layers[1].weight) for b in ctx.blocks], 0).contiguous())
        bcat = prog.own(th.cat([prog.f32(b.emb_layers[1].bias) for b in ctx.blocks], 0).contiguous())
        freqs = prog.own(timestep_frequencies(D).to(device))
        ctx.out = prog.empty((N, ctx.J), th.float32)
        h1 = prog.empty((N, E), th.float32)
        emb = prog.empty((N, E), th.float32)
        i_t = prog.temb(dict(
            t=0, freqs=_lib.ptr(freqs), w1=_lib.ptr(prog.f32(te1.weight)), b1=_lib.ptr(prog.f32(te1.bias)),
            w2=_lib.ptr(prog.f32(te2.weight)), b2=_lib.ptr(prog.f32(te2.bias)),
            label_emb=_lib.ptr(prog.f32(self.label_emb.weight)) if with_y else 0, y=0,
            wcat=_lib.ptr(wcat), bcat=_lib.ptr(bcat), h1=_lib.ptr(h1), emb=_lib.ptr(emb), out=_lib.ptr(ctx.out),
            N=N, D=D, E=E, J=ctx.J))
        prog.bind("t", i_t, lambda op, v: setattr(op.u.temb, "t", v))
        prog.bind("t_f32", i_t, lambda op, v: setattr(op.u.temb, "t_f32", v))
        if with_y:
            prog.bind("y", i_t, lambda op, v: setattr(op.u.temb, "y", v))

        # ---- encoder ----
        hs = []
        conv0 = self.input_blocks[0][0]
        epc = 16 // prog.tdtype.itemsize
        if c_pad // epc in (1, 2, 4) and c_pad % epc == 0:
            # thin input: K runs over the flattened [tap][channel] axis (2-5 K-steps instead of 9 mostly-zero ones)
            h, i0 = prog.conv(a0, prog.pack_conv_tapmajor(conv0.weight, c_pad), prog.f32(conv0.bias), conv0.out_channels,
                              stats=True, w_tapmajor=True)
        else:
            h, i0 = prog.conv(a0, prog.pack_conv(conv0.weight, cin_pad=c_pad), prog.f32(conv0.bias), conv0.out_channels,
                              stats=True)
        prog.ops[i0]._cin_alg = cx + ccond  # algorithmic K excludes the zero padding (bench accounting only)
        hs.append(h)
        for blk in list(self.input_blocks)[1:]:
            h = blk._emit(prog, h, ctx)
            hs.append(h)
        h = self.middle_block._emit(prog, h, ctx)
        for blk in self.output_blocks:
            h = blk._emit(prog, (h, hs.pop()), ctx)
        # ---- head ----
        gn, conv = self.out[0], self.out[2]
        ss = prog.gn_stats([h], prog.f32(gn.weight), prog.f32(gn.bias), eps=gn.eps)
        _, i_out = prog.conv(h, prog.pack_conv(conv.weight), prog.f32(conv.bias), self.out_channels, out_nchw_f32=True,
                             gn=(ss, True))
        prog.bind("out", i_out, lambda op, v: setattr(op.u.conv, "y", v))
        prog.finalize()
        prog.out_shape = (N, self.out_channels, H, W)
        return prog

    def program_for(self, N, cx, ccond, H, W, device, with_y):
        cache = self.__dict__.setdefault("_eod_cache", {})
        key = (N, cx, ccond, H, W, str(device), with_y, self.precision, self.training)
        fp = self._fingerprint()
        hit = cache.get(key)
        if hit is None or hit[1] != fp:
            cache.clear()  # one live plan per model keeps the HBM footprint bounded
            hit = (self._build(N, cx, ccond, H, W, device, with_y), fp)
            cache[key] = hit
        return hit[0]

    def forward(self, x, timesteps, cond=None, y=None):
        """x [N,C,H,W] fp32, timesteps [N] int, cond [N,Cc,H,W] (channel-concatenated, :754-756), y [N] labels."""
        require_gpu(x, "UNetModel.forward")
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        if th.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training (train.py:116-118): forward that keeps the activations + HIP backward behind torch.autograd
            from ..training import unet_train_forward
            return unet_train_forward(self, x, timesteps, cond, y)
        N, cx, H, W = x.shape
        ccond = 0
        if cond is not None:
            cond = cond.to(x.device)
            ccond = cond.shape[1]
        if y is not None:
            assert y.shape == (N,), (y.shape, x.shape)
        prog = self.program_for(N, cx, ccond, H, W, x.device, y is not None)
        use_graph = self._use_graph if getattr(self, "_use_graph", None) is not None else os.environ.get("EOD_GRAPH", "0") == "1"
        if use_graph and not prog.drop_ops and not timesteps.is_floating_point():
            # (a captured graph would replay ONE dropout mask: train-mode dropout runs un-captured; so do fractional timesteps -- the
            #  captured program reads the int64 slot)
            return self._forward_graph(prog, x, timesteps, cond, y)
        xin = x if (x.dtype == th.float32 and x.is_contiguous()) else x.float().contiguous()
        t64, t_f32 = device_timesteps(timesteps, x.device)
        assert t64.shape == (N,)
        out = th.empty(prog.out_shape, dtype=th.float32, device=x.device)
        prog.set_binding("x", xin.data_ptr())
        if ccond:
            cin = cond if (cond.dtype == th.float32 and cond.is_contiguous()) else cond.float().contiguous()
            prog.set_binding("cond", cin.data_ptr())
        if y is not None:
            y64 = y.to(device=x.device, dtype=th.int64).contiguous()
            prog.set_binding("y", y64.data_ptr())
        prog.set_binding("t", t64.data_ptr())
        prog.set_binding("t_f32", t_f32)
        prog.set_binding("out", out.data_ptr())
        prog.next_dropout_step()
        prog.run()
        # xin/t64/cin/y64 may be temporaries: the caching allocator is stream-ordered on this stream,
        # so reuse after this frame is ordered behind the kernels that read them.
        return out.type(x.dtype) if x.dtype != th.float32 else out


def _forward_graph(self, prog, x, timesteps, cond, y):
    """hipGraph replay path of UNetModel.forward (see enable_graph)."""
    st = getattr(prog, "_graph_state", None)
    if st is None:
        dev = x.device
        st = {"x": th.empty(tuple(x.shape), dtype=th.float32, device=dev),
              "t": th.zeros((x.shape[0],), dtype=th.int64, device=dev),
              "out": th.empty(prog.out_shape, dtype=th.float32, device=dev)}
        prog.set_binding("x", st["x"].data_ptr())
        prog.set_binding("t", st["t"].data_ptr())
        prog.set_binding("t_f32", 0)  # the captured program reads the int64 slot (an earlier fractional-timestep call may have left 1 here)
        prog.set_binding("out", st["out"].data_ptr())
        if cond is not None:
            st["cond"] = th.empty(tuple(cond.shape), dtype=th.float32, device=dev)
            prog.set_binding("cond", st["cond"].data_ptr())
        if y is not None:
            st["y"] = th.zeros((x.shape[0],), dtype=th.int64, device=dev)
            prog.set_binding("y", st["y"].data_ptr())
        st["x"].copy_(x)
        st["t"].copy_(timesteps)
        if cond is not None:
            st["cond"].copy_(cond)
        if y is not None:
            st["y"].copy_(y)
        prog.run()                      # warm-up outside capture (one-off kernel attribute calls)
        th.cuda.current_stream(dev).synchronize()
        g = th.cuda.CUDAGraph()
        with th.cuda.graph(g):
            prog.run()                  # eod_program_run enqueues on the capture stream; nothing in it syncs or allocates
        st["graph"] = g
        prog._graph_state = st
    st["x"].copy_(x)
    st["t"].copy_(timesteps)
    if cond is not None:
        st["cond"].copy_(cond)
    if y is not None:
        st["y"].copy_(y)
    st["graph"].replay()
    return st["out"]


UNetModel._forward_graph = _forward_graph


def unet_param_shapes(**cfg):
    """{state_dict key: shape} of UNetModel(**cfg), computed on the meta device (no allocation)."""
    with th.device("meta"):
        m = UNetModel(**cfg)
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def _factory(image_size, in_channels, out_channels, base_width, num_classes, nrb, head_ch, tef=4):
    """UNetBig / UNet / UNetSmall presets (unet_openai.py:783-922)."""
    mults = {128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4), 32: (1, 2, 2, 2), 28: (1, 2, 2, 2)}
    if image_size not in mults:
        raise ValueError(f"unsupported image size: {image_size}")
    res = "28,14,7" if image_size == 28 else "32,16,8"
    attention_ds = tuple(image_size // int(r) for r in res.split(","))
    return UNetModel(image_size=image_size, in_channels=in_channels, model_channels=base_width,
                     out_channels=out_channels, num_res_blocks=nrb, attention_resolutions=attention_ds,
                     time_emb_factor=tef, dropout=0.1, channel_mult=mults[image_size], num_classes=num_classes,
                     use_checkpoint=False, use_fp16=False, num_heads=4, num_head_channels=head_ch,
                     num_heads_upsample=-1, use_scale_shift_norm=True, resblock_updown=True,
                     use_new_attention_order=True)


def UNetBig(image_size, in_channels=3, out_channels=3, base_width=192, num_classes=None):
    return _factory(image_size, in_channels, out_channels, base_width, num_classes, 3, 64)


def UNet(image_size, in_channels=3, out_channels=3, base_width=64, num_classes=None):
    return _factory(image_size, in_channels, out_channels, base_width, num_classes, 3, 64)


def UNetSmall(image_size, in_channels=3, out_channels=3, base_width=32, num_classes=None):
    return _factory(image_size, in_channels, out_channels, base_width, num_classes, 2, 32, tef=2)
