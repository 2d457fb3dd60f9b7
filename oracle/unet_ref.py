"""Oracle: guided-diffusion UNet forward on the CPU (test infrastructure -- see oracle/__init__.py).

A functional restatement (state_dict in, tensor out) of backbones/unet_openai.py written on
stock torch CPU ops.  It walks the checkpoint key layout directly, so it also documents the
key contract the product modules must reproduce.  Reference lines each function follows:

  timestep_embedding          unet_openai.py:81-99
  group_norm32                unet_openai.py:11-13, 71-78      (fp32 GroupNorm(32, C), eps 1e-5)
  res_block                   unet_openai.py:365-385           (+ FiLM :377-381, up/down :366-371)
  attention_block             unet_openai.py:427-433
  qkv_attention_legacy/new    unet_openai.py:465-481 / 497-515
  upsample / downsample       unet_openai.py:229-242 / 269-271
  unet_forward                unet_openai.py:746-780 with the module graph of :597-744
"""
import math

import torch
import torch.nn.functional as F


def timestep_embedding(timesteps, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def group_norm32(x, w, b):
    return F.group_norm(x.float(), 32, w, b, eps=1e-5).type(x.dtype)


def _conv(sd, key, x, stride=1, padding=0):
    w = sd[key + ".weight"]
    if w.dim() == 3:  # Conv1d 1x1 on [N,C,T]
        return F.conv1d(x, w, sd[key + ".bias"])
    return F.conv2d(x, w, sd[key + ".bias"], stride=stride, padding=padding)


def upsample(sd, pfx, x, use_conv=True):
    out = F.interpolate(x, scale_factor=2, mode="nearest")
    if x.shape[-1] == x.shape[-2] == 3:
        out = F.pad(out, (1, 0, 1, 0))
    if use_conv:
        out = _conv(sd, pfx + ".conv", out, padding=1)
    return out


def downsample(sd, pfx, x, use_conv=True):
    if use_conv:
        return _conv(sd, pfx + ".op", x, stride=2, padding=1)
    return F.avg_pool2d(x, 2, 2)


def res_block(sd, pfx, x, emb, *, film=False, up=False, down=False, drop=None):
    """drop: optional callable pfx -> multiplicative mask (already divided by 1 - p) applied where the reference's nn.Dropout
    sits (out_layers[2], unet_openai.py:339); None = eval mode / p = 0 (identity)"""
    h = F.silu(group_norm32(x, sd[pfx + ".in_layers.0.weight"], sd[pfx + ".in_layers.0.bias"]))
    if up:
        h = upsample(sd, "", h, use_conv=False)
        x = upsample(sd, "", x, use_conv=False)
    elif down:
        h = downsample(sd, "", h, use_conv=False)
        x = downsample(sd, "", x, use_conv=False)
    h = _conv(sd, pfx + ".in_layers.2", h, padding=1)
    e = F.linear(F.silu(emb), sd[pfx + ".emb_layers.1.weight"], sd[pfx + ".emb_layers.1.bias"])
    e = e[:, :, None, None]
    gw, gb = sd[pfx + ".out_layers.0.weight"], sd[pfx + ".out_layers.0.bias"]
    if film:
        scale, shift = torch.chunk(e, 2, dim=1)
        h = group_norm32(h, gw, gb) * (1 + scale) + shift
        h = F.silu(h)
    else:
        h = h + e
        h = F.silu(group_norm32(h, gw, gb))
    if drop is not None:
        h = h * drop(pfx, h)
    h = _conv(sd, pfx + ".out_layers.3", h, padding=1)  # (dropout is the identity in eval mode)
    if (pfx + ".skip_connection.weight") in sd:
        w = sd[pfx + ".skip_connection.weight"]
        x = F.conv2d(x, w, sd[pfx + ".skip_connection.bias"], padding=w.shape[-1] // 2)
    return x + h


def qkv_attention_legacy(qkv, n_heads):
    bs, width, length = qkv.shape
    assert width % (3 * n_heads) == 0
    ch = width // (3 * n_heads)
    q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    a = torch.einsum("bts,bcs->bct", w, v)
    return a.reshape(bs, -1, length)


def qkv_attention_new(qkv, n_heads):
    bs, width, length = qkv.shape
    assert width % (3 * n_heads) == 0
    ch = width // (3 * n_heads)
    q, k, v = qkv.chunk(3, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum(
        "bct,bcs->bts",
        (q * scale).view(bs * n_heads, ch, length),
        (k * scale).view(bs * n_heads, ch, length),
    )
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    a = torch.einsum("bts,bcs->bct", w, v.reshape(bs * n_heads, ch, length))
    return a.reshape(bs, -1, length)


def attention_block(sd, pfx, x, n_heads, new_order=False):
    b, c, *spatial = x.shape
    x = x.reshape(b, c, -1)
    qkv = _conv(sd, pfx + ".qkv", group_norm32(x, sd[pfx + ".norm.weight"], sd[pfx + ".norm.bias"]))
    h = (qkv_attention_new if new_order else qkv_attention_legacy)(qkv, n_heads)
    h = _conv(sd, pfx + ".proj_out", h)
    return (x + h).reshape(b, c, *spatial)


def _heads(ch, cfg, upsample_side=False):
    nhc = cfg.get("num_head_channels", -1)
    if nhc != -1:
        return ch // nhc
    if upsample_side and cfg.get("num_heads_upsample", -1) != -1:
        return cfg["num_heads_upsample"]
    return cfg.get("num_heads", 1)


def unet_forward(sd, cfg, x, timesteps, cond=None, y=None, drop=None):
    """cfg keys mirror UNetModel.__init__ (unet_openai.py:553-575): model_channels,
    num_res_blocks, attention_resolutions, channel_mult, num_classes, num_heads,
    num_head_channels, num_heads_upsample, use_scale_shift_norm, resblock_updown,
    use_new_attention_order, conv_resample.  drop: see res_block (training-mode dropout with injected masks)."""
    mc = cfg["model_channels"]
    mult = tuple(cfg.get("channel_mult", (1, 2, 4, 8)))
    nrb = cfg["num_res_blocks"]
    attn_res = tuple(cfg.get("attention_resolutions", ()))
    film = cfg.get("use_scale_shift_norm", False)
    rud = cfg.get("resblock_updown", False)
    new_order = cfg.get("use_new_attention_order", False)
    conv_resample = cfg.get("conv_resample", True)
    num_classes = cfg.get("num_classes", None)

    if cond is not None:
        x = torch.cat([x, cond], 1)
    assert (y is not None) == (num_classes is not None)

    emb = timestep_embedding(timesteps, mc)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    if num_classes is not None:
        assert y.shape == (x.shape[0],)
        emb = emb + F.embedding(y, sd["label_emb.weight"])

    hs = []
    h = x.float()
    # ---- encoder (unet_openai.py:608-664) ----
    h = _conv(sd, "input_blocks.0.0", h, padding=1)
    hs.append(h)
    idx, ds, ch = 1, 1, int(mult[0] * mc)
    for level, m in enumerate(mult):
        for _ in range(nrb):
            p = f"input_blocks.{idx}"
            h = res_block(sd, p + ".0", h, emb, film=film, drop=drop)
            ch = int(m * mc)
            if ds in attn_res:
                h = attention_block(sd, p + ".1", h, _heads(ch, cfg), new_order)
            hs.append(h)
            idx += 1
        if level != len(mult) - 1:
            p = f"input_blocks.{idx}.0"
            if rud:
                h = res_block(sd, p, h, emb, film=film, down=True, drop=drop)
            else:
                h = downsample(sd, p, h, use_conv=conv_resample)
            hs.append(h)
            idx += 1
            ds *= 2
    # ---- middle (unet_openai.py:666-690) ----
    h = res_block(sd, "middle_block.0", h, emb, film=film, drop=drop)
    h = attention_block(sd, "middle_block.1", h, _heads(ch, cfg), new_order)
    h = res_block(sd, "middle_block.2", h, emb, film=film, drop=drop)
    # ---- decoder (unet_openai.py:693-737, 772-774) ----
    idx = 0
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            p = f"output_blocks.{idx}"
            h = torch.cat([h, hs.pop()], dim=1)
            h = res_block(sd, p + ".0", h, emb, film=film, drop=drop)
            ch = int(mc * m)
            sub = 1
            if ds in attn_res:
                h = attention_block(sd, f"{p}.{sub}", h, _heads(ch, cfg, True), new_order)
                sub += 1
            if level and i == nrb:
                if rud:
                    h = res_block(sd, f"{p}.{sub}", h, emb, film=film, up=True, drop=drop)
                else:
                    h = upsample(sd, f"{p}.{sub}", h, use_conv=conv_resample)
                ds //= 2
            idx += 1
    # ---- head (unet_openai.py:739-743, 780) ----
    h = F.silu(group_norm32(h, sd["out.0.weight"], sd["out.0.bias"]))
    return _conv(sd, "out.2", h, padding=1)
