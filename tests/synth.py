"""Deterministic synthetic weights / inputs shared by the golden-vector generator and the tests.

Weights are a pure function of (key, shape, seed) so that fixtures only need to store inputs and
expected outputs: the same state_dict is rebuilt on the GPU box without the reference.
Every `zero_module` parameter of the reference (unet_openai.py:340-342, 422, 742) is filled
with non-zero values too -- with the reference's zero init the UNet output is identically 0
and parity would be vacuous.
"""
import zlib

import numpy as np
import torch


def _rng(key, seed):
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def synth_tensor(key, shape, seed=0):
    shape = tuple(int(s) for s in shape)
    r = _rng(key, seed)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "n_averaged":
        return np.zeros(shape, np.int64)
    if len(shape) == 1:
        if leaf == "weight":  # GroupNorm gamma
            return (1.0 + 0.2 * r.standard_normal(shape)).astype(np.float32)
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    if "label_emb" in key:
        return (0.5 * r.standard_normal(shape)).astype(np.float32)
    return (r.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)


def synth_state_dict(shapes, seed=0, as_torch=True):
    """shapes: {key: shape}.  Returns {key: tensor} in the given key order."""
    out = {}
    for k, shp in shapes.items():
        a = synth_tensor(k, shp, seed)
        out[k] = torch.from_numpy(a) if as_torch else a
    return out


def synth_input(tag, shape, seed=0, scale=1.0, uniform=False):
    r = _rng("input:" + tag, seed)
    if uniform:
        return torch.from_numpy(r.random(tuple(shape)).astype(np.float32) * scale)
    return torch.from_numpy((scale * r.standard_normal(tuple(shape))).astype(np.float32))


def rect_mask(n, h, w, seed=0):
    """mask[n,1,h,w] = 1 outside one random axis-aligned rectangle per sample (1 = keep), the
    convention of inference.py:100-109 / script_utils/utils.py:17-37 after `mask = 1 - mask`."""
    r = _rng("mask", seed)
    m = np.ones((n, 1, h, w), np.float32)
    for i in range(n):
        rh = int(r.integers(max(1, h // 10), max(2, (4 * h) // 10) + 1))
        rw = int(r.integers(max(1, w // 10), max(2, (4 * w) // 10) + 1))
        y0 = int(r.integers(0, h - rh + 1))
        x0 = int(r.integers(0, w - rw + 1))
        m[i, 0, y0:y0 + rh, x0:x0 + rw] = 0.0
    return torch.from_numpy(m)
