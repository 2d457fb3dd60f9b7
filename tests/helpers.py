"""Shared test helpers (CPU side)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gload(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def gt(name):
    """fixture as torch tensors"""
    return {k: torch.from_numpy(np.asarray(v)) for k, v in gload(name).items()}


def unet_cfgs():
    with open(os.path.join(GOLDEN, "unet_cfgs.json")) as f:
        return json.load(f)


def key_contracts():
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        return json.load(f)


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def bits_equal(a, b):
    a = a.contiguous()
    b = b.contiguous()
    return a.shape == b.shape and a.dtype == b.dtype and bool((a.view(torch.int32) == b.view(torch.int32)).all())


def close_ulp(a, b, rel=4e-7):
    """|a-b| <= rel * max(|b|, tiny) elementwise scaled by the tensor's magnitude: a few fp32 ulps.
    Used where the reference's own bits are machine-dependent (torch's AVX512 CPU sqrt is not correctly
    rounded, see oracle/sampler_ref.py:_sqrt)."""
    a = a.double()
    b = b.double()
    scale = b.abs().max().clamp_min(1e-30)
    return bool(((a - b).abs() <= rel * torch.maximum(b.abs(), 1e-3 * scale) * 4).all())
