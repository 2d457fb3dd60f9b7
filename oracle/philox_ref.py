"""Oracle (numpy) for the product's own counter-based noise source eod_randn_philox (test infrastructure).

Philox4x32-10 (Salmon et al., SC'11) + Box-Muller; counter = (element_index/4, global sample index, step,
stream_id), key = 64-bit seed.  The reference draws with torch's global generator (model.py:48,55), which
cannot be made invariant to multi-GPU sharding; this generator is the build's addition.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def _philox(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = p0.astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return c0, c1, c2, c3


def _u01(u):
    return ((u >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def philox_randn(n, chw, seed, sample0, step, stream_id):
    quads = (chw + 3) // 4
    out = np.zeros((n, quads * 4), np.float32)
    q = np.arange(quads, dtype=np.uint64)
    for i in range(n):
        c0, c1, c2, c3 = _philox(q.astype(np.uint32), np.full(quads, (sample0 + i) & 0xFFFFFFFF, np.uint32),
                                 np.full(quads, step & 0xFFFFFFFF, np.uint32),
                                 np.uint32(stream_id) ^ (q >> np.uint64(32)).astype(np.uint32),
                                 seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        r0 = np.sqrt(np.float32(-2.0) * np.log(_u01(c0)))
        r1 = np.sqrt(np.float32(-2.0) * np.log(_u01(c2)))
        a0 = np.float32(6.28318530717958647692) * _u01(c1)
        a1 = np.float32(6.28318530717958647692) * _u01(c3)
        z = np.stack([r0 * np.cos(a0), r0 * np.sin(a0), r1 * np.cos(a1), r1 * np.sin(a1)], 1).astype(np.float32)
        out[i] = z.reshape(-1)
    return out[:, :chw]
