"""ORACLE (test infrastructure, NOT product code): numpy restatement of the keep-mask streams of the train-mode dropout sites
(include/stedm_hip.h, "train-mode dropout"; kernels in stedm_amd/csrc/dropmask.hpp). Only tests/ import this.

The reference draws its masks from torch's generator (nn.Dropout at networks/vit_set.py:28-30, 43/62, 49, 187); no other
implementation can reproduce that stream, so the build specifies its own counter-based stream and the oracle applies the SAME masks to the
reference's arithmetic (train-mode nn.Dropout: keep with probability 1 - p, kept values scaled by 1 / (1 - p)).

  Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; pinned below by the Random123 known-answer vectors)
  elementwise sites : element e of a tensor keeps iff u16 >= thr16, thr16 = round(p * 65536);
                      u16 = 16-bit field (e & 7) of Philox(counter = (lo32(e >> 3), hi32(e >> 3), site, 0), key = (lo32 seed, hi32 seed)),
                      field j = bits [16 (j & 1), 16 (j & 1) + 16) of output word j >> 1
  attention site    : one xorshift128 stream (Marsaglia 2003) per (sample-head bh, query q, key half h), state = Philox(counter =
                      (q, bh, site, h), key = seed); per 64-key tile kt = 0, 1, ... the stream yields 16 words, the first the most
                      significant bit-plane of 32 16-bit uniforms; uniform i = 16 sub + e belongs to key
                      64 kt + 32 sub + (e & 3) + 8 (e >> 2) + 4 h; keep iff u16 >= thr16
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over equally shaped uint32 arrays (or scalars). Returns the four output words."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK32 for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def thr16(p: float) -> int:
    t = int(round(float(np.float32(p)) * 65536.0))          # (the C ABI takes p as a float: lrint((double)p * 65536))
    assert 0 <= t <= 65535, "dropout probability must be in [0, 1)"
    return t


def keep_elementwise(n: int, p: float, seed: int, site: int) -> np.ndarray:
    """bool [n]: keep mask of an elementwise site over the tensor's linear element index."""
    g = np.arange((n + 7) // 8, dtype=np.uint64)
    r = philox4x32_10(g & MASK32, g >> np.uint64(32), np.uint64(site), np.uint64(0), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = np.empty((len(g), 8), dtype=np.uint32)
    for j in range(8):
        u[:, j] = (r[j >> 1] >> np.uint32(16 * (j & 1))) & np.uint32(0xFFFF)
    return (u.reshape(-1)[:n] >= thr16(p))


def _xs128_next(s):
    x, y, z, w = s
    t = x ^ (x << np.uint32(11))
    w2 = w ^ (w >> np.uint32(19)) ^ t ^ (t >> np.uint32(8))
    return (y, z, w, w2), w2


def keep_attention(nbh: int, T: int, p: float, seed: int, site: int) -> np.ndarray:
    """bool [nbh, T, T] (sample-head, query, key): keep mask of the attention-probability dropout (vit_set.py:62)."""
    th = thr16(p)
    ntiles = (T + 63) // 64
    bh = np.arange(nbh, dtype=np.uint32)[:, None, None]
    q = np.arange(T, dtype=np.uint32)[None, :, None]
    h = np.arange(2, dtype=np.uint32)[None, None, :]
    shape = (nbh, T, 2)
    st = philox4x32_10(np.broadcast_to(q, shape), np.broadcast_to(bh, shape), np.uint32(site), np.broadcast_to(h, shape),
                       seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    keep = np.zeros((nbh, T, ntiles * 64), dtype=bool)
    old = np.seterr(over="ignore")
    try:
        for kt in range(ntiles):
            u = np.zeros(shape + (32,), dtype=np.uint32)
            bit = np.arange(32, dtype=np.uint32)
            for plane in range(15, -1, -1):
                st, w = _xs128_next(st)
                u |= ((w[..., None] >> bit) & np.uint32(1)) << np.uint32(plane)
            kp = u >= th                                            # [nbh, T, 2, 32]
            for hh in range(2):
                for i in range(32):
                    sub, e = i >> 4, i & 15
                    keep[:, :, kt * 64 + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh] = kp[:, :, hh, i]
    finally:
        np.seterr(**old)
    return keep[:, :, :T]


# Random123 known-answer vectors of philox4x32-10 (kat_vectors: counter words, key words -> output words)
KAT = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
       ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]


def normal_rows(seed: int, sample_ids, n: int, stream: int) -> np.ndarray:
    """numpy restatement of stedm_philox_normal (include/stedm_hip.h): [len(ids), n] float32. Group g of four consecutive elements of a row =
    philox4x32_10(counter {g, stream, 0x4E524D4C, 0}, key {seed & 0xFFFFFFFF, sample id}); Box-Muller on the word pairs with
    u = (w + 0.5) 2^-32 in float32 (the device uses the hardware log / sin / cos: equal to a few float32 ulps, not bitwise)."""
    ng = (n + 3) // 4
    g = np.arange(ng, dtype=np.uint64)
    rows = []
    for sid in sample_ids:
        r = philox4x32_10(g & MASK32, np.uint64(stream), np.uint64(0x4E524D4C), np.uint64(0), seed & 0xFFFFFFFF, int(sid) & 0xFFFFFFFF)
        w = [np.asarray(x, dtype=np.uint64).astype(np.float32) for x in r]
        k = np.float32(2.3283064365386963e-10)
        u = [np.clip((x + np.float32(0.5)) * k, np.float32(1.1641532e-10), np.float32(0.99999994)) if i in (0, 2) else (x + np.float32(0.5)) * k
             for i, x in enumerate(w)]
        ra = np.sqrt(np.float32(-2.0) * np.log(u[0])); rb = np.sqrt(np.float32(-2.0) * np.log(u[2]))
        two_pi = np.float32(6.283185307179586)
        z = np.stack([ra * np.cos(two_pi * u[1]), ra * np.sin(two_pi * u[1]), rb * np.cos(two_pi * u[3]), rb * np.sin(two_pi * u[3])], axis=1)
        rows.append(z.reshape(-1)[:n].astype(np.float32))
    return np.stack(rows)
