"""Oracle side of the build's dropout (training mode).

The reference draws its dropout masks from torch's global generator (nn.Dropout, css/models/conformer.py:127,249,254,
302,317,319; css/models/blstm.py:147,158): a stream no other implementation can reproduce, so parity in training mode
is defined per MASK - the build generates a mask as a pure function of (seed, site, element index)
(include/csship.h, "dropout") and this module states the same function in numpy.  The oracle's forward functions take
a ``drop(x, site)`` callable; ``Masks`` builds one that applies exactly the masks the HIP kernels use.
Test infrastructure only - see oracle/__init__.py.
"""
import numpy as np
import torch


def hash32(x):
    """lowbias32 mixer on uint32 arrays (wraps modulo 2^32)."""
    x = np.asarray(x, dtype=np.uint64)
    x = (x ^ (x >> np.uint64(16))) & np.uint64(0xffffffff)
    x = (x * np.uint64(0x7feb352d)) & np.uint64(0xffffffff)
    x = x ^ (x >> np.uint64(15))
    x = (x * np.uint64(0x846ca68b)) & np.uint64(0xffffffff)
    x = x ^ (x >> np.uint64(16))
    return x.astype(np.uint64)


def hash_derive(w):
    """second word of a four-element group (css_amd/csrc/common.hpp hash_derive): ((w ^ c) mod 2^24) * m mod 2^32, plus
    w >> 11, then an xorshift."""
    w = np.asarray(w, dtype=np.uint64)
    v = (((w ^ np.uint64(0x5bd1e995)) & np.uint64(0xffffff)) * np.uint64(0x85ebca)) & np.uint64(0xffffffff)
    v = (v + (w >> np.uint64(11))) & np.uint64(0xffffffff)
    return v ^ (v >> np.uint64(14))


def keep_mask(seed, stream, n, p):
    """bool [n]: elements 4q .. 4q+3 share w0 = hash32(key + q) and w1 = hash_derive(w0); element 4q + f survives iff field
    f of (w0 lo16, w0 hi16, w1 lo16, w1 hi16) >= round(p * 65536)."""
    thresh = int(p * 65536.0 + 0.5)
    if thresh == 0:
        return np.ones(n, dtype=bool)
    key = int(hash32(np.uint64((int(seed) ^ ((int(stream) * 0x9E3779B9) & 0xffffffff)) & 0xffffffff)))
    q = np.arange((n + 3) // 4, dtype=np.uint64)
    w0 = hash32((np.uint64(key) + q) & np.uint64(0xffffffff))
    w1 = hash_derive(w0)
    m16 = np.uint64(0xffff)
    field = np.stack([w0 & m16, w0 >> np.uint64(16), w1 & m16, w1 >> np.uint64(16)], 1).reshape(-1)[:n]
    return field >= thresh


def scale(p):
    thresh = int(p * 65536.0 + 0.5)
    return 65536.0 / (65536 - thresh)


class Masks:
    """drop(x, site) for the oracle: site -> stream id comes from ``streams`` (the build's numbering, css_amd.dropout)."""

    def __init__(self, seed, p, streams):
        self.seed, self.p, self.streams = int(seed), float(p), streams

    def __call__(self, x, site):
        if self.p <= 0.0:
            return x
        k = keep_mask(self.seed, self.streams(site), x.numel(), self.p).reshape(tuple(x.shape))
        return x * torch.from_numpy(k.astype(np.float32) * np.float32(scale(self.p)))
