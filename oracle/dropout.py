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


def keep_mask(seed, stream, n, p):
    """bool [n]: element e survives iff field (e & 1) of hash32(key + (e >> 1)) >= round(p * 65536)."""
    thresh = int(p * 65536.0 + 0.5)
    if thresh == 0:
        return np.ones(n, dtype=bool)
    key = int(hash32(np.uint64((int(seed) ^ ((int(stream) * 0x9E3779B9) & 0xffffffff)) & 0xffffffff)))
    e = np.arange(n, dtype=np.uint64)
    h = hash32((np.uint64(key) + (e >> np.uint64(1))) & np.uint64(0xffffffff))
    field = np.where((e & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xffff))
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
