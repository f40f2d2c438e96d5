#!/usr/bin/env python3
"""Would fp8 (OCP e4m3) GEMM operands fit the 1e-4 mask-MSE budget?  CPU emulation on the oracle: every Linear's
weights (per-output-channel scale) and, optionally, its input activations (per-row scale) are rounded to
float8_e4m3fn before the product; everything else stays fp32.  Reports mask MSE vs the unquantised oracle.

    python tools/fp8_budget.py [--blocks 16]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from css_amd.utils import synthetic as syn  # noqa: E402

E4M3_MAX = 448.0


def q8(t, dim):
    s = t.abs().amax(dim=dim, keepdim=True).clamp_min(1e-12) / E4M3_MAX
    return (t / s).to(torch.float8_e4m3fn).float() * s


def qbf(t):
    return t.to(torch.bfloat16).float()


def run(mode, f, sd, blocks):
    orig = F.linear

    def lin(x, w, b=None):
        if mode == "bf16":
            return orig(qbf(x), qbf(w), b)
        if mode == "fp8-weights":
            return orig(qbf(x), q8(w, 1), b)
        if mode == "fp8-both":
            return orig(q8(x, -1), q8(w, 1), b)
        return orig(x, w, b)
    F.linear = lin
    try:
        _, masks = oracle.conformer.forward(f, sd, num_blocks=blocks)
    finally:
        F.linear = orig
    return torch.cat(masks, 2)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=16)
    a = ap.parse_args()
    torch.manual_seed(0)
    sd = syn.fill_state_dict(syn.conformer_shapes(num_blocks=a.blocks), 31 if a.blocks == 16 else 30)
    import numpy as np
    f = torch.from_numpy(np.random.RandomState(11).standard_normal((2, 100, 257)).astype(np.float32)).abs()
    ref = run("fp32", f, sd, a.blocks)
    for mode in ("bf16", "fp8-weights", "fp8-both"):
        m = run(mode, f, sd, a.blocks)
        print(f"{mode:12s} Linear operands: mask MSE vs fp32 {float(((m - ref) ** 2).mean()):.3e}   (budget 1e-4; masks mean {float(ref.mean()):.3f})")
