#!/usr/bin/env python3
"""What "fp8 MFMA attention" (BASELINE.json configs[4]) would cost in mask MSE, measured BEFORE building a kernel
(VERDICT r3 missing 1): the attention's matrix products evaluated with block-scaled fp8 (OCP MXFP8, e4m3 + one e8m0 scale
per 32 elements of the reduction dimension - the only 8-bit MFMA form that is faster than bf16 on gfx950) while the
projections stay bf16, on the TRAINED checkpoint of tools/fp8_trained_parity.py (Conformer-base, 16 blocks, seeded
initialisation, 300 optimiser steps of css_amd training) and on the untrained seeded weights.
Every row: mask MSE of the oracle evaluated with those operand formats against the fp32 oracle; budget 1e-4.
    python tools/fp8_attn_budget.py [--steps 300]        (needs the GPU for the training steps only)"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from css_amd.utils import synthetic as syn  # noqa: E402


def checkpoint(steps, blocks=16):
    from css_amd.datasets import DATASETS
    from css_amd.models import MODELS
    from css_amd.objectives import OBJECTIVES
    from css_amd.optim import FusedAdam
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=blocks, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True)
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    m.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(num_blocks=blocks), 31))
    m = m.cuda().train()
    opt = FusedAdam(m.parameters(), lr=3e-4, max_norm=5.0)
    ds = iter(DATASETS["OnDeviceMixture"](batch_size=16, frames=200, seed=3))
    obj = OBJECTIVES["MSE"].build_objective({})
    for _ in range(steps):
        opt.zero_grad()
        obj(m, next(ds), device="cuda").backward()
        opt.step()
    return {k: v.detach().float().cpu() for k, v in m.state_dict().items()}


def table(sd, blocks=16):
    f = torch.from_numpy(np.random.RandomState(11).standard_normal((2, 100, 257)).astype(np.float32)).abs()
    torch.set_num_threads(16)
    mse = lambda a, b: float(((a - b) ** 2).mean())  # noqa: E731
    rows = []
    with torch.no_grad():
        ref = torch.cat(oracle.conformer.forward(f, sd, num_blocks=blocks)[1], -1)
        for name, kw in (("bf16 everywhere (the HIP path)", dict(scores=False, pv=False)),
                         ("q k^T and q pe^T in MXFP8", dict(scores=True, pv=False)),
                         ("P v in MXFP8", dict(scores=False, pv=True)),
                         ("all three attention products in MXFP8", dict(scores=True, pv=True))):
            with oracle.mxfp8.attention_emulation(**kw):
                got = torch.cat(oracle.conformer.forward(f, sd, num_blocks=blocks)[1], -1)
            rows.append((name, mse(got, ref)))
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    for steps in (0, a.steps):
        print(f"Conformer-base, 16 blocks, after {steps} training steps: mask MSE vs the fp32 oracle (budget 1e-4)", flush=True)
        for name, e in table(checkpoint(steps)):
            print(f"    {name:44s} {e:.3e}", flush=True)
