#!/usr/bin/env python3
"""Parity statement of the opt-in MXFP8 FeedForward on a TRAINED checkpoint.

The deterministic recipe: Conformer-base (16 blocks) from the seeded synthetic initialisation (seed 31), `steps`
optimiser steps of css_amd training (HIP forward / backward, FusedAdam lr 3e-4, clip 5) on on-device synthetic mixtures
(OnDeviceMixture, batch 16 x 200 frames, seed 3); then one evaluation batch (2 x 100 frames, seed 11) through
  * the fp32 oracle (CPU),
  * the HIP model, bf16 path,
  * the HIP model with fp8_ffn = True (csship_ffn_mx in every FeedForward),
  * the oracle with the same operand formats (oracle/mxfp8.py).
Reported: mask MSE of each against the fp32 oracle.  python tools/fp8_trained_parity.py [--steps 300]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from css_amd.utils import synthetic as syn  # noqa: E402


def trained_model(steps=300, blocks=16, verbose=True):
    """the recipe's model after `steps` optimiser steps (on the GPU, train mode) and its dataset iterator"""
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
    for i in range(steps):
        opt.zero_grad()
        loss = obj(m, next(ds), device="cuda")
        loss.backward()
        opt.step()
        if verbose and (i % 100 == 0 or i == steps - 1):
            print(f"  train step {i:4d} loss {float(loss):.5f}", flush=True)
    return m, ds


def run(steps=300, blocks=16, verbose=True):
    m, _ = trained_model(steps, blocks, verbose)
    m.eval()
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    f = torch.from_numpy(np.random.RandomState(11).standard_normal((2, 100, 257)).astype(np.float32)).abs()
    torch.set_num_threads(16)
    with torch.no_grad():
        _, ref = oracle.conformer.forward(f, sd, num_blocks=blocks)
        with oracle.mxfp8.ffn_emulation("bf16"):
            _, emu = oracle.conformer.forward(f, sd, num_blocks=blocks)
        ref, emu = torch.cat(ref, -1), torch.cat(emu, -1)
        m.fp8_ffn = False
        bf16 = m.forward_masks(f.cuda()).cpu()
        m.fp8_ffn = True
        fp8 = m.forward_masks(f.cuda()).cpu()
    mse = lambda a, b: float(((a - b) ** 2).mean())  # noqa: E731
    return {"steps": steps, "bf16_vs_fp32": mse(bf16, ref), "fp8_vs_fp32": mse(fp8, ref), "fp8_vs_emulation": mse(fp8, emu),
            "emulation_vs_fp32": mse(emu, ref), "mask_mean": float(ref.mean())}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    for steps in (0, a.steps):
        r = run(steps)
        print(f"after {steps} training steps (Conformer-base, 16 blocks; masks mean {r['mask_mean']:.3f}): mask MSE vs the fp32 oracle - "
              f"HIP bf16 {r['bf16_vs_fp32']:.3e}, HIP fp8_ffn {r['fp8_vs_fp32']:.3e} (budget 1e-4), MXFP8-emulating oracle "
              f"{r['emulation_vs_fp32']:.3e}; HIP fp8_ffn vs that emulation {r['fp8_vs_emulation']:.3e}", flush=True)
