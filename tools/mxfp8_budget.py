#!/usr/bin/env python3
"""Does block-scaled fp8 (OCP MXFP8: e4m3 elements, one e8m0 power-of-two scale per 32 consecutive K elements - the
operand format of v_mfma_scale_f32_16x16x128_f8f6f4, the only 8-bit MFMA form that is faster than bf16 on CDNA4) fit
the 1e-4 mask-MSE budget of BASELINE.json configs[4] ("fp8 MFMA attention/FFN")?

CPU emulation on the ORACLE: the chosen Linear layers get both operands (weights and input activations) quantised
block-wise along their reduction dimension before an fp32 product; every other Linear gets bf16 operands (what the
build runs today); LayerNorm, softmax, BatchNorm, residuals stay fp32.  Reported: mask MSE against the all-fp32 oracle.
Weights: the seeded synthetic model the goldens use, and (with --train-steps N, needs the GPU) the same architecture
after N optimiser steps of css_amd training on on-device synthetic mixtures - trained weights have heavier tails than
the Gaussian initialisation.

    python tools/mxfp8_budget.py [--blocks 16] [--large] [--train-steps 300]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from css_amd.utils import synthetic as syn  # noqa: E402


def mxfp8(t):
    """Quantise along the LAST dim in blocks of 32: scale 2^(floor(log2(max|block|)) - 8), elements rounded to e4m3
    (saturating at 448), as the OCP MX specification defines MXFP8 (E4M3)."""
    shp = t.shape
    K = shp[-1]
    pad = (-K) % 32
    x = F.pad(t, (0, pad)).reshape(-1, (K + pad) // 32, 32)
    amax = x.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -126)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)
    q = (x / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * scale
    return q.reshape(*shp[:-1], K + pad)[..., :K]


def bf(t):
    return t.to(torch.bfloat16).float()


GROUPS = {
    "ffn": (".net.0", ".net.3"),
    "ffn+qkv+out": (".net.0", ".net.3", ".linear_q", ".linear_k", ".linear_v", ".linear_out"),
    "all linears": ("",),
}


def run(f, sd, blocks, n_head, num_spk, which):
    orig = oracle.conformer._lin

    def lin(x, sdd, p):
        w, b = sdd[p + ".weight"], sdd[p + ".bias"]
        if which == "fp32":
            return F.linear(x, w, b)
        if which != "bf16" and any(p.endswith(sfx) for sfx in GROUPS[which]):
            return F.linear(mxfp8(x), mxfp8(w), b)
        return F.linear(bf(x), bf(w), b)
    oracle.conformer._lin = lin
    try:
        with torch.no_grad():
            h = oracle.conformer.encoder(oracle.conformer.mvn_time(f), sd, n_head=n_head, num_blocks=blocks)
            return torch.relu(lin(h, sd, "linear"))
    finally:
        oracle.conformer._lin = orig


def trained_state(a, shapes_kw, steps):
    """`steps` optimiser steps of css_amd training (HIP forward / backward, fused Adam) on on-device mixtures."""
    from css_amd.datasets import DATASETS
    from css_amd.models import MODELS
    from css_amd.objectives import OBJECTIVES
    from css_amd.optim import FusedAdam
    conf = dict(attention_dim=shapes_kw["attention_dim"], attention_heads=shapes_kw["attention_heads"],
                linear_units=shapes_kw["linear_units"], num_blocks=a.blocks, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True)
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    m.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(num_blocks=a.blocks, **{k: v for k, v in shapes_kw.items()}), 31))
    m = m.cuda().train()
    opt = FusedAdam(m.parameters(), lr=3e-4, max_norm=5.0)
    ds = iter(DATASETS["OnDeviceMixture"](batch_size=16, frames=200, seed=3))
    obj = OBJECTIVES["MSE"].build_objective({})
    for i in range(steps):
        opt.zero_grad()
        loss = obj(m, next(ds), device="cuda")
        loss.backward()
        opt.step()
        if i % 50 == 0 or i == steps - 1:
            print(f"  train step {i:4d} loss {float(loss):.5f}", flush=True)
    return {k: v.detach().float().cpu() for k, v in m.state_dict().items()}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=16)
    ap.add_argument("--large", action="store_true", help="Conformer-large widths (d 512, 8 heads, ffn 2048) instead of base")
    ap.add_argument("--train-steps", type=int, default=0)
    a = ap.parse_args()
    torch.set_num_threads(16)
    kw = dict(attention_dim=512, attention_heads=8, linear_units=2048) if a.large else \
        dict(attention_dim=256, attention_heads=4, linear_units=1024)
    f = torch.from_numpy(np.random.RandomState(11).standard_normal((2, 100, 257)).astype(np.float32)).abs()
    states = [("seeded synthetic weights", syn.fill_state_dict(syn.conformer_shapes(num_blocks=a.blocks, **kw), 31))]
    if a.train_steps:
        states.append((f"after {a.train_steps} training steps", trained_state(a, kw, a.train_steps)))
    for name, sd in states:
        ref = run(f, sd, a.blocks, kw["attention_heads"], 2, "fp32")
        print(f"{name} ({'large' if a.large else 'base'}, {a.blocks} blocks): masks mean {float(ref.mean()):.3f}")
        for which in ("bf16", "ffn", "ffn+qkv+out", "all linears"):
            m = run(f, sd, a.blocks, kw["attention_heads"], 2, which)
            tag = "bf16 operands everywhere (today)" if which == "bf16" else f"MXFP8 operands in: {which}"
            print(f"  {tag:42s} mask MSE vs fp32 {float(((m - ref) ** 2).mean()):.3e}   (budget 1e-4)")
