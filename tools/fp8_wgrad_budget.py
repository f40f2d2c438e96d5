#!/usr/bin/env python3
"""What would MXFP8 operands in the WEIGHT-GRADIENT products cost in gradient accuracy?  

DESIGN "next" 3 names fp8 operands as the one lever left on bgemm_wgrad_group_kernel (it is bound by operand delivery:
169 MB of bf16 rows per encoder block at 3.2 TB/s).  Before anyone rewrites that loop this prices the arithmetic: the
oracle's autograd (tests/grad_ref.py's graph) evaluated on the deterministic trained checkpoint of
tools/fp8_trained_parity.py (Conformer-base, 16 blocks, 300 optimiser steps) and on the seeded initialisation, with
  fp32      everything in fp32                                              (the reference)
  bf16      bf16 operands in every product, forward and backward            (oracle/emulate.py = what the HIP path does)
  mx-ffn    as bf16, but dW = dY^T X of the four FeedForward Linears of every block on MXFP8 operands
  mx-all    as bf16, but the weight gradient of EVERY Linear on MXFP8 operands
MXFP8 = e4m3 elements + one e8m0 scale per 32 consecutive elements of the REDUCTION dimension (the B*T rows), both
operands (oracle/mxfp8.quantize) - the only 8-bit MFMA form that is faster than bf16 on gfx950.
Printed: global relative L2 error of the parameter gradients against fp32, the same over the FeedForward weights only,
and the worst FeedForward weight.  python tools/fp8_wgrad_budget.py [--steps 300] [--frames 100] [--batch 4]"""
import argparse
import contextlib
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import conformer as oc, emulate, mxfp8  # noqa: E402
import fp8_trained_parity as tp  # noqa: E402


def _bf(t):
    return t.to(torch.bfloat16).float()


class _LinearMxWgrad(torch.autograd.Function):
    """y = bf16(x) bf16(w)^T; dx = bf16(g) bf16(w); dw = mx(g)^T mx(x), blocks of 32 along the rows (the reduction)"""

    @staticmethod
    def forward(ctx, x, w):
        xb, wb = _bf(x), _bf(w)
        ctx.save_for_backward(x, wb)
        return xb @ wb.t()

    @staticmethod
    def backward(ctx, g):
        x, wb = ctx.saved_tensors
        dx = _bf(g) @ wb
        g2, x2 = g.reshape(-1, g.shape[-1]), x.reshape(-1, x.shape[-1])
        # operands as the kernel would see them: bf16 tensors (what the producers store), then block-quantised along rows
        qg = mxfp8.quantize(_bf(g2).t().contiguous())   # [out, rows]
        qx = mxfp8.quantize(_bf(x2).t().contiguous())   # [in, rows]
        return dx, qg @ qx.t()


@contextlib.contextmanager
def mx_wgrad(which):
    """bf16 operands everywhere; the weight gradient of the Linears selected by which(prefix) on MXFP8 operands"""
    with emulate.bf16_operands():
        inner = oc._lin

        def lin(x, sd, p):
            if which(p):
                return _LinearMxWgrad.apply(x, sd[p + ".weight"]) + sd[p + ".bias"]
            return inner(x, sd, p)
        oc._lin = lin
        try:
            yield
        finally:
            oc._lin = inner


def grads(sd, batch, ctx):
    sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    with ctx:
        y, _ = oracle.conformer.forward(batch["mix"], sdg, num_blocks=len([k for k in sd if k.endswith("feed_forward_in.net.0.weight")]))
        oracle.pit.mse_objective(y, batch["source1"], batch["source2"]).backward()
    return {k: v.grad for k, v in sdg.items() if v.grad is not None}


def rel(g, ref, keys):
    num = sum(float(((g[k] - ref[k]) ** 2).sum()) for k in keys)
    den = sum(float((ref[k] ** 2).sum()) for k in keys)
    return (num / den) ** 0.5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    torch.set_num_threads(16)
    for steps in (0, a.steps):
        m, ds = tp.trained_model(steps, verbose=False)
        sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
        b = next(ds)
        batch = {k: b[k][:a.batch, :a.frames].float().cpu().contiguous() for k in ("mix", "source1", "source2")}
        is_ffn = lambda p: p.endswith(".net.0") or p.endswith(".net.3")  # noqa: E731
        g32 = grads(sd, batch, contextlib.nullcontext())
        res = {"bf16": grads(sd, batch, emulate.bf16_operands()), "mx-ffn": grads(sd, batch, mx_wgrad(is_ffn)),
               "mx-all": grads(sd, batch, mx_wgrad(lambda p: True))}
        allk = [k for k in g32 if float((g32[k] ** 2).sum()) > 0]
        ffk = [k for k in allk if (".net.0.weight" in k or ".net.3.weight" in k)]
        print(f"after {steps} training steps, batch {a.batch} x {a.frames} frames: relative L2 error of the parameter gradients vs fp32 autograd")
        for name, g in res.items():
            worst = max(ffk, key=lambda k: rel(g, g32, [k]))
            print(f"  {name:7s} all parameters {rel(g, g32, allk):.3e}   FeedForward weights {rel(g, g32, ffk):.3e}   worst FeedForward weight "
                  f"{worst} {rel(g, g32, [worst]):.3e}", flush=True)
        del m


if __name__ == "__main__":
    main()
