"""Oracle evaluated with the operand roundings of the HIP path (TEST INFRASTRUCTURE ONLY).

The HIP kernels multiply bf16 operands and accumulate in f32 - in the forward AND in the explicit backward, where the
incoming gradient of every product is rounded to bf16 as well (css_amd/csrc/bgemm.hip).  torch.autograd through the
fp32 oracle therefore differs from the GPU's gradients by "bf16 operand noise" of a few per cent, which by itself
says nothing about whether a small parameter's gradient carries a wrong factor.  ``bf16_operands()`` re-evaluates the
same oracle graph with every Linear and every attention product replaced by

    forward:   bf16(a) @ bf16(b)                         (f32 accumulate)
    backward:  da = bf16(g) @ bf16(b)^T,  db = bf16(a)^T @ bf16(g)

so that what is left between this and the GPU is summation order and the few f32-vs-bf16 storage choices listed in
DESIGN.md - the floor the gradient tests are written against (tools/grad_floor.py prints the per-parameter table).
"""
import contextlib

import torch

from . import conformer as _conformer


def _bf(t):
    return t.to(torch.bfloat16).float()


def _reduce_to(t, shape):
    """sum the broadcast (leading) dimensions of t away so that it has `shape`"""
    while t.dim() > len(shape):
        t = t.sum(0)
    for i, (a, b) in enumerate(zip(t.shape, shape)):
        if a != b:
            t = t.sum(i, keepdim=True)
    return t


class _RoundedMatmul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ab, bb = _bf(a), _bf(b)
        ctx.save_for_backward(ab, bb)
        ctx.shapes = (a.shape, b.shape)
        return ab @ bb

    @staticmethod
    def backward(ctx, g):
        ab, bb = ctx.saved_tensors
        gb = _bf(g)
        da = gb @ bb.transpose(-1, -2)
        db = ab.transpose(-1, -2) @ gb
        return _reduce_to(da, ctx.shapes[0]), _reduce_to(db, ctx.shapes[1])


def rounded_matmul(a, b):
    return _RoundedMatmul.apply(a, b)


@contextlib.contextmanager
def bf16_operands():
    """Inside the block oracle.conformer's Linears and attention products use bf16 operands in forward and backward."""
    orig_lin, orig_mm = _conformer._lin, _conformer._mm

    def lin(x, sd, p):
        w, b = sd[p + ".weight"], sd[p + ".bias"]
        return rounded_matmul(x, w.t()) + b
    _conformer._lin, _conformer._mm = lin, rounded_matmul
    try:
        yield
    finally:
        _conformer._lin, _conformer._mm = orig_lin, orig_mm
