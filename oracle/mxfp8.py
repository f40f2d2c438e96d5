"""Oracle for the opt-in MXFP8 FeedForward (TEST INFRASTRUCTURE ONLY): CPU emulation of OCP MXFP8 operands.

BASELINE.json configs[4] names "fp8 MFMA attention/FFN on CDNA4".  The only 8-bit MFMA form that is faster than bf16 on
gfx950 is the block-scaled one (v_mfma_scale_f32_16x16x128_f8f6f4): e4m3 elements with one e8m0 (power-of-two) scale
per 32 consecutive elements of the reduction dimension, for BOTH operands.  The reference has no such path (it is fp32
PyTorch), so what is pinned here is the FORMAT, against the OCP Microscaling Formats (MX) v1.0 specification:
    shared exponent of a block = floor(log2(max|x|)) - emax_elem,  emax_elem(e4m3) = 8
    element = round-to-nearest-even(x / 2^shared) to e4m3, saturating at +-448
``quantize`` restates that; ``ffn_emulation`` evaluates the oracle Conformer with the FeedForward Linears' operands
(LayerNorm output / ReLU'd hidden activation and both weight matrices) quantised that way and every other Linear's
operands rounded to bf16 - what css_amd computes with ``fp8_ffn=True`` - so the HIP kernel can be checked TIGHTLY
against it, while its distance to the fp32 oracle is the (weight-dependent) price of the format.
"""
import contextlib

import torch
import torch.nn.functional as F

from . import conformer as _conformer


def quantize(t):
    """MXFP8 (e4m3) quantise-dequantise along the LAST dim in blocks of 32 consecutive elements."""
    shp = t.shape
    K = shp[-1]
    pad = (-K) % 32
    x = F.pad(t.float(), (0, pad)).reshape(-1, (K + pad) // 32, 32)
    amax = x.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -126)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)
    q = (x / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * scale
    return q.reshape(*shp[:-1], K + pad)[..., :K]


def _bf16(t):
    return t.to(torch.bfloat16).float()


@contextlib.contextmanager
def ffn_emulation(other="bf16"):
    """Inside the block oracle.conformer evaluates FeedForward Linears (``*.net.0`` / ``*.net.3``) on MXFP8 operands;
    ``other``: operand rounding of every other Linear ("bf16" = what the HIP path does, "fp32" = none)."""
    orig = _conformer._lin
    cache = {}

    def lin(x, sd, p):
        w, b = sd[p + ".weight"], sd[p + ".bias"]
        if p.endswith(".net.0") or p.endswith(".net.3"):
            if p not in cache:
                cache[p] = quantize(w)
            return F.linear(quantize(x), cache[p], b)
        if other == "bf16":
            return F.linear(_bf16(x), _bf16(w), b)
        return F.linear(x, w, b)
    _conformer._lin = lin
    try:
        yield
    finally:
        _conformer._lin = orig


@contextlib.contextmanager
def attention_emulation(scores=True, pv=True, other="bf16"):
    """Inside the block oracle.conformer evaluates the attention's matrix products (conformer.py:229-253) on MXFP8 operands,
    quantised along the REDUCTION dimension of each product in blocks of 32 (what a block-scaled MFMA would consume):
    ``scores``: q k^T and q pe^T (reduction over d_k = 64: two blocks per row of q / k / pe); ``pv``: P v (reduction over the
    keys: P in blocks of 32 consecutive keys of a query row, v in blocks of 32 consecutive keys of a feature column).
    Products that are not selected, and every Linear, use ``other`` operands ("bf16" = the HIP path, "fp32" = exact).
    Used by tools/fp8_attn_budget.py to price "fp8 MFMA attention" (BASELINE.json configs[4]) before building a kernel."""
    orig_lin, orig_mm = _conformer._lin, _conformer._mm
    rnd = _bf16 if other == "bf16" else (lambda t: t)

    def mm(a, b):
        is_pv = a.shape[-1] != 64 or (b.shape[-2] != 64)      # P [.., T, T] x v [.., T, 64]; the score products reduce over d_k = 64
        if a.shape[-1] == 64 and b.shape[-2] == 64:
            is_pv = False
        if (is_pv and pv) or (not is_pv and scores):
            return torch.matmul(quantize(a), quantize(b.transpose(-1, -2)).transpose(-1, -2))
        return torch.matmul(rnd(a), rnd(b))

    def lin(x, sd, p):
        return F.linear(rnd(x), rnd(sd[p + ".weight"]), sd[p + ".bias"])
    _conformer._mm = mm
    if other == "bf16":
        _conformer._lin = lin
    try:
        yield
    finally:
        _conformer._lin, _conformer._mm = orig_lin, orig_mm
