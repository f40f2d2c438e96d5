#!/usr/bin/env python3
"""Diagnostic: phase timing inside csship_outproj_conv (s_memrealtime stamps, 100 MHz)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import ops, _lib
lib = _lib.lib()
h = C.CDLL(_lib.LIB_PATH)
h.csship_debug_stamps.argtypes = [C.c_void_p]
DEV = "cuda"; B, T, D = 32, 400, 256
M = B * T
x = torch.randn(M, D, device=DEV); out = torch.empty_like(x)
ao = torch.randn(M, D, device=DEV).to(torch.bfloat16)
wo = ops.pack_frag(torch.randn(D, D, device=DEV) / 16); bo = torch.zeros(D, device=DEV)
W = {"ln_g": torch.ones(D, device=DEV), "ln_b": torch.zeros(D, device=DEV), "pw1_w": torch.tensor([0.9, -0.7], device=DEV),
     "pw1_b": torch.tensor([0.1, 0.2], device=DEV), "dw_w": torch.randn(33, D, device=DEV) * 0.2, "dw_b": torch.zeros(D, device=DEV),
     "bn_g": torch.ones(D, device=DEV), "bn_b": torch.zeros(D, device=DEV), "bn_mean": torch.zeros(D, device=DEV),
     "bn_var": torch.ones(D, device=DEV), "pw2_w": torch.tensor([1.3], device=DEV), "pw2_b": torch.tensor([-0.2], device=DEV)}
cw = ops.conv_weights_struct(W)
nwg = 8 * B
buf = torch.zeros(nwg, 8, device=DEV, dtype=torch.int64)
fn = lambda: _lib.check(lib.csship_outproj_conv(ops._ptr(ao), ops._ptr(wo), ops._ptr(bo), ops._ptr(x), ops._ptr(out), C.byref(cw), B, T, D, 33, ops._stream()))
for _ in range(3): fn()
torch.cuda.synchronize()
h.csship_debug_stamps(C.c_void_p(buf.data_ptr())); buf.zero_(); torch.cuda.synchronize()
fn(); torch.cuda.synchronize(); h.csship_debug_stamps(None)
s = buf.cpu().double() * 10.0
t0 = s[:, 0].min()
print("wg start rel ns: med %.0f max %.0f" % ((s[:, 0] - t0).median(), (s[:, 0] - t0).max()))
for i, nm in zip(range(1, 5), ("stage ao panel", "out-proj GEMM + park", "row phase (residual, LN, GLU)", "stencil + BN + store")):
    d = s[:, i] - s[:, i - 1]
    print("  %-32s ns: min %.0f med %.0f max %.0f" % (nm, d.min(), d.median(), d.max()))
print("total %.0f ns" % (s[:, :5].max() - t0))
