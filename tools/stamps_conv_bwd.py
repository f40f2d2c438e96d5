#!/usr/bin/env python3
"""Diagnostic: phase timing inside the fused conv-module backward kernel (s_memrealtime stamps, 100 MHz)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import ops, ops_bwd as ob, _lib
from css_amd.dropout import DropState
lib = _lib.lib()
h = C.CDLL(_lib.LIB_PATH)
h.csship_debug_stamps.argtypes = [C.c_void_p]
DEV = "cuda"; B, T, D = 32, 400, 256
W = {"ln_g": torch.ones(D, device=DEV), "ln_b": torch.zeros(D, device=DEV), "pw1_w": torch.tensor([0.9, -0.7], device=DEV),
     "pw1_b": torch.tensor([0.1, 0.2], device=DEV), "dw_w": torch.randn(33, D, device=DEV) * 0.2, "dw_b": torch.zeros(D, device=DEV),
     "bn_g": torch.ones(D, device=DEV), "bn_b": torch.zeros(D, device=DEV), "bn_mean": torch.zeros(D, device=DEV),
     "bn_var": torch.ones(D, device=DEV), "pw2_w": torch.tensor([1.3], device=DEV), "pw2_b": torch.tensor([-0.2], device=DEV)}
x = torch.randn(B, T, D, device=DEV); dy = torch.randn(B, T, D, device=DEV) * 0.1
st = DropState(torch.tensor([5], dtype=torch.int32, device=DEV), 0.1, 0)
c_out, c_next = st.cfg("0.conv.out"), st.cfg("0.attn.out")
_, fws = ops.convmod(x, W, 33, batch_stats=True, return_workspace=True, drop=c_out)
fn = lambda: ob.convmod_bwd(x, dy, W, fws, 33, drop=c_out, want_next=True, next_drop=c_next)
for _ in range(3): fn()
torch.cuda.synchronize()
nwg = ((T + 31) // 32) * B
buf = torch.zeros(4096, 8, device=DEV, dtype=torch.int64)
h.csship_debug_stamps(C.c_void_p(buf.data_ptr())); buf.zero_(); torch.cuda.synchronize()
fn(); torch.cuda.synchronize(); h.csship_debug_stamps(None)
s = buf[:nwg].cpu().double() * 10.0
t0 = s[:, 0].min()
print("%d workgroups; start rel ns: med %.0f max %.0f" % (nwg, (s[:, 0] - t0).median(), (s[:, 0] - t0).max()))
names = ("0 dc rows (dy, z loads) -> LDS", "1 dg = dc * taps, store", "2 LN + GLU rows -> LDS", "3 dw sums", "4 hand-over + partial store",
         "C rows: loads, GLU / LN backward, dx store", "column sums + tail")
for i, nm in zip(range(1, 8), names):
    d = s[:, i] - s[:, i - 1]
    print("  %-46s ns: min %6.0f med %6.0f max %6.0f" % (nm, d.min(), d.median(), d.max()))
print("kernel span %.0f ns; per-workgroup total med %.0f" % (s[:, 7].max() - t0, (s[:, 7] - s[:, 0]).median()))
