#!/usr/bin/env python3
"""Diagnostic: per-phase timing inside the panel kernels via s_memrealtime stamps (100 MHz)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import ops, _lib
lib = _lib.lib()
h = C.CDLL(_lib.LIB_PATH)
h.csship_debug_stamps.argtypes = [C.c_void_p]
DEV = "cuda"; M, D = 12800, 256
x = torch.randn(M, D, device=DEV); g = torch.ones(D, device=DEV); b = torch.zeros(D, device=DEV)
nwg = (M + 63) // 64
buf = torch.zeros(nwg, 8, device=DEV, dtype=torch.int64)

def report(name, fn, nph=6):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    h.csship_debug_stamps(C.c_void_p(buf.data_ptr())); buf.zero_(); torch.cuda.synchronize()
    fn(); torch.cuda.synchronize(); h.csship_debug_stamps(None)
    s = buf.cpu().double() * 10.0  # ns
    t0 = s[:, 0].min()
    print(name)
    print("  wg start (rel to first) ns: min %.0f med %.0f max %.0f" % (0, (s[:,0]-t0).median(), (s[:,0]-t0).max()))
    for i in range(1, nph):
        if (s[:, i] == 0).all(): continue
        prev = max(j for j in range(i) if not (s[:, j] == 0).all())
        d = s[:, i] - s[:, prev]
        print("  phase %d-%d ns: min %.0f med %.0f max %.0f" % (prev, i, d.min(), d.median(), d.max()))
    print("  total last end - first start: %.0f ns" % (s[:, :nph].max() - t0))

ff = 1024
w1 = ops.pack_frag(torch.randn(ff, D, device=DEV) / 16); w2 = ops.pack_frag(torch.randn(D, ff, device=DEV) / 32)
b1 = torch.zeros(ff, device=DEV); b2 = torch.zeros(D, device=DEV); out = torch.empty_like(x)
report("ffn_fused 1024 final_ln=1 (0 start,1 prologue done,2 mma done,3 residual done,4 LN done,5 stored)",
       lambda: ops.ffn_fused(x, (g, b), w1, b1, w2, b2, ff, 0.5, final_ln=(g, b), out=out))
for N in (256, 768):
    w = ops.pack_frag(torch.randn(N, D, device=DEV) / 16); bias = torch.zeros(N, device=DEV)
    ob = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    report(f"panel_linear LN->bf16 N={N} (0 start,1 prologue,2 first pass mma,5 end)",
           lambda: ops.panel_linear(x, w, bias, N, ln=(g, b), out_bf16=True, out=ob))
