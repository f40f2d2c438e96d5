#!/usr/bin/env python3
"""Diagnostic: phase boundaries (shader cycles, wave 0 of every workgroup) inside chain2_kernel.  Needs a library built
with -DP_STAMPS:  tools/panel_ablate.sh build stamps;  python tools/with_lib.py build/abl/libp_stamps.so tools/chain2_stamps.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import ops, _lib
lib = _lib.lib(); h = C.CDLL(_lib.LIB_PATH); h.csship_debug_stamps.argtypes = [C.c_void_p]
M, D = 12800, 256
x = torch.randn(M, D, device="cuda")
g1, b0 = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
def ffn_w():
    return ((g1, b0), ops.pack_frag(torch.randn(1024, D, device="cuda") / 16), torch.zeros(1024, device="cuda"),
            ops.pack_frag(torch.randn(D, 1024, device="cuda") / 32), torch.zeros(D, device="cuda"))
fa, fb = ffn_w(), ffn_w()
wq, bq = ops.pack_frag(torch.randn(768, D, device="cuda") / 16), torch.zeros(768, device="cuda")
out = torch.empty_like(x)
run = lambda: ops.encoder_chain(x, 1024, ffn_a=fa, final_ln=(g1, b0), ffn_b=fb, qkv=((g1, b0), wq, bq, 768), out=out)
buf = torch.zeros(256, 32, device="cuda", dtype=torch.int64)
for _ in range(3): run()
torch.cuda.synchronize()
h.csship_debug_stamps(C.c_void_p(buf.data_ptr())); run(); torch.cuda.synchronize(); h.csship_debug_stamps(None)
s = buf[:200].cpu().double()
names = ["", "A: LN -> panel (2 barriers)", "A: passes", "A: residual + final LN", "B: LN -> panel", "B: passes", "B: residual",
         "C: LN -> panel", "C: 3 passes + stores"]
print("cycles between stamps (median / max over the 200 workgroups)")
for i in range(1, 9):
    d = s[:, i] - s[:, i - 1]
    print(f"  {i:2d} {names[i]:34s} {d.median():8.0f} {d.max():8.0f}")
print(f"  stamp 0 -> 8: {(s[:, 8] - s[:, 0]).median():.0f} cycles median; start skew {(s[:, 0].max() - s[:, 0].min()):.0f}, end skew {(s[:, 8].max() - s[:, 8].min()):.0f}")
print("stage B LayerNorm -> panel, from the end of stage A (stamp 3):")
for i, nm in zip(range(10, 15), ["weight prefetch + gamma/beta/bias loads issued", "local stats (permlane sums), LDS write", "barrier", "exchange read + combine", "normalise + panel write"]):
    d = s[:, i] - (s[:, 3] if i == 10 else s[:, i - 1])
    print(f"  {nm:50s} {d.median():8.0f} {d.max():8.0f}")
print(f"  {'closing barrier':50s} {(s[:, 4] - s[:, 14]).median():8.0f}")
print("stage A, chunk 1 (wave 0 | wave 7): W1 pass of chunk 2, hidden store, barrier, W2 pass of chunk 1")
for i, nm in enumerate(["W1 pass", "store_hidden", "barrier", "W2 pass"]):
    print(f"  {nm:20s} {(s[:, 17 + i] - s[:, 16 + i]).median():8.0f} | {(s[:, 22 + i] - s[:, 21 + i]).median():8.0f}")
