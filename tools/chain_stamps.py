#!/usr/bin/env python3
"""Diagnostic: phase boundaries (shader cycles, wave 0 of every workgroup) inside chain_kernel.  Needs a library built
with -DP_STAMPS:  tools/panel_ablate.sh build stamps;  python tools/with_lib.py build/abl/libp_stamps.so tools/chain_stamps.py"""
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
FOLD = "--fold" in sys.argv
TRAIN = "--train" in sys.argv     # the training form: dropout 0.1 at all four sites, rows kept for the backward
if FOLD:
    fa, fb = (None,) + fa[1:], (None,) + fb[1:]
run = lambda: ops.encoder_chain(x, 1024, ffn_a=fa, final_ln=(g1, b0), ffn_b=fb, qkv=((None if FOLD else (g1, b0)), wq, bq, 768), out=out, flags=1 if FOLD else 0)
if TRAIN:
    from css_amd.dropout import DropState
    dstate = DropState(torch.tensor([1234], dtype=torch.int32, device="cuda"), 0.1, 0)
    drops = tuple(dstate.cfg(f"0.{t}") for t in ("ffn_out.hidden", "ffn_out.out", "ffn_in.hidden", "ffn_in.out"))
    run = lambda: ops.encoder_chain_train(x, 1024, fb, ((g1, b0), wq, bq, 768), ffn_a=fa, final_ln=(g1, b0), drops=drops)
buf = torch.zeros(256, 32, device="cuda", dtype=torch.int64)
for _ in range(3): run()
torch.cuda.synchronize()
h.csship_debug_stamps(C.c_void_p(buf.data_ptr())); run(); torch.cuda.synchronize(); h.csship_debug_stamps(None)
s = buf[:200].cpu().double()
names = ["x loaded (kernel start -> 0 not measured)", "A: LN -> panel, barrier", "A: 4 x (W1 pass, hidden epilogue, W2 pass)",
         "A: barrier + park + barrier", "A: read rows, residual, final LN", "B: LN -> panel", "B: passes", "B: park",
         "B: read rows + residual", "store x, C: LN -> panel", "C: 3 x (pass, park, rows, store)"]
print("cycles between stamps (median / max over the 200 workgroups)")
for i in range(1, 11):
    d = s[:, i] - s[:, i - 1]
    print(f"  {i:2d} {names[i]:46s} {d.median():8.0f} {d.max():8.0f}")
tot = s[:, 10] - s[:, 0]
print(f"  stamp 0 -> 10: {tot.median():.0f} cycles median")
print("stage A, summed over its 4 hidden chunks (median): wave 0 | wave 7")
for i, nm in enumerate(["W1 pass (8 k-steps)", "hidden epilogue (ReLU, pack, LDS)", "barrier wait", "W2 pass (8 k-steps)"]):
    print(f"  {nm:36s} {s[:, 12 + i].median():8.0f} | {s[:, 16 + i].median():8.0f}")
