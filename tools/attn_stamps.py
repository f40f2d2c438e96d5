#!/usr/bin/env python3
"""Diagnostic: per-wave cycle sums of the phases of mha_fwd_kernel's key-tile loop (needs a library built with
-DA2_STAMPS: tools/attn_ablate.sh build stamps; run: python tools/with_lib.py build/abl/lib_stamps.so tools/attn_stamps.py)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import ops, _lib
lib = _lib.lib(); h = C.CDLL(_lib.LIB_PATH); h.csship_debug_stamps.argtypes = [C.c_void_p]
B, T = int(os.environ.get("ATTN_B", "32")), 400
pe = ops.pack_relpos(0.3 * torch.randn(2000, 64, device="cuda"))
qkv = torch.randn(B, T, 768, device="cuda"); qkv[..., :256] *= 0.125; qkv = qkv.to(torch.bfloat16)
buf = torch.zeros(4096, 8, device="cuda", dtype=torch.int64)
for _ in range(3): ops.mha_relpos(qkv, pe, 4)
torch.cuda.synchronize()
h.csship_debug_stamps(C.c_void_p(buf.data_ptr())); ops.mha_relpos(qkv, pe, 4); torch.cuda.synchronize(); h.csship_debug_stamps(None)
NWV = int(os.environ.get("ATTN_NW", "8")); s = buf[:256 * NWV].cpu().double().view(256, NWV, 8)
names = ["S (K reads + 16 MFMA)", "max / rescale", "exp + band(j+1)", "P.V", "staging stores", "barrier"]
print("cycles per wave summed over the 7 key tiles (median over workgroups; waves 0-3 | waves 4-7)")
for i, nm in enumerate(names):
    a, b = s[:, :4, i].median(), s[:, NWV - 4:, i].median()
    print(f"  {nm:26s} {a:9.0f} | {b:9.0f}")
tot = s[:, :, :6].sum(-1)
print(f"  total {tot[:, :4].median():.0f} | {tot[:, NWV - 4:].median():.0f}")
