#!/usr/bin/env python3
"""Diagnostic: phase stamps inside convmod and attention kernels."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import ops, _lib
from css_amd.utils import synthetic as syn
lib = _lib.lib(); h = C.CDLL(_lib.LIB_PATH); h.csship_debug_stamps.argtypes = [C.c_void_p]
DEV = "cuda"; B, T, D = 32, 400, 256
buf = torch.zeros(4096, 8, device=DEV, dtype=torch.int64)
def report(name, fn, nph, nwg):
    for _ in range(3): fn()
    torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    h.csship_debug_stamps(C.c_void_p(buf.data_ptr())); fn(); torch.cuda.synchronize(); h.csship_debug_stamps(None)
    s = buf[:nwg].cpu().double() * 10.0
    t0 = s[:, 0].min(); print(name)
    st = s[:, 0] - t0
    print("  wg start ns: med %.0f  p90 %.0f max %.0f" % (st.median(), st.quantile(0.9), st.max()))
    for i in range(1, nph):
        d = s[:, i] - s[:, i-1]; print("  phase %d-%d ns: min %.0f med %.0f max %.0f" % (i-1, i, d.min(), d.median(), d.max()))
    print("  total: %.0f ns" % (s[:, :nph].max() - t0))
full = syn.conformer_shapes(num_blocks=1); pre = "conformer.encoders.0.conv."
sd = syn.fill_state_dict({k[len(pre):]: v for k, v in full.items() if k.startswith(pre)}, 1)
w = {"ln_g": sd["layer_norm.weight"], "ln_b": sd["layer_norm.bias"], "pw1_w": sd["pw_conv_1.weight"].reshape(2),
     "pw1_b": sd["pw_conv_1.bias"], "dw_w": sd["dw_conv_1d.weight"].reshape(256, 33).t(), "dw_b": sd["dw_conv_1d.bias"],
     "bn_g": sd["BN.weight"], "bn_b": sd["BN.bias"], "bn_mean": sd["BN.running_mean"], "bn_var": sd["BN.running_var"],
     "pw2_w": sd["pw_conv_2.weight"].reshape(1), "pw2_b": sd["pw_conv_2.bias"]}
w = {k: v.contiguous().to(DEV) for k, v in w.items()}
x = torch.randn(B, T, D, device=DEV)
report("convmod (0 start, 1 LN+GLU parked, 2 weights loaded, 3 done)", lambda: ops.convmod(x, w, 33), 4, 13 * 32)
pe = ops.pack_relpos(0.3 * torch.randn(2000, 64, device=DEV))
qkv = torch.randn(B, T, 768, device=DEV); qkv[..., :256] *= 0.125; qkv = qkv.to(torch.bfloat16)
report("mha_relpos (0 start, 1 first tile staged, 2 loop done)", lambda: ops.mha_relpos(qkv, pe, 4), 3, 7 * 4 * 32)
