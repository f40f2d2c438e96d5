"""Stand-alone timing of csship_attn_dvdk (dV + dK of the attention in one pass over P' and dS) at the training shape.
python tools/bench_dvdk.py [B H T]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from css_amd import _lib
if "--lib" in sys.argv:   # another build of the library (e.g. a -DDV_PROBE variant of attnbwd.hip)
    i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
from css_amd import ops_bwd as ob

ldp = None
if "--ldp" in sys.argv:   # wider rows than the default round-up to 8
    i = sys.argv.index("--ldp")
    ldp = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
B, H, T = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 4, 400)
ldp = ldp or (T + 7) // 8 * 8
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
sets = []
for _ in range(4):   # 4 operand sets (4 x 82 MB at the default shape): more than the L2s, inside the Infinity Cache
    p = torch.rand(B, H, T, ldp, device=dev, generator=g).to(torch.bfloat16)
    ds = (torch.rand(B, H, T, ldp, device=dev, generator=g) - 0.5).to(torch.bfloat16)
    p[..., T:] = 0
    ds[..., T:] = 0
    sets.append((p, ds))
qkv = torch.randn(B, T, 3 * H * 64, device=dev, generator=g).to(torch.bfloat16)
dao = torch.randn(B * T, H * 64, device=dev, generator=g).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
for p, ds in sets:
    ob.attn_dvdk(p, ds, dao, qkv, dqkv, H)
torch.cuda.synchronize()
n = 200
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    p, ds = sets[i & 3]
    ob.attn_dvdk(p, ds, dao, qkv, dqkv, H)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
byts = 2 * B * H * T * T * 2
print(f"attn_dvdk {B}x{H}x{T} ld {ldp}: {us:.2f} us per launch = {byts / us * 1e-6:.2f} TB/s of the two [B,H,T,T] operands")
