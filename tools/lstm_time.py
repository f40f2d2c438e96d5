"""Time the BLSTM recurrence kernels alone: python tools/lstm_time.py [lib.so]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
if len(sys.argv) > 1:
    from css_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from css_amd import ops
B, T, H = 32, 400, 512
g = torch.Generator().manual_seed(0)
gx = (torch.randn((B * T, 8 * H), generator=g) * 0.5).cuda()
whh = torch.cat([ops.pack_frag((torch.randn((4 * H, H), generator=g) / H ** 0.5).cuda()) for _ in range(2)]).contiguous()
for _ in range(3):
    ops.lstm_bidir(gx, whh, B, T, H)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.lstm_bidir(gx, whh, B, T, H)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"lstm_bidir B={B} T={T} H={H}: {ms:.3f} ms = {ms * 1e3 / T:.2f} us per step")
