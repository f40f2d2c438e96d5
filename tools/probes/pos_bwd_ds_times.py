#!/usr/bin/env python3
"""csship_attn_pos_bwd (skewed dS in memory) against csship_attn_pos_bwd_ds (row-major dS skewed while staging) at 32 x 4 x 400."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from css_amd import ops_bwd as ob  # noqa: E402

dev = torch.device("cuda", 0)
B, T, H = 32, 400, 4
W = (2 * T - 1 + 7) // 8 * 8
dsk = (0.1 * torch.randn(B, T, H, W, device=dev)).to(torch.bfloat16)
ds = ob.slack_bf16((B, H, T, T), dev)
ds.copy_((0.1 * torch.randn(B, H, T, T, device=dev)).to(torch.bfloat16))
win = (0.1 * torch.randn(W, 64, device=dev)).to(torch.bfloat16)
q = (0.1 * torch.randn(B * T * H, 64, device=dev)).to(torch.bfloat16)
dq1 = torch.randn(B, H, T, 64, device=dev)
dqkv = torch.zeros(B, T, 3 * H * 64, dtype=torch.bfloat16, device=dev)
acc = torch.zeros(W, 64, device=dev)
wt = ob.pe_window_t(win, T)
for name, fn in (("skewed dS", lambda: ob.attn_pos_bwd(dsk, wt, q, dq1, dqkv, acc, H)),
                 ("row-major dS", lambda: ob.attn_pos_bwd_ds(ds, wt, q, dq1, dqkv, acc, H))):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:14s} {e0.elapsed_time(e1) * 1e3 / 100:7.1f} us per call (both kernels)")
