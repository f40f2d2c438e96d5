#!/usr/bin/env python3
"""csship_attn_pos_bwd at 32 x 400 frames for 1, 2, 4 and 8 heads: what a head adds to the launch (the kernel walks the
heads of a 64-frame block one after the other)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from css_amd import ops_bwd as ob  # noqa: E402

dev = torch.device("cuda", 0)
B, T = 32, 400
W = (2 * T - 1 + 7) // 8 * 8
for H in (1, 2, 4, 8):
    dsk = (0.1 * torch.randn(B, T, H, W, device=dev)).to(torch.bfloat16)
    win = (0.1 * torch.randn(W, 64, device=dev)).to(torch.bfloat16)
    q = (0.1 * torch.randn(B * T * H, 64, device=dev)).to(torch.bfloat16)
    dq1 = torch.randn(B, H, T, 64, device=dev)
    dqkv = torch.zeros(B, T, 3 * H * 64, dtype=torch.bfloat16, device=dev)
    acc = torch.zeros(W, 64, device=dev)
    wt = ob.pe_window_t(win, T)
    for _ in range(5):
        ob.attn_pos_bwd(dsk, wt, q, dq1, dqkv, acc, H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            ob.attn_pos_bwd(dsk, wt, q, dq1, dqkv, acc, H)
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"H = {H}: {e0.elapsed_time(e1) * 1e3 / 100:7.1f} us per call (both kernels)")
