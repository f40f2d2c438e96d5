"""Attention forward (inference form) timed over sequence lengths: what the deal of 16-query blocks to the 8 SIMDs of a
head's two workgroups costs (25 blocks at T = 400 -> some SIMD runs 4 per key tile).
python tools/bench_attn_T.py [T ...]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from css_amd import ops

B, H = 32, 4
dev = "cuda:0"
Ts = [int(x) for x in sys.argv[1:]] or [320, 352, 368, 384, 400, 416, 448, 512]
pe = torch.randn(2000, 64, device=dev) * 0.1
pack = ops.pack_relpos(pe)
for T in Ts:
    qkv = (torch.randn(B, T, 3 * H * 64, device=dev) * 0.5).to(torch.bfloat16)
    for _ in range(3):
        ops.mha_relpos(qkv, pack, H, q_log2e=True)
    torch.cuda.synchronize()
    # 20 launches captured in one graph: back-to-back on the GPU, no host time between them
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.mha_relpos(qkv, pack, H, q_log2e=True)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            ops.mha_relpos(qkv, pack, H, q_log2e=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (n * 20)
    nblk, ntile = (T + 15) // 16, (T + 63) // 64
    print(f"T {T:4d}: {us:6.2f} us   {nblk} query blocks, {ntile} key tiles   us per (block x tile) x 1000: {us / (nblk * ntile) * 1e3:.1f}"
          f"   max blocks per SIMD {-(-nblk // 8)}")
