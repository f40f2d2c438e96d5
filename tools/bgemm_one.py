"""Time one bgemm configuration: python tools/bgemm_one.py M N K akm bkm a_f32 b_f32 [splits...]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from css_amd import ops_bwd as ob

M, N, K, akm, bkm, af, bf = [int(x) for x in sys.argv[1:8]]
splits = [int(x) for x in sys.argv[8:]] or [0]
dev = "cuda"
a = torch.randn((K, M) if akm else (M, K), device=dev, dtype=torch.float32 if af else torch.bfloat16)
b = torch.randn((K, N) if bkm else (N, K), device=dev, dtype=torch.float32 if bf else torch.bfloat16)


def launch_time(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for s in splits:
    us = launch_time(lambda: ob.bgemm(a, b, a_kmajor=bool(akm), b_kmajor=bool(bkm), splits=s))
    print(f"M={M} N={N} K={K} akm={akm} bkm={bkm} f32={af}{bf} splits={s}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s")
