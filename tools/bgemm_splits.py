"""Split-K sweep for the weight-gradient shapes (slab traffic vs parallelism)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from css_amd import ops_bwd as ob
from bench_kernels import timeit
DEV, BF = "cuda", torch.bfloat16
M = 12800
for (Mo, N) in ((256, 256), (512, 256), (768, 256), (256, 1024), (1024, 256)):
    dy, x = torch.randn(M, Mo, device=DEV), torch.randn(M, N, device=DEV).to(BF)
    res = []
    for s in (0, 6, 8, 12, 16, 24, 32, 48, 64, 96):
        us = timeit(lambda: ob.bgemm(dy, x, a_kmajor=True, b_kmajor=True, splits=s))
        res.append(f"{s}:{us:.1f}")
    print(f"dW [{Mo} x {N}] K=12800  splits:us  " + "  ".join(res))
