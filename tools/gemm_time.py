"""Throughput of the tiled forward GEMM (csship_gemm) on the Conformer-large / BLSTM shapes."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
if len(sys.argv) > 1:
    from css_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from css_amd import ops

dev = "cuda"
M = 12800


def launch_time(fn, reps=10):
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


for K, N, kind in ((512, 2048, "bf16 relu->bf16"), (2048, 512, "bf16 +res f32"), (512, 1536, "bf16 ->bf16"), (512, 512, "bf16 +res f32"),
                   (1024, 4096, "f32 ->f32"), (1024, 771, "f32 relu f32")):
    w = ops.pad_weight_bf16(torch.randn(N, K, device=dev) / K ** 0.5)
    bias = torch.zeros(N, device=dev)
    if kind.startswith("bf16"):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    else:
        a = torch.randn(M, K, device=dev)
    if "+res" in kind:
        res = torch.randn(M, N, device=dev)
        out = torch.empty(M, N, device=dev)
        fn = lambda: ops.gemm(a, w, bias, K=K, residual=res, alpha=0.5, out=out)
    elif kind.endswith("->bf16"):
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: ops.gemm(a, w, bias, K=K, relu="relu" in kind, out_bf16=True, out=out)
    else:
        out = torch.empty(M, N, device=dev)
        fn = lambda: ops.gemm(a, w, bias, K=K, relu="relu" in kind, out=out)
    us = launch_time(fn)
    print(f"gemm M={M} K={K:5d} N={N:5d} {kind:18s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
