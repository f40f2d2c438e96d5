"""Time csship_bgemm_wgrad (product + slab reduction) for the training step's weight-gradient shapes over split counts.
tools/wgrad_sweep.py [--sets 8]   (CSSHIP_BGEMM_NOGLDS=1 in the environment selects the register-staged loop)"""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from css_amd import _lib
from css_amd.ops import _ptr, _stream, check, lib

ap = argparse.ArgumentParser()
ap.add_argument("--sets", type=int, default=8)
ap.add_argument("--iters", type=int, default=48)
ap.add_argument("--splits", default="0,8,12,16,21,24,32,48,64")
a = ap.parse_args()
dev = "cuda"
K = 12800
for M, N in ((256, 1024), (1024, 256), (768, 256), (256, 256), (512, 256)):
    dys = [torch.randn(K, M, device=dev).to(torch.bfloat16) for _ in range(a.sets)]
    xs = [torch.randn(K, N, device=dev).to(torch.bfloat16) for _ in range(a.sets)]
    dw = torch.empty(M, N, device=dev)
    db = torch.empty(M, device=dev)
    line = []
    for s in [int(v) for v in a.splits.split(",")]:
        p = _lib.BgemmParams(M=M, N=N, K=K, lda=M, ldb=N, ldc=N, ldaux=0, a_kmajor=1, b_kmajor=1, a_f32=0, b_f32=0, epilogue=0,
                             batch1=1, batch2=1, splits=s, sA1=0, sA2=0, sB1=0, sB2=0, sC1=0, sC2=0, sX1=0, sX2=0, alpha=1.0)
        nws = lib().csship_bgemm_wgrad_workspace(C.byref(p))
        ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
        def run(i):
            check(lib().csship_bgemm_wgrad(_ptr(dys[i % a.sets]), _ptr(xs[i % a.sets]), _ptr(dw), _ptr(db), None, C.byref(p), 0.0, 0,
                                           _ptr(ws), nws, _stream()))
        for i in range(8):
            run(i)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()   # one graph of all calls: kernel time, not the host's launch rate
        with torch.cuda.graph(gr):
            for i in range(a.iters):
                run(i)
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        line.append("s=%d: %.1f" % (s, e0.elapsed_time(e1) * 1e3 / a.iters))
    print("M %4d N %4d  us per call  " % (M, N) + "  ".join(line), flush=True)
