#!/usr/bin/env python3
"""How fast does the LDS-DMA weight-gradient loop (bgemm_kernel<..., GL>) run per 128 x 128 x 64 tile step, as a function
of how many workgroups run at once?  One product dW[256, 256] = dY^T X over K rows, its reduction cut into ``splits``
pieces: 4 tiles x splits workgroups, K / 64 / splits tile steps each.  Few workgroups = the loop's own rate (LDS, issue,
latency); 256 = the rate with every CU pulling operands through L2 / fabric at once.

    python tools/probes/wgrad_rate.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402


def main():
    from css_amd import _lib
    from css_amd.ops_bwd import lib, _ptr, _stream, check
    dev = torch.device("cuda", 0)
    K, M, N = 204800, 256, 256
    g = torch.Generator().manual_seed(0)
    dy = (0.5 * torch.randn(K, M, generator=g)).to(torch.bfloat16).to(dev)
    x = (0.5 * torch.randn(K, N, generator=g)).to(torch.bfloat16).to(dev)
    dw = torch.empty(M, N, device=dev)
    db = torch.empty(M, device=dev)
    for splits in (1, 2, 4, 8, 16, 32, 64):
        p = _lib.BgemmParams(M=M, N=N, K=K, lda=M, ldb=N, ldc=N, ldaux=0, a_kmajor=1, b_kmajor=1, a_f32=0, b_f32=0, epilogue=0,
                             batch1=1, batch2=1, splits=splits, sA1=0, sA2=0, sB1=0, sB2=0, sC1=0, sC2=0, sX1=0, sX2=0, alpha=1.0)
        nws = max(lib().csship_bgemm_wgrad_workspace(C.byref(p)), 16)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)

        def run():
            check(lib().csship_bgemm_wgrad(_ptr(dy), _ptr(x), _ptr(dw), _ptr(db), None, C.byref(p), 0.0, 0, _ptr(ws), nws, _stream()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        steps = K / 64 / splits
        print(f"splits {splits:3d}: {4 * splits:4d} workgroups x {steps:7.1f} tile steps   {us:9.1f} us   {us / steps * 1e3:7.1f} ns per tile step"
              f"   {2.0 * K * M * N / us / 1e6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
