#!/usr/bin/env python3
"""The six weight-gradient products of one encoder block (d_model 256, d_ffn 1024, B*T rows): one by one against the
grouped launch (csship_wgrad_hold / csship_wgrad_flush), operands resident in the last-level cache (the loop re-reads the
same 169 MB) - the in-step number (profiles/, bgemm_wgrad_group_kernel) reads them cold.

    python tools/bench_wgrad_group.py [--rows 12800] [--iters 50]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=12800)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    from css_amd import ops_bwd as ob
    dev = torch.device("cuda", 0)
    K = args.rows
    shapes = [(1024, 256), (256, 1024), (1024, 256), (256, 1024), (768, 256), (256, 256)]
    g = torch.Generator(device="cpu").manual_seed(0)
    prods = []
    for M, N in shapes:
        dy = (0.5 * torch.randn(K, M, generator=g)).to(torch.bfloat16).to(dev)
        x = (0.5 * torch.randn(K, N, generator=g)).to(torch.bfloat16).to(dev)
        prods.append((dy, x, torch.empty(M, N, device=dev), torch.empty(M, device=dev)))

    def single():
        for dy, x, dw, db in prods:
            ob.linear_wgrad(dy, x, dw_out=dw, db_out=db)

    def grouped():
        with ob.deferred_reductions(group_wgrads=True):
            for dy, x, dw, db in prods:
                ob.linear_wgrad(dy, x, dw_out=dw, db_out=db)

    flop = sum(2.0 * K * M * N for M, N in shapes)
    for name, fn in (("one by one", single), ("grouped", grouped)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        print(f"{name:12s} {us:8.1f} us per block   {flop / us / 1e6:7.1f} TFLOP/s (products + reductions, eager launches)")


if __name__ == "__main__":
    main()
