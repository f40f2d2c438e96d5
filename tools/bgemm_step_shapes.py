"""The batched-GEMM launches of one Conformer-base training step (per encoder block), timed one shape at a time by graph
replay: where the 4.5 ms of bgemm time per step goes."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from css_amd import ops_bwd as ob
from bench_kernels import timeit
DEV, BF = "cuda", torch.bfloat16
B, H, T, D, FF = 32, 4, 400, 256, 1024
M = B * T
f32 = lambda *s: torch.randn(*s, device=DEV)
bf = lambda *s: torch.randn(*s, device=DEV).to(BF)
rows = []
def add(name, flop, fn, n=1):
    us = timeit(fn)
    rows.append((name, n, us, flop / us / 1e6))
dy, a_bf, h_bf, dh_bf = f32(M, D), bf(M, D), bf(M, FF), bf(M, FF)
add("wgrad dW2 [256 x 1024], K=12800 (+db)", 2.0 * M * D * FF, lambda: ob.linear_wgrad(dy, h_bf, 0.5))
add("wgrad dW1 [1024 x 256], K=12800 (+db)", 2.0 * M * D * FF, lambda: ob.linear_wgrad(dh_bf, a_bf))
dq2 = f32(M, 3 * D)
add("wgrad dWq [256 x 256] (+db)", 2.0 * M * D * D, lambda: ob.linear_wgrad(dq2[:, :D], a_bf, 0.125))
add("wgrad dWkv [512 x 256] (+db)", 4.0 * M * D * D, lambda: ob.linear_wgrad(dq2[:, D:], a_bf))
ao = bf(M, D)
add("wgrad dWo [256 x 256] (+db)", 2.0 * M * D * D, lambda: ob.linear_wgrad(dy, ao))
wo_bf, wqkv_bf = bf(D, D), bf(3 * D, D)
add("dgrad dao = dy Wo [12800 x 256], K=256 -> bf16", 2.0 * M * D * D, lambda: ob.bgemm(dy, wo_bf, b_kmajor=True, out_bf16=True))
add("dgrad da = dqkv Wqkv [12800 x 256], K=768", 2.0 * M * D * 3 * D, lambda: ob.bgemm(dq2, wqkv_bf, b_kmajor=True))
Tp = (T + 7) // 8 * 8
p, ds = bf(B, H, T, Tp), bf(B, H, T, Tp)
qkv = bf(B, T, 3 * D)
q = qkv[..., :D].view(B, T, H, 64).permute(0, 2, 1, 3)
k = qkv[..., D:2 * D].view(B, T, H, 64).permute(0, 2, 1, 3)
do = bf(B, T, D).view(B, T, H, 64).permute(0, 2, 1, 3)
dv, dk = torch.empty(B, T, H, 64, device=DEV).permute(0, 2, 1, 3), torch.empty(B, T, H, 64, device=DEV).permute(0, 2, 1, 3)
fl = 2.0 * B * H * T * T * 64
add("attn dV = P^T dO  [128 x (400 x 64)], K=400", fl, lambda: ob.bgemm(p[..., :T], do, a_kmajor=True, b_kmajor=True, out=dv))
add("attn dK = dS^T q", fl, lambda: ob.bgemm(ds[..., :T], q, a_kmajor=True, b_kmajor=True, out=dk))
add("attn dq1 = dS k", fl, lambda: ob.bgemm(ds[..., :T], k, b_kmajor=True))
W = 800
dsk, win = bf(B, T, H, W), bf(W, 64)
dq1, dq = f32(B, H, T, 64), torch.empty(B, T, H, 64, device=DEV).permute(0, 2, 1, 3)
add("attn dq = dS' win + dq1  [.. x (400 x 64)], K=800", 2.0 * B * H * T * W * 64, lambda: ob.bgemm(dsk.permute(0, 2, 1, 3), win[None, None], b_kmajor=True, add=dq1, out=dq))
q_rows = qkv[..., :D].reshape(B * T * H, 64)
add("attn dwin = dS'^T q  [800 x 64], K=51200", 2.0 * B * T * H * W * 64, lambda: ob.bgemm(dsk.view(B * T * H, W), q_rows, a_kmajor=True, b_kmajor=True))
tot = 0.0
for name, n, us, tf in rows:
    tot += n * us
    print(f"{name:58s} {us:7.1f} us {tf:7.1f} TFLOP/s")
print(f"sum per encoder block: {tot:.0f} us -> x16 = {tot * 16 / 1e3:.2f} ms")
