#!/usr/bin/env python3
"""Per-kernel micro-benchmarks at the headline shapes (B=32, T=400, d=256): optimisation harness.

Usage: python tools/bench_kernels.py [names...]   (ffn, linear, attn, conv, stft, pit, ln)
Each line: kernel, shape, average microseconds over REPS back-to-back launches (HIP events on
the launch stream), achieved TFLOP/s or GB/s.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import ops  # noqa: E402
from css_amd.utils import synthetic as syn  # noqa: E402

DEV = "cuda"
REPS = int(os.environ.get("REPS", "30"))
B, T, D = 32, 400, 256
M = B * T


def timeit(fn, reps=REPS, warm=3):
    """Average microseconds per call: `reps` calls captured into one HIP graph, the replay bracketed by HIP events on
    the launch stream (an eager Python loop around a 10-30 us kernel measures ctypes + the allocator, not the kernel)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def bench_ffn():
    x = torch.randn(M, D, device=DEV)
    g, b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    for ff in (256, 1024, 2048):
        w1 = ops.pack_frag(torch.randn(ff, D, device=DEV) / 16)
        w2 = ops.pack_frag(torch.randn(D, ff, device=DEV) / 32)
        b1, b2 = torch.zeros(ff, device=DEV), torch.zeros(D, device=DEV)
        out = torch.empty_like(x)
        for fin in (False, True):
            us = timeit(lambda: ops.ffn_fused(x, (g, b), w1, b1, w2, b2, ff, 0.5, final_ln=(g, b) if fin else None, out=out))
            print(f"ffn_fused d_ffn={ff:5d} final_ln={int(fin)}  {us:8.1f} us  {4.0 * M * D * ff / us / 1e6:8.1f} TFLOP/s")


def bench_ffn_bwd():
    """the fused FeedForward backward (activation side) alone, with and without the reference's dropouts"""
    from css_amd import ops_bwd as ob
    from css_amd.dropout import DropState
    x, dy = torch.randn(M, D, device=DEV), torch.randn(M, D, device=DEV)
    g, b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    ff = 1024
    w1, w2 = torch.randn(ff, D, device=DEV) / 16, torch.randn(D, ff, device=DEV) / 32
    w1p, w2tp, w1tp = ops.pack_frag(w1), ops.pack_frag(w2.t().contiguous()), ops.pack_frag(w1.t().contiguous())
    b1 = torch.zeros(ff, device=DEV)
    st = DropState(torch.tensor([5], dtype=torch.int32, device=DEV), 0.1, 0)
    for drop in (False, True):
        dh_, do_ = (st.cfg("0.ffn_in.hidden"), st.cfg("0.ffn_in.out")) if drop else (None, None)
        us = timeit(lambda: ob.ffn_bwd(x, dy, g, b, w1p, b1, w2tp, w1tp, ff, 0.5, drop_hidden=dh_, drop_out=do_))
        print(f"ffn_bwd d_ffn={ff} dropout={int(drop)}  {us:8.1f} us  {6.0 * M * D * ff / us / 1e6:8.1f} TFLOP/s")


def bench_linear():
    x = torch.randn(M, D, device=DEV)
    xb = x.to(torch.bfloat16)
    g, b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    for N in (256, 768, 1536, 771):
        w = ops.pack_frag(torch.randn(N, D, device=DEV) / 16)
        bias = torch.zeros(N, device=DEV)
        ob = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        of = torch.empty(M, N, device=DEV)
        us = timeit(lambda: ops.panel_linear(x, w, bias, N, ln=(g, b), out_bf16=True, out=ob))
        print(f"panel_linear LN->bf16 N={N:5d}  {us:8.1f} us  {2.0 * M * D * N / us / 1e6:8.1f} TFLOP/s")
        us = timeit(lambda: ops.panel_linear(x, w, bias, N, relu=True, out=of))
        print(f"panel_linear f32->relu f32 N={N:5d}  {us:8.1f} us  {2.0 * M * D * N / us / 1e6:8.1f} TFLOP/s")
        if N == 256:
            res = x.clone()
            us = timeit(lambda: ops.panel_linear(xb, w, bias, N, residual=res, alpha=1.0, out=res))
            print(f"panel_linear bf16->+res N={N:5d}  {us:8.1f} us  {2.0 * M * D * N / us / 1e6:8.1f} TFLOP/s")


def bench_attn():
    H = 4
    pe = ops.pack_relpos(0.3 * torch.randn(2000, 64, device=DEV))
    shapes = ((32, 400), (32, 150), (8, 1600))
    if os.environ.get('ATTN_SHAPE'):
        shapes = (tuple(int(v) for v in os.environ['ATTN_SHAPE'].split('x')),)
    for (b_, t_) in shapes:
        qkv = torch.randn(b_, t_, 3 * H * 64, device=DEV)
        qkv[..., :256] *= 0.125
        qkv = qkv.to(torch.bfloat16)
        us = timeit(lambda: ops.mha_relpos(qkv, pe, H))
        fl = b_ * H * (2.0 * t_ * t_ * 64 * 2 + 2.0 * t_ * t_ * 64)  # QK^T + Q.PE^T + PV over full T x T
        print(f"mha_relpos B={b_} T={t_}  {us:8.1f} us  {fl / us / 1e6:8.1f} TFLOP/s (algorithmic)")


def bench_attn_bwd():
    """the two attention-shaped kernels of the backward pass: probability recompute and softmax backward + skewed copy"""
    from css_amd import ops_bwd as ob
    H, B_, T_ = 4, 32, 400
    pe = ops.pack_relpos(0.3 * torch.randn(2000, 64, device=DEV))
    qkv = torch.randn(B_, T_, 3 * H * 64, device=DEV)
    qkv[..., :256] *= 0.125
    qkv = qkv.to(torch.bfloat16)
    ao, lse = ops.mha_relpos_train(qkv, pe, H)
    us = timeit(lambda: ops.mha_relpos_probs(qkv, pe, lse, H))
    print(f"mha_relpos_probs B={B_} T={T_}  {us:8.1f} us")
    p = ops.mha_relpos_probs(qkv, pe, lse, H)
    dao = torch.randn(B_, T_, H * 64, device=DEV).to(torch.bfloat16)
    d = ob.attn_rowdot(dao, ao, B_, T_, H)
    skew = torch.zeros(B_, T_, H, 800, device=DEV, dtype=torch.bfloat16)
    us = timeit(lambda: ob.attn_softmax_bwd(qkv, dao, p, d, skew, H))
    print(f"attn_softmax_bwd (dS + skewed copy) B={B_} T={T_}  {us:8.1f} us")


def bench_chain():
    """The dominant kernel of the step: FFN_out + block LayerNorm -> FFN_in -> LayerNorm + q/k/v (conformer.py:322-326,355-360,222-225)."""
    x = torch.randn(M, D, device=DEV)
    g1, b0 = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)

    def ffn_w():
        return ((g1, b0), ops.pack_frag(torch.randn(1024, D, device=DEV) / 16), torch.zeros(1024, device=DEV),
                ops.pack_frag(torch.randn(D, 1024, device=DEV) / 32), torch.zeros(D, device=DEV))
    fa, fb = ffn_w(), ffn_w()
    wq, bq = ops.pack_frag(torch.randn(768, D, device=DEV) / 16), torch.zeros(768, device=DEV)
    out = torch.empty_like(x)
    us = timeit(lambda: ops.encoder_chain(x, 1024, ffn_a=fa, final_ln=(g1, b0), ffn_b=fb, qkv=((g1, b0), wq, bq, 768), out=out))
    fl = 2 * 4.0 * M * D * 1024 + 2.0 * M * D * 768
    print(f"encoder_chain (FFN+LN, FFN, LN+QKV)  {us:8.1f} us  {fl / us / 1e6:8.1f} TFLOP/s = {fl / us / 1e6 / 2500:.3f} of 2.5 PF")
    # the form the model runs: LayerNorm affines folded into the weights, q | k | v rows interleaved (16-byte stores)
    fa2, fb2 = (None,) + fa[1:], (None,) + fb[1:]
    us = timeit(lambda: ops.encoder_chain(x, 1024, ffn_a=fa2, final_ln=(g1, b0), ffn_b=fb2, qkv=(None, wq, bq, 768), out=out, flags=1))
    print(f"encoder_chain, folded LN + IL8       {us:8.1f} us  {fl / us / 1e6:8.1f} TFLOP/s = {fl / us / 1e6 / 2500:.3f} of 2.5 PF")
    us = timeit(lambda: ops.encoder_chain(x, 1024, ffn_a=fa, final_ln=(g1, b0), out=out))
    print(f"encoder_chain (FFN+LN only)          {us:8.1f} us  {4.0 * M * D * 1024 / us / 1e6:8.1f} TFLOP/s")


def bench_conv():
    full = syn.conformer_shapes(num_blocks=1)
    pre = "conformer.encoders.0.conv."
    sd = syn.fill_state_dict({k[len(pre):]: s for k, s in full.items() if k.startswith(pre)}, 1)
    w = {"ln_g": sd["layer_norm.weight"], "ln_b": sd["layer_norm.bias"], "pw1_w": sd["pw_conv_1.weight"].reshape(2),
         "pw1_b": sd["pw_conv_1.bias"], "dw_w": sd["dw_conv_1d.weight"].reshape(256, 33).t(), "dw_b": sd["dw_conv_1d.bias"],
         "bn_g": sd["BN.weight"], "bn_b": sd["BN.bias"], "bn_mean": sd["BN.running_mean"], "bn_var": sd["BN.running_var"],
         "pw2_w": sd["pw_conv_2.weight"].reshape(1), "pw2_b": sd["pw_conv_2.bias"]}
    w = {k: v.contiguous().to(DEV) for k, v in w.items()}
    x = torch.randn(B, T, D, device=DEV)
    for bs in (False, True):
        us = timeit(lambda: ops.convmod(x, w, 33, batch_stats=bs))
        print(f"convmod batch_stats={int(bs)}  {us:8.1f} us  {2 * M * D * 4 / us / 1e3:8.1f} GB/s (x in + y out)")


def bench_conv_bwd():
    """the conv module's backward (csship_convmod_bwd_next: statistics pass + reduce + the fused tile kernel, or the four
    launches with CSSHIP_CONV_BWD_UNFUSED=1), with the reference's dropout sites on"""
    from css_amd import ops_bwd as ob
    from css_amd.dropout import DropState
    full = syn.conformer_shapes(num_blocks=1)
    pre = "conformer.encoders.0.conv."
    sd = syn.fill_state_dict({k[len(pre):]: s for k, s in full.items() if k.startswith(pre)}, 1)
    w = {"ln_g": sd["layer_norm.weight"], "ln_b": sd["layer_norm.bias"], "pw1_w": sd["pw_conv_1.weight"].reshape(2),
         "pw1_b": sd["pw_conv_1.bias"], "dw_w": sd["dw_conv_1d.weight"].reshape(256, 33).t(), "dw_b": sd["dw_conv_1d.bias"],
         "bn_g": sd["BN.weight"], "bn_b": sd["BN.bias"], "bn_mean": sd["BN.running_mean"], "bn_var": sd["BN.running_var"],
         "pw2_w": sd["pw_conv_2.weight"].reshape(1), "pw2_b": sd["pw_conv_2.bias"]}
    w = {k: v.contiguous().to(DEV) for k, v in w.items()}
    x = torch.randn(B, T, D, device=DEV)
    dy = torch.randn(B, T, D, device=DEV) * 0.1
    st = DropState(torch.tensor([5], dtype=torch.int32, device=DEV), 0.1, 0)
    c_out, c_next = st.cfg("0.conv.out"), st.cfg("0.attn.out")
    _, fws = ops.convmod(x, w, 33, batch_stats=True, return_workspace=True, drop=c_out)
    us = timeit(lambda: ob.convmod_bwd(x, dy, w, fws, 33, drop=c_out, want_next=True, next_drop=c_next))
    print(f"convmod_bwd (+ next operand)  {us:8.1f} us  {M * D * (4 * 4 + 2) / us / 1e3:8.1f} GB/s algorithmic (x, dy, z in; dx + bf16 operand out)")


def bench_stft():
    n = 512 + 256 * (T - 1)
    a = 0.1 * torch.randn(3, B, n, device=DEV)
    us = timeit(lambda: ops.stft(a, ("mag",)))
    frames = 3 * B * T
    print(f"stft mag 3x{B}x{n}  {us:8.1f} us  {frames * 2052 / us / 1e3:8.1f} GB/s algorithmic ({frames / us:.1f} Mframes/s)")


def bench_pit():
    F = 257
    masks = torch.rand(B, T, 3 * F, device=DEV)
    f = torch.rand(B, T, F, device=DEV)
    tg = [torch.rand(B, T, F, device=DEV) for _ in range(2)]
    us = timeit(lambda: ops.pit_mse(masks, f, tg))
    print(f"pit_mse K=2  {us:8.1f} us  {M * 5 * F * 4 / us / 1e3:8.1f} GB/s algorithmic (2 masks + f + 2 targets)")


def bench_ln():
    x = torch.randn(M, D, device=DEV)
    g, b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    us = timeit(lambda: ops.layernorm(x, g, b))
    print(f"layernorm  {us:8.1f} us  {M * D * 6 / us / 1e3:8.1f} GB/s")


ALL = {"chain": bench_chain, "ffn": bench_ffn, "ffn_bwd": bench_ffn_bwd, "linear": bench_linear, "attn": bench_attn, "attn_bwd": bench_attn_bwd, "conv": bench_conv, "conv_bwd": bench_conv_bwd, "stft": bench_stft,
       "pit": bench_pit, "ln": bench_ln}

if __name__ == "__main__":
    names = sys.argv[1:] or list(ALL)
    torch.cuda.set_device(0)
    for nm in names:
        ALL[nm]()
