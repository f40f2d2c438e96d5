"""Forward + PIT timing of the Conformer-large shape class (BASELINE.json configs[4]: d_model 512, 8 heads, ffn 2048,
18 blocks, 3 speakers + noise, 32 x 400 frames) through the any-width path, in its three FeedForward forms:
  unfused  LayerNorm + two tiled Linears (round 2)           bf16
  wide     csship_ffn_wide, one kernel per FeedForward        bf16   (the parity path)
  mxfp8    csship_ffn_mx, block-scaled fp8 operands           OPT-IN (model.fp8_ffn = True; oracle/mxfp8.py)
plus the FeedForward kernels alone (M = 12 800 rows) against the dense MFMA peak of their operand type.
Usage: python tools/bench_large.py [--blocks 18] [--base]   (--base: the same three forms at d_model 256 / ffn 1024)"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from css_amd import ops
from css_amd.models import MODELS
from css_amd.utils import synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=18)
ap.add_argument("--base", action="store_true")
a = ap.parse_args()
D, H, FF, K = (256, 4, 1024, 2) if a.base else (512, 8, 2048, 3)
conf = dict(attention_dim=D, attention_heads=H, linear_units=FF, num_blocks=a.blocks, kernel_size=33, dropout_rate=0.1,
            relative_pos_emb=True)
sd = syn.fill_state_dict(syn.conformer_shapes(num_spk=K, attention_dim=D, attention_heads=H, linear_units=FF, num_blocks=a.blocks), 0)
B, T = 32, 400
f = torch.rand(B, T, 257, device="cuda")
tg = [torch.rand(B, T, 257, device="cuda") * 0.5 for _ in range(K)]
flop = B * T * (52.57e6 * a.blocks / 16 if a.base else 212.8e6 * a.blocks / 18)


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


res, masks_ref = {}, None
for name, attrs in (("unfused", dict(force_generic=True, unfused_ffn=True)), ("wide", dict(force_generic=True)),
                    ("mxfp8", dict(fp8_ffn=True))):
    m = MODELS["Conformer"](257, 257, K, 1, conf)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    for k, v in attrs.items():
        setattr(m, k, v)
    with torch.no_grad():
        out = {}
        def step():
            out["m"] = m.forward_masks(f); out["l"] = ops.pit_mse(out["m"], f, tg)[0]
        dt = timed(step)
    if masks_ref is None:
        masks_ref = out["m"].clone()
    res[name] = {"ms_per_step": dt * 1e3, "frames_per_s": B * T / dt, "algorithmic_TFLOPs": flop / dt / 1e12,
                 "frac_of_bf16_peak": flop / dt / 2.5e15, "loss": float(out["l"]),
                 "mask_mse_vs_unfused_bf16": float(((out["m"] - masks_ref) ** 2).mean())}
    print(f"{name:8s} {dt * 1e3:7.3f} ms/step = {B * T / dt / 1e6:.3f} M frames/s, {flop / dt / 1e12:6.0f} TFLOP/s algorithmic "
          f"({flop / dt / 2.5e15:.3f} of the bf16 peak), mask MSE vs unfused bf16 {res[name]['mask_mse_vs_unfused_bf16']:.2e}", flush=True)
    del m

# the FeedForward kernels alone
M = B * T
x = torch.randn(M, D, device="cuda")
g1, b0 = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
w1, w2 = torch.randn(FF, D, device="cuda") * D ** -0.5, torch.randn(D, FF, device="cuda") * FF ** -0.5
b1, b2 = torch.zeros(FF, device="cuda"), torch.zeros(D, device="cuda")
o = torch.empty_like(x)
fl = 4.0 * M * D * FF
w1p, w2p, w1m, w2m = ops.pack_frag(w1), ops.pack_frag(w2), ops.pack_mx(w1), ops.pack_mx(w2)
kern = {}
t = timed(lambda: ops.ffn_wide(x, (g1, b0), w1p, b1, w2p, b2, FF, 0.5, out=o), 50, 5)
kern["ffn_wide_bf16"] = {"us": t * 1e6, "TFLOPs": fl / t / 1e12, "frac_of_peak": fl / t / 2.5e15, "peak": "2.5 PF bf16 dense"}
t = timed(lambda: ops.ffn_mx(x, (g1, b0), w1m, b1, w2m, b2, FF, 0.5, out=o), 50, 5)
kern["ffn_mxfp8"] = {"us": t * 1e6, "TFLOPs": fl / t / 1e12, "frac_of_peak": fl / t / 5.0e15, "peak": "5 PF block-scaled fp8 dense"}
if D == 256:
    t = timed(lambda: ops.ffn_fused(x, (g1, b0), w1p, b1, w2p, b2, FF, 0.5, out=o), 50, 5)
    kern["ffn_fused_bf16 (panel.hip)"] = {"us": t * 1e6, "TFLOPs": fl / t / 1e12, "frac_of_peak": fl / t / 2.5e15, "peak": "2.5 PF bf16 dense"}
for k, v in kern.items():
    print(f"{k:28s} {v['us']:8.1f} us  {v['TFLOPs']:7.0f} TFLOP/s = {v['frac_of_peak']:.3f} of {v['peak']}")
print(json.dumps({"metric": "STFT-frames/sec (fwd+PIT) Conformer-" + ("base, any-width path" if a.base else "large"),
                  "config": {"workload": f"d_model {D}, {H} heads, ffn {FF}, {a.blocks} blocks, {K} spk + noise, 32x400 frames, eval forward + PIT"},
                  "variants": res, "ffn_kernels_M12800": kern}))
