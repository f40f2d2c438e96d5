"""Where does the one-sided 7.8e-4 loss difference of the benchmark batch come from?  

The bench line's `loss_parity` has read HIP 0.68928 vs oracle 0.68875 for three rounds: same sign, same size.  This tool
evaluates the SAME batch (bench.py's seed-0 model, rank-0 mixtures, 32 x 400 frames, 16 blocks, eval mode)
  * through the HIP path in its packing variants (LayerNorm folding, log2e folded into q, the any-width kernels), and
    through the materialised-mask route (head kernel + pit_sse) instead of the fused loss kernel;
  * through the fp32 oracle, and through the oracle with bf16 operands in every Linear / attention product
    (oracle/emulate.py) - the arithmetic the HIP path is specified to have;
  * through the fp32 oracle with ONLY the weights rounded to bf16 (a fixed perturbation of the model);
and prints every loss, its relative distance from the fp32 oracle, and the mask MSE of chunks 0-1.
Usage (GPU box): python tools/loss_bias.py [--chunks 32]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle import emulate
from css_amd import ops
from css_amd.models import MODELS
from css_amd.utils import synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=32)
ap.add_argument("--blocks", type=int, default=16)
ap.add_argument("--seed", type=int, default=0, help="weight seed (0 = bench.py's model); the mixtures stay rank 0's")
ap.add_argument("--quick", action="store_true", help="only: default, no_ln_fold, fp32 oracle, bf16-operand oracle")
a = ap.parse_args()
B, T = a.chunks, 400
N = 512 + 256 * (T - 1)
conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=a.blocks, kernel_size=33, dropout_rate=0.1,
            relative_pos_emb=True)
sd = syn.fill_state_dict(syn.conformer_shapes(num_blocks=a.blocks), a.seed)
mix, src = syn.synthetic_sources(32, N, seed=0)
mix, src = mix[:B], [s[:B] for s in src]
audio = torch.stack([mix, src[0], src[1]], 0).cuda()
out = {}


def hip(name, **attrs):
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    for k, v in attrs.items():
        setattr(m, k, v)
    with torch.no_grad():
        mags = ops.stft(audio, ("mag",))["mag"]
        loss, perm, _ = m.forward_loss(mags[0], [mags[1], mags[2]])
        masks = m.forward_masks(mags[0][:2].contiguous())
    torch.cuda.synchronize()
    out[name] = (float(loss), masks.cpu())
    print(f"hip  {name:28s} loss {float(loss):.7f}", flush=True)


hip("default (fused head+PIT)")
if not a.quick:
    hip("materialised masks + pit_sse", no_head_pit=True)
    hip("no_q_log2e", no_q_log2e=True)
hip("no_ln_fold", no_ln_fold=True)
if not a.quick:
    hip("any-width kernels", force_generic=True)

torch.set_num_threads(min(16, os.cpu_count() or 1))
o_mag = [oracle.feature.stft(x)[0].transpose(1, 2).contiguous() for x in (mix, src[0], src[1])]


def ora(name, sd_, ctx=None):
    t0 = time.time()
    with torch.no_grad():
        if ctx is None:
            y, masks = oracle.conformer.forward(o_mag[0], sd_, num_blocks=a.blocks)
        else:
            with ctx():
                y, masks = oracle.conformer.forward(o_mag[0], sd_, num_blocks=a.blocks)
        loss = float(oracle.pit.mse_objective(y, o_mag[1], o_mag[2]))
    out[name] = (loss, torch.cat(masks, -1)[:2])
    print(f"orac {name:28s} loss {loss:.7f}  ({time.time() - t0:.1f} s)", flush=True)


ora("fp32", sd)
ora("bf16 operands (emulate)", sd, emulate.bf16_operands)
if not a.quick:
    sd_w = {k: (v.to(torch.bfloat16).float() if (k.endswith(".weight") and v.dim() == 2 and "pe_k" not in k) else v) for k, v in sd.items()}
    ora("fp32 math, bf16 Linear weights", sd_w)

ref_loss, ref_masks = out["fp32"]
rows = []
for k, (l, mk) in out.items():
    mse = float(((mk[..., :771] - ref_masks) ** 2).mean())
    rows.append({"variant": k, "loss": l, "rel_vs_fp32_oracle": (l - ref_loss) / ref_loss, "mask_mse_chunks01_vs_fp32_oracle": mse})
    print(f"{k:34s} loss {l:.7f}  rel {(l - ref_loss) / ref_loss:+.3e}  mask MSE (chunks 0-1) {mse:.3e}")
em_loss = out["bf16 operands (emulate)"][0]
print(f"HIP default vs bf16-operand oracle: rel {(out['default (fused head+PIT)'][0] - em_loss) / em_loss:+.3e}")
print(json.dumps({"chunks": B, "blocks": a.blocks, "seed": a.seed, "rows": rows}))
