"""Training-step timing at the headline shape (Conformer-base, 32 x 400 frames): HIP forward + PIT + explicit HIP
backward (+ Adam).  Usage: python tools/bench_train.py [--graph]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from css_amd.models import MODELS  # noqa: E402
from css_amd.objectives import OBJECTIVES  # noqa: E402
from css_amd.utils import synthetic as syn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--graph", action="store_true", help="capture the whole step in a HIP graph and replay it")
ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam + clip_grad_norm_ instead of FusedAdam")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--blocks", type=int, default=16)
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--T", type=int, default=400)
ap.add_argument("--large", action="store_true", help="Conformer-large widths (d_model 512, 8 heads, ffn 2048, 18 blocks; two speakers + noise: "
                "the MSE objective is written for two sources) through the any-width training path")
a = ap.parse_args()
dev = "cuda"
if a.large:
    a.blocks = 18 if a.blocks == 16 else a.blocks
    dims = dict(attention_dim=512, attention_heads=8, linear_units=2048)
else:
    dims = dict(attention_dim=256, attention_heads=4, linear_units=1024)
conf = dict(num_blocks=a.blocks, kernel_size=33, dropout_rate=0.1, relative_pos_emb=True, **dims)
m = MODELS["Conformer"](257, 257, 2, 1, conf)
m.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(num_blocks=a.blocks, **dims), 0))
m = m.to(dev).train()
obj = OBJECTIVES["MSE"].build_objective({})
if a.torch_adam:
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
else:
    from css_amd.optim import FusedAdam
    opt = FusedAdam(m.parameters(), lr=1e-4, max_norm=5.0)
g = torch.Generator().manual_seed(0)
batch = {k: (torch.rand((a.B, a.T, 257), generator=g) * s).to(dev) for k, s in (("mix", 1.0), ("source1", 0.5), ("source2", 0.5))}


def step():
    if a.torch_adam:
        opt.zero_grad(set_to_none=True)
    else:
        opt.zero_grad()
    loss = obj(m, batch, device=dev)
    loss.backward()
    if a.torch_adam:
        torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
if a.graph:
    from css_amd.trainer import GraphedTrainStep
    stepper = GraphedTrainStep(m, obj, opt, batch, max_norm=5.0)

    def step():  # noqa: F811
        return stepper(batch)[0]
    step()
    torch.cuda.synchronize()
t0 = time.perf_counter()
trace = []
for _ in range(a.steps):
    loss = step()
    if os.environ.get("TRACE_LOSS"):
        trace.append(float(loss))
if trace:
    print("losses", " ".join(f"{x:.9f}" for x in trace))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"train step (hip backward, {a.blocks} blocks): {dt * 1e3:.3f} ms  "
      f"= {a.B * a.T / dt / 1e6:.3f} M frames/s   loss {float(loss):.5f}   peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
