#!/usr/bin/env python3
"""The per-step weight re-pack of the Conformer-base training step (css_amd.autograd._conformer_pack_plan: every bf16
shadow of the model by ONE csship_pack_weights_multi launch): time per launch, replayed back to back in a HIP graph.
An alternative library can be given with CSSHIP_LIB (tools/with_lib.py) to compare two builds on one box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from css_amd import autograd as ag  # noqa: E402
from css_amd.models import MODELS  # noqa: E402
from css_amd.utils import synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=16, kernel_size=33, dropout_rate=0.1,
            relative_pos_emb=True)
model = MODELS["Conformer"](257, 257, 2, 1, conf)
model.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(num_blocks=16), 0))
model = model.to(dev).train()
plan = ag._conformer_pack_plan(model, False, dev)["plan"]
for _ in range(3):
    plan.run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        plan.run()
g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    g.replay()
e1.record()
torch.cuda.synchronize()
src = sum(j[0].numel() * 4 for j in plan.jobs)
dst = sum((j[1].numel() * 2 if j[1] is not None else 0) + (j[2].numel() * 2 if j[2] is not None and not j[7] else 0)
          for j in plan.jobs)
us = e0.elapsed_time(e1) * 1e3 / 100
print(f"{len(plan.jobs)} jobs, {src / 1e6:.1f} MB read, {dst / 1e6:.1f} MB written: {us:.1f} us per launch = {(src + dst) / us / 1e6:.2f} TB/s")
