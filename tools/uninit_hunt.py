"""Poison every fresh allocation (torch.empty*) with NaN bytes and run training steps: a NaN in the loss or in a
gradient means some kernel read memory it (or its producer) never wrote."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch

_empty, _empty_like, _empty_strided = torch.empty, torch.empty_like, torch.empty_strided


def _poison(t):
    if t.is_cuda and t.numel():
        if t.dtype in (torch.float32, torch.bfloat16, torch.float16, torch.float64):
            t.fill_(float("nan"))
        elif t.dtype == torch.uint8:
            t.fill_(255)
    return t


torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
torch.empty_strided = lambda *a, **k: _poison(_empty_strided(*a, **k))

from css_amd.models import MODELS
from css_amd.objectives import OBJECTIVES
from css_amd.optim import FusedAdam
from css_amd.utils import synthetic as syn

dev = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "conformer"
if which == "conformer":
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=2, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True)
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    m.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(num_blocks=2), 0))
else:
    m = MODELS["BLSTM"](257, 257, 2, 1, dict(hidden_dim=1024, num_layers=2, dropout_rate=0.1))
    m.load_state_dict(syn.fill_state_dict(syn.blstm_shapes(hidden_dim=1024, num_layers=2), 0))
m = m.to(dev).train()
obj = OBJECTIVES["MSE"].build_objective({})
opt = FusedAdam(m.parameters(), lr=1e-4, max_norm=5.0)
g = torch.Generator().manual_seed(0)
B, T = (32, 400) if which == "conformer" else (20, 50)
batch = {k: (torch.rand((B, T, 257), generator=g) * s).to(dev) for k, s in (("mix", 1.0), ("source1", 0.5), ("source2", 0.5))}
for step in range(3):
    opt.zero_grad()
    loss = obj(m, batch, device=dev)
    loss.backward()
    bad = [n for n, p in m.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    print(f"step {step}: loss {float(loss):.6f}  non-finite gradients: {len(bad)} {bad[:6]}")
    opt.step()
m.eval()
with torch.no_grad():
    y = m.forward_masks(batch["mix"])
print("eval forward finite:", bool(torch.isfinite(y).all()))
