"""BLSTM (config 1 shape class: hidden 1024 x 3 layers, train.sh flags) forward + PIT and training-step timing."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from css_amd import ops
from css_amd.models import MODELS
from css_amd.objectives import OBJECTIVES
from css_amd.optim import FusedAdam
from css_amd.utils import synthetic as syn

hd, nl, B, T = 1024, 3, 32, 400
m = MODELS["BLSTM"](257, 257, 2, 1, dict(hidden_dim=hd, num_layers=nl, dropout_rate=0.0))
m.load_state_dict(syn.fill_state_dict(syn.blstm_shapes(hidden_dim=hd, num_layers=nl), 0))
m = m.cuda().eval()
f = torch.rand(B, T, 257, device="cuda")
tg = [torch.rand(B, T, 257, device="cuda") * 0.5 for _ in range(2)]
with torch.no_grad():
    for _ in range(3):
        ops.pit_mse(m.forward_masks(f), f, tg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        loss = ops.pit_mse(m.forward_masks(f), f, tg)[0]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"BLSTM {hd}x{nl} fwd+PIT 32x400: {dt*1e3:.2f} ms/step = {B*T/dt/1e6:.3f} M frames/s")
m.train()
obj = OBJECTIVES["MSE"].build_objective({})
opt = FusedAdam(m.parameters(), lr=1e-4, max_norm=5.0)
batch = {"mix": f, "source1": tg[0], "source2": tg[1]}
def step():
    opt.zero_grad(); l = obj(m, batch, device="cuda"); l.backward(); opt.step(); return l
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"BLSTM {hd}x{nl} train step (HIP BPTT): {dt*1e3:.1f} ms = {B*T/dt/1e6:.3f} M frames/s, loss {float(l):.4f}")
