import os, sys
sys.path.insert(0, "/root/repo")
import torch
from css_amd import ops_bwd as ob
dev = torch.device("cuda", 0)
K = 12800
shapes = [(1024, 256), (256, 1024), (1024, 256), (256, 1024), (768, 256), (256, 256)]
g = torch.Generator(device="cpu").manual_seed(0)
prods = []
for M, N in shapes:
    dy = (0.5 * torch.randn(K, M, generator=g)).to(torch.bfloat16).to(dev)
    x = (0.5 * torch.randn(K, N, generator=g)).to(torch.bfloat16).to(dev)
    prods.append((dy, x, torch.empty(M, N, device=dev), torch.empty(M, device=dev)))
def t(fn, n=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for rep in range(4):
    row = []
    for dy, x, dw, db in prods:
        row.append(t(lambda: ob.linear_wgrad(dy, x, dw_out=dw, db_out=db)))
    print(rep, " ".join(f"{v:7.1f}" for v in row), "ws ptr", hex(ob._ws_cache[list(ob._ws_cache)[0]].data_ptr()) if ob._ws_cache else None)
