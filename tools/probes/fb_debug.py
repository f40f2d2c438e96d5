"""debug: fused attention backward vs the old path, mismatch map per 16x16 tile"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from css_amd import ops, ops_bwd as ob
B, T, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))
D = H * 64
qkv = rnd((B, T, 3 * D), 60, 0.5).cuda().to(torch.bfloat16)
pe = rnd((2000, 64), 61, 0.3).cuda()
if len(sys.argv) > 4 and sys.argv[4] == "nope": pe = pe * 0
pe_pack = ops.pack_relpos(pe, 1000)
dao = rnd((B * T, D), 62, 0.2).cuda().to(torch.bfloat16)
ao, lse = ops.mha_relpos_train(qkv, pe_pack, H)
ao2 = ao.view(B * T, D)
p_v, ds0, _, dq10, q_rows = ob.mha_bwd_core(qkv, pe_pack, lse, dao, None, None, H, out=ao2, want_q_rows=True)
ds1, parts, dqkv1 = ob.mha_bwd_fused(qkv, pe_pack, lse, dao, ao2, H, dqkv=torch.zeros_like(qkv))
torch.cuda.synchronize()
a, b = ds1.float().cpu(), ds0.float().cpu()
print("G", parts.shape[0], "max|ds0|", float(b.abs().max()), "max diff", float((a - b).abs().max()))
bad = ((a - b).abs() > 0.02 * b.abs().max())
nt = (T + 15) // 16
for bh in range(min(B * H, 2)):
    m = bad.view(B * H, T, -1)[bh]
    print("bh", bh)
    for ti in range(nt):
        print("".join("X" if m[16 * ti:16 * ti + 16, 16 * tj:16 * tj + 16].any() else "." for tj in range((m.shape[1] + 15) // 16)))
# inside the first bad tile
idx = bad.view(B * H, T, -1).nonzero()
if len(idx):
    bh, i, j = idx[0].tolist()
    ti, tj = i // 16, j // 16
    print("first bad tile", bh, ti, tj)
    sub = bad.view(B * H, T, -1)[bh, 16 * ti:16 * ti + 16, 16 * tj:16 * tj + 16]
    for r in range(sub.shape[0]):
        print("".join("X" if x else "." for x in sub[r].tolist()))
    print("fused", a.view(B * H, T, -1)[bh, i, j], "old", b.view(B * H, T, -1)[bh, i, j])
