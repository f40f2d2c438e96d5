"""Attention backward at 32 x 4 x 400 (dropout 0.1): the key-stationary fused pass (csship_mha_relpos_bwd_fused) against the
query-stationary pass + csship_attn_dvdk of rounds 3-5, and the position pass behind each.  python tools/bench_attn_bwd.py [B T H]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from css_amd import ops, ops_bwd as ob
from css_amd.dropout import DropState
B, T, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 400, 4)
p_drop = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))
D = H * 64
qkv = rnd((B, T, 3 * D), 60, 0.5).cuda().to(torch.bfloat16)
pe = rnd((2000, 64), 61, 0.3).cuda()
pe_pack = ops.pack_relpos(pe, 1000)
dao = rnd((B * T, D), 62, 0.2).cuda().to(torch.bfloat16)
cfg = DropState(torch.tensor([1234], dtype=torch.int32, device="cuda"), p_drop, 0).cfg("0.attn.p") if p_drop > 0 else None
ao, lse = ops.mha_relpos_train(qkv, pe_pack, H, drop=cfg)
ao2 = ao.view(B * T, D)
rt = ob.relpos_tables(pe, T, 1000, want_t=True)
win, idx, win_t = rt["window"]
acc = torch.zeros((win.shape[0], 64), dtype=torch.float32, device="cuda")
dqkv = torch.zeros_like(qkv)
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
st = {}
def old_core(): st["o"] = ob.mha_bwd_core(qkv, pe_pack, lse, dao, None, None, H, drop=cfg, out=ao2, want_q_rows=True)
def old_dvdk(): ob.attn_dvdk(st["o"][0], st["o"][1], dao, qkv, dqkv, H)
def old_pos(): ob.attn_pos_bwd_ds(st["o"][1], win_t, st["o"][4], st["o"][3], dqkv, acc, H)
def new_core(): st["n"] = ob.mha_bwd_fused(qkv, pe_pack, lse, dao, ao2, H, drop=cfg, dqkv=dqkv)
def new_pos(): ob.attn_pos_bwd_parts(st["n"][0], win_t, qkv, st["n"][1], dqkv, acc, H)
old_core(); new_core()
r = {k: timed(f) for k, f in (("old one-pass core", old_core), ("old dvdk", old_dvdk), ("old position pass", old_pos), ("fused key-stationary", new_core), ("new position pass", new_pos))}
for k, v in r.items(): print(f"{k:24s} {v:7.1f} us")
print(f"old total {r['old one-pass core'] + r['old dvdk'] + r['old position pass']:.1f} us, new total {r['fused key-stationary'] + r['new position pass']:.1f} us  (B={B} T={T} H={H} p={p_drop}; eager launches incl. allocations)")
