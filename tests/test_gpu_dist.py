"""Two ranks of the REAL css_amd training step on one GPU box (both on cuda:0, collectives over gloo - RCCL refuses two
ranks on one device): rank logic, bucketed / overlapped gradient all-reduce, distributed GraphedTrainStep and the
multi-rank bench entry.  The N-GPU RCCL run itself is the driver's (SCALE_rNN.json)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

WORKER = r"""
import os, sys, types, torch, torch.distributed as dist
sys.path.insert(0, os.environ["CSS_ROOT"])
import numpy as np
from css_amd import parallel
from css_amd.models import MODELS
from css_amd.objectives import OBJECTIVES
from css_amd.optim import FusedAdam
from css_amd.trainer import GraphedTrainStep, LRScheduler, train_one_epoch
from css_amd.utils import synthetic as syn
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
DEV = "cuda"
def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))
def build():
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=2, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True)
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    m.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(num_blocks=2), 7))
    return m.to(DEV)
def batch(i, r):   # rank r's shard of global batch i
    return {"mix": rnd((2, 40, 257), 100 * i + r).abs().to(DEV), "source1": (rnd((2, 40, 257), 100 * i + 10 + r).abs() * 0.5).to(DEV),
            "source2": (rnd((2, 40, 257), 100 * i + 20 + r).abs() * 0.5).to(DEV), "ovl": 0.5}
obj = OBJECTIVES["MSE"].build_objective({})
ok = True
# (1) eager train_one_epoch with the overlapped bucket reducer: BatchNorm frozen (eval + eval_gradients) so that the
# 2-rank average equals the single-process gradient of the concatenated batch
m = build().eval(); m.eval_gradients = True
opt = FusedAdam(m.parameters(), lr=1e-3, max_norm=0.0)
reducer = parallel.OverlappedGradReducer(m, opt, blocks_per_bucket=1)
opt.zero_grad(); reducer.begin(); obj(m, batch(0, rank), device=DEV).backward(); reducer.finish()
g_dist = opt.flat_grad.clone()
ref = build().eval(); ref.eval_gradients = True
opt_r = FusedAdam(ref.parameters(), lr=1e-3, max_norm=0.0)
opt_r.zero_grad()
for r in range(ws):
    (obj(ref, batch(0, r), device=DEV) / ws).backward()
err = float((g_dist - opt_r.flat_grad).norm() / opt_r.flat_grad.norm())
ok = ok and err < 1e-5
print(f"rank {rank}: overlapped-bucket gradient vs single-process mean gradient: rel err {err:.2e}", flush=True)
# (2) distributed GraphedTrainStep (two graphs + bucketed all-reduce between) == eager distributed steps, bit for bit,
# and the ranks stay in lockstep
def run(graphed):
    mm = build().train()
    oo = FusedAdam(mm.parameters(), lr=1e-3, max_norm=5.0)
    losses = []
    if graphed:
        st = GraphedTrainStep(mm, obj, oo, batch(1, rank), max_norm=5.0, buckets=3)
        for i in range(1, 4):
            losses.append(float(st(batch(i, rank))[0]))
    else:
        for i in range(1, 4):
            oo.zero_grad(); l = obj(mm, batch(i, rank), device=DEV); l.backward()
            parallel.allreduce_plan(oo.flat_grad, parallel.bucket_plan(mm, oo, 4)); oo.step(max_norm=5.0); losses.append(float(l))
    return oo.flat_param.clone(), losses
p_e, l_e = run(False)
p_g, l_g = run(True)
ok = ok and torch.equal(p_e, p_g) and l_e == l_g
gathered = [torch.zeros_like(p_g.cpu()) for _ in range(ws)]
dist.all_gather(gathered, p_g.cpu())
ok = ok and all(torch.equal(gathered[0], g) for g in gathered)
print(f"rank {rank}: graphed == eager {torch.equal(p_e, p_g)}, ranks identical {all(torch.equal(gathered[0], g) for g in gathered)}", flush=True)
# (3) SyncBatchNorm: train mode (batch statistics), dropout off; two ranks x 2 windows with the statistics all-reduced
# == one process on the 4 concatenated windows: loss, averaged gradient and running statistics
def build_nodrop():
    mm = build(); mm.conformer.dropout_rate = 0.0
    return mm.train()
m = parallel.convert_sync_batchnorm(build_nodrop())
opt = FusedAdam(m.parameters(), lr=1e-3, max_norm=0.0)
opt.zero_grad(); l_sync = obj(m, batch(5, rank), device=DEV); l_sync.backward(); parallel.allreduce_flat(opt.flat_grad)
l_mean = parallel.allreduce_sum_(l_sync.detach().clone().reshape(1)) / ws
ref = build_nodrop()
opt_r = FusedAdam(ref.parameters(), lr=1e-3, max_norm=0.0)
opt_r.zero_grad()
shards = [batch(5, r) for r in range(ws)]
big = {k: torch.cat([sh[k] for sh in shards]) for k in ("mix", "source1", "source2")}; big["ovl"] = 0.5
l_big = obj(ref, big, device=DEV); l_big.backward()
err = float((opt.flat_grad - opt_r.flat_grad).norm() / opt_r.flat_grad.norm())
bn, bn_r = m.conformer.encoders[0].conv.BN, ref.conformer.encoders[0].conv.BN
err_rm = float((bn.running_mean - bn_r.running_mean).abs().max()); err_rv = float((bn.running_var - bn_r.running_var).abs().max() / bn_r.running_var.abs().max())
# and WITHOUT the switch the per-rank statistics give a different gradient (the test has power)
m2 = build_nodrop(); o2 = FusedAdam(m2.parameters(), lr=1e-3, max_norm=0.0)
o2.zero_grad(); obj(m2, batch(5, rank), device=DEV).backward(); parallel.allreduce_flat(o2.flat_grad)
err_nosync = float((o2.flat_grad - opt_r.flat_grad).norm() / opt_r.flat_grad.norm())
# the operator itself (f32 throughout): shards with sync == the concatenated batch without
from css_amd import ops, ops_bwd as ob
Dm = 256
Wc = {"ln_g": 1 + 0.1 * rnd((Dm,), 1), "ln_b": 0.1 * rnd((Dm,), 2), "pw1_w": rnd((2,), 3), "pw1_b": 0.1 * rnd((2,), 4),
      "dw_w": 0.2 * rnd((33, Dm), 5), "dw_b": 0.1 * rnd((Dm,), 6), "bn_g": 1 + 0.1 * rnd((Dm,), 7), "bn_b": 0.1 * rnd((Dm,), 8),
      "bn_mean": torch.zeros(Dm), "bn_var": torch.ones(Dm), "pw2_w": rnd((1,), 9), "pw2_b": 0.1 * rnd((1,), 10)}
Wc = {k: v.to(DEV) for k, v in Wc.items()}
xs = [rnd((3, 50, Dm), 40 + r).to(DEV) for r in range(ws)]; dys = [rnd((3, 50, Dm), 50 + r).to(DEV) for r in range(ws)]
y_s, fws_s = ops.convmod(xs[rank], Wc, 33, batch_stats=True, return_workspace=True, sync=True)
dx_s, g_s = ob.convmod_bwd(xs[rank], dys[rank], Wc, fws_s, 33, sync=True)
y_b, fws_b = ops.convmod(torch.cat(xs), Wc, 33, batch_stats=True, return_workspace=True)
dx_b, g_b = ob.convmod_bwd(torch.cat(xs), torch.cat(dys), Wc, fws_b, 33)
lo = 3 * rank
e_y = float((y_s - y_b[lo:lo + 3]).abs().max()); e_dx = float((dx_s - dx_b[lo:lo + 3]).abs().max() / dx_b.abs().max())
e_g, e_k = 0.0, {}
for k in g_b:
    gs = parallel.allreduce_sum_(g_s[k].clone())
    # d dw_b is zero in exact arithmetic (BatchNorm removes the channel mean): measured against d dw_w's scale
    e_k[k] = float((gs - g_b[k]).abs().max() / ((g_b["dw_w"] if k == "dw_b" else g_b[k]).abs().max() + 1e-20))
    e_g = max(e_g, e_k[k])
op_ok = e_y < 1e-4 and e_dx < 1e-4 and e_g < 1e-4
ok = ok and op_ok
print(f"rank {rank}: SyncBN conv module vs concatenated: y {e_y:.1e} dx {e_dx:.1e} grads {e_g:.1e} " + " ".join(f"{k}:{v:.0e}" for k, v in e_k.items()), flush=True)
# noise floor of "2 x 2 windows vs 4 windows" with nothing to synchronise (frozen BatchNorm): launch shapes differ with M
# (tiles, split-K of the weight gradients), so bf16 roundings fall differently
fz = build().eval(); fz.eval_gradients = True; fz.conformer.dropout_rate = 0.0
of = FusedAdam(fz.parameters(), lr=1e-3, max_norm=0.0)
of.zero_grad(); obj(fz, big, device=DEV).backward(); g_big = of.flat_grad.clone()
of.zero_grad()
for sh in shards:
    (obj(fz, sh, device=DEV) / ws).backward()
err_floor = float((of.flat_grad - g_big).norm() / g_big.norm())
good = err < 8 * err_floor and abs(float(l_mean) - float(l_big)) < 1e-4 * abs(float(l_big)) and err_rm < 1e-5 and err_rv < 1e-4 and err_nosync > 5 * err
ok = ok and good
print(f"rank {rank}: SyncBN gradient vs concatenated batch: rel err {err:.2e} (unsynced {err_nosync:.2e}, frozen-BN floor {err_floor:.2e}), loss {float(l_mean):.6f} vs {float(l_big):.6f}, "
      f"running mean/var err {err_rm:.1e}/{err_rv:.1e}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 3)
"""


def _run_ranks(script, n=2, port="29541"):
    env = dict(os.environ, CSS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE=str(n))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]
    return [p.wait(timeout=600) for p in procs]


def test_two_rank_training_step_on_one_gpu(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    assert _run_ranks(script) == [0, 0]


@pytest.mark.parametrize("mode", ["fwd", "train"])
def test_bench_launches_its_own_ranks(mode):
    """`bench.py --gpus 2` without a launcher starts two rank processes and rank 0 reports n_gpus = 2 (rehearsal flags:
    both ranks on cuda:0, gloo)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--steps", "2",
                          "--warmup", "1", "--blocks", "2", "--mode", mode, "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env={k: v for k, v in os.environ.items()
                                                                          if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and len(rec["per_rank_ms_per_step"]) == 2 and rec["config"]["global_batch"] == 64
    assert rec["roofline"]["frac"] > 0 and rec["roofline"]["attention"]["avg_launch_us"] > 0
