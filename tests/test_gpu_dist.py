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


DRY_RCCL = r"""
# ONE rank, backend "nccl" (= RCCL), CSS_AMD_DRY_DIST=1: the distributed forms of the step run for real - the captured
# step as a chain of sub-graphs with RCCL AVG all-reduces of the bucket schedule issued between the replays (on RCCL's
# stream, behind events), the eager step with the overlapped reducer - and, the collectives being identities at world
# size 1, must reproduce the single-process step bit for bit.  What this pins on a one-GPU box: the RCCL process group
# comes up on ROCm, ReduceOp.AVG exists, asynchronous handles + graph replays + the optimiser graph are ordered
# correctly, nothing in the distributed path needs more than one device to be exercised.
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["CSS_ROOT"])
import numpy as np
from css_amd import parallel
from css_amd.models import MODELS
from css_amd.objectives import OBJECTIVES
from css_amd.optim import FusedAdam
from css_amd.trainer import GraphedTrainStep
from css_amd.utils import synthetic as syn
torch.cuda.set_device(0)
DEV = "cuda"
def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))
def build():
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=6, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True)
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    m.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(num_blocks=6), 7))
    return m.to(DEV).train()
def batch(i):
    return {"mix": rnd((2, 72, 257), 100 * i).abs().to(DEV), "source1": (rnd((2, 72, 257), 100 * i + 10).abs() * 0.5).to(DEV),
            "source2": (rnd((2, 72, 257), 100 * i + 20).abs() * 0.5).to(DEV), "ovl": 0.5}
obj = OBJECTIVES["MSE"].build_objective({})
def run(form):
    mm = build()
    oo = FusedAdam(mm.parameters(), lr=1e-3, max_norm=5.0)
    losses, extra = [], None
    if form == "graphed":
        st = GraphedTrainStep(mm, obj, oo, batch(1), max_norm=5.0, blocks_per_bucket=2)
        extra = (st.distributed, len(getattr(st, "graphs", [])), list(st.plan))
        for i in range(1, 5):
            losses.append(float(st(batch(i))[0]))
    else:
        red = parallel.OverlappedGradReducer(mm, oo, blocks_per_bucket=2) if parallel.is_distributed() else None
        for i in range(1, 5):
            oo.zero_grad()
            if red: red.begin()
            l = obj(mm, batch(i), device=DEV); l.backward()
            if red: red.finish()
            oo.step(max_norm=5.0); losses.append(float(l))
    torch.cuda.synchronize()
    return oo.flat_param.clone(), losses, extra
p0, l0, _ = run("eager")                       # no process group: the single-process step
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_world_size() == 1 and not parallel.is_distributed()
os.environ["CSS_AMD_DRY_DIST"] = "1"
assert parallel.is_distributed() and dist.get_backend() == "nccl"
t = torch.arange(8, device=DEV, dtype=torch.float32)
dist.all_reduce(t, op=dist.ReduceOp.AVG)       # the op allreduce_flat / allreduce_segment rely on
assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
p1, l1, _ = run("eager")                       # overlapped reducer -> RCCL
p2, l2, ex = run("graphed")                    # sub-graph chain + RCCL between the replays
ok = torch.equal(p0, p1) and l0 == l1 and torch.equal(p0, p2) and l0 == l2
print(f"dry RCCL: eager == single-process {torch.equal(p0, p1)}, graphed == single-process {torch.equal(p0, p2)}; "
      f"distributed capture {ex[0]}, {ex[1]} sub-graphs, {len(ex[2])} all-reduces per step of "
      f"{[hi - lo for lo, hi in ex[2]]} floats", flush=True)
ok = ok and ex[0] and ex[1] >= 3 and len(ex[2]) >= 4
dist.destroy_process_group()
sys.exit(0 if ok else 3)
"""


def test_dry_rccl_one_rank_pushes_the_real_collective_schedule(tmp_path):
    """the RCCL path of the training step exercised on a one-GPU box (world size 1, backend nccl)."""
    script = tmp_path / "dry.py"
    script.write_text(DRY_RCCL)
    env = dict(os.environ, CSS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("CSS_AMD_DRY_DIST", None)
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    print(out.stdout[-1500:])
    assert out.returncode == 0, out.stderr[-3000:]


P2P_WORKER = r"""
# Two rank processes SHARING cuda:0 (a one-GPU box): the direct all-reduce over peer-mapped buffers - hipIpc handles of
# the flat gradient buffers, interprocess events, the host sequence block - against the exact mean in rank order, over
# repeated calls, odd sizes and sub-stretches; then the overlapped gradient reduction of a real training backward
# through it (CSS_AMD_P2P=1) against the same step over the gloo path.
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["CSS_ROOT"])
import numpy as np
from css_amd import parallel
from css_amd.models import MODELS
from css_amd.objectives import OBJECTIVES
from css_amd.optim import FusedAdam
from css_amd.utils import synthetic as syn
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
DEV = "cuda"
def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))
ok = True
n = 1_000_003 + 1          # not a multiple of the world size x 4
flat = torch.empty(n, device=DEV)
red = parallel.P2PAllReduce(flat)
for call, (lo, hi, avg) in enumerate([(0, n, True), (0, n, False), (4096, 777_777, True), (12, 16, True), (0, n, True), (n - 4, n, False)]):
    vals = [rnd((n,), 1000 * call + r) for r in range(ws)]
    flat.copy_(vals[rank].to(DEV))
    want = vals[rank].clone()
    acc = vals[0][lo:hi].clone()
    for r in range(1, ws):
        acc += vals[r][lo:hi]
    want[lo:hi] = acc * (1.0 / ws) if avg else acc
    red.allreduce(lo, hi, avg)
    torch.cuda.synchronize()
    same = torch.equal(flat.cpu(), want)
    ok = ok and same
    print(f"rank {rank}: p2p all-reduce call {call} [{lo}, {hi}) mean={avg}: exact {same}", flush=True)
# asynchronous form, twice in a row on different stretches, then wait
vals = [rnd((n,), 77 + r) for r in range(ws)]
flat.copy_(vals[rank].to(DEV))
h = [red.allreduce_async(0, 500_000, True), red.allreduce_async(500_000, n, True)]
for x in h:
    x.wait()
torch.cuda.synchronize()
want = sum(vals[1:], vals[0].clone()) * (1.0 / ws)
ok = ok and torch.equal(flat.cpu(), want)
print(f"rank {rank}: two asynchronous stretches: exact {torch.equal(flat.cpu(), want)}", flush=True)
dist.barrier()
# a real backward with the overlapped bucket reducer: P2P path == gloo path, bit for bit (world 2: (a + b) / 2 both ways)
def build():
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=3, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True)
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    m.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(num_blocks=3), 7))
    return m.to(DEV).eval()
def batch(i, r):
    return {"mix": rnd((2, 40, 257), 100 * i + r).abs().to(DEV), "source1": (rnd((2, 40, 257), 100 * i + 10 + r).abs() * 0.5).to(DEV),
            "source2": (rnd((2, 40, 257), 100 * i + 20 + r).abs() * 0.5).to(DEV), "ovl": 0.5}
obj = OBJECTIVES["MSE"].build_objective({})
grads = {}
for mode in ("gloo", "p2p"):
    os.environ["CSS_AMD_P2P"] = "1" if mode == "p2p" else "0"
    m = build(); m.eval_gradients = True
    opt = FusedAdam(m.parameters(), lr=1e-3, max_norm=0.0)
    r_ = parallel.OverlappedGradReducer(m, opt, blocks_per_bucket=1)
    for it in range(2):
        opt.zero_grad(); r_.begin(); obj(m, batch(it, rank), device=DEV).backward(); r_.finish()
    torch.cuda.synchronize()
    grads[mode] = opt.flat_grad.clone()
same = torch.equal(grads["gloo"], grads["p2p"])
ok = ok and same and float(grads["p2p"].abs().max()) > 0
print(f"rank {rank}: overlapped bucket reduction through P2P == through gloo: {same}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 3)
"""


def test_p2p_direct_allreduce_two_ranks_on_one_gpu(tmp_path):
    """VERDICT r3 item 8 / SURVEY 5.8: the hand-written direct all-reduce over hipIpc peer mappings, functionally, with
    two rank processes that share the one GPU of the box."""
    script = tmp_path / "p2p.py"
    script.write_text(P2P_WORKER)
    assert _run_ranks(script, port="29549") == [0, 0]


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
    if mode == "fwd":   # the per-kernel records belong to the default mode's line (train-mode forward + PIT, eval leg beside it)
        assert rec["roofline"]["frac"] > 0 and rec["roofline"]["attention"]["avg_launch_us"] > 0
        assert rec["eval_fwd_pit"]["ms_per_step"] > 0 and "train-mode" in rec["config"]["mode"].lower()
    else:
        assert rec["roofline"]["whole_step_frac"] > 0
