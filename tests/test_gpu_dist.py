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
            parallel.allreduce_flat(oo.flat_grad, buckets=3); oo.step(max_norm=5.0); losses.append(float(l))
    return oo.flat_param.clone(), losses
p_e, l_e = run(False)
p_g, l_g = run(True)
ok = ok and torch.equal(p_e, p_g) and l_e == l_g
gathered = [torch.zeros_like(p_g.cpu()) for _ in range(ws)]
dist.all_gather(gathered, p_g.cpu())
ok = ok and all(torch.equal(gathered[0], g) for g in gathered)
print(f"rank {rank}: graphed == eager {torch.equal(p_e, p_g)}, ranks identical {all(torch.equal(gathered[0], g) for g in gathered)}", flush=True)
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
