"""CPU-side tests: C-ABI surface, host-side packing/layout logic, trainer utilities, data parallel helpers
(gloo, world_size 2).  No kernel is launched here."""
import ctypes
import json
import os
import re
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT


def test_abi_exports_every_declared_symbol():
    """libcsship.so must export exactly the entry points include/csship.h declares."""
    from css_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "csship.h")).read()
    declared = set(re.findall(r"\b(csship_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    assert os.path.exists(_lib.LIB_PATH), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    h = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(h, name), f"{name} declared in csship.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes signature table and header disagree"
    # ... and nothing else: every csship_* symbol the library exports with C linkage is declared in the header
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln and ln.split()[-1].startswith("csship_")}
    assert exported == declared, f"exported but not declared: {sorted(exported - declared)}; declared only: {sorted(declared - exported)}"
    lib = _lib.lib()
    assert lib.csship_version() >= 100
    assert b"NULL" in lib.csship_strerror(-5)


def test_null_and_shape_errors_need_no_gpu():
    """Argument validation happens on the host before any launch: safe to call without a GPU."""
    from css_amd import _lib
    lib = _lib.lib()
    assert lib.csship_stft(None, 1, 0, 1, 0, 1024, None, None, None, None, None) == -5
    assert lib.csship_mvn_stats(None, 1, 1, 1, 1, 0.0, 0.0, None, None, None) == -5
    p = _lib.GemmParams(M=0, N=1, K=1)
    assert lib.csship_gemm(ctypes.c_void_p(16), ctypes.c_void_p(16), None, None, ctypes.c_void_p(16), None, None,
                           None, None, ctypes.byref(p), None) == -1


def test_ops_refuse_cpu_tensors():
    from css_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.stft(torch.zeros(1, 1024))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.mvn_stats(torch.zeros(1, 4, 257))
    from css_amd.objectives import pit_loss
    with pytest.raises(RuntimeError):
        pit_loss(torch.zeros(2, 4, 8), torch.zeros(2, 4, 8), axis=0)


def test_pack_frag_layout():
    """block(rt, ks)[16*g + m][j] == W[16*rt + m][32*ks + 8*g + j], zero beyond R (include/csship.h)."""
    from css_amd import ops
    R, K = 40, 64
    w = torch.arange(R * K, dtype=torch.float32).reshape(R, K) / 8.0  # exactly representable in bf16? keep small
    w = (torch.arange(R)[:, None] * 1.0 + torch.arange(K)[None, :] * 0.0078125).float()
    p = ops.pack_frag(w).float().view(3, 2, 64, 8)
    wb = w.to(torch.bfloat16).float()
    for rt, ks, g, m, j in [(0, 0, 0, 0, 0), (1, 1, 3, 15, 7), (2, 0, 2, 7, 3), (2, 1, 1, 8, 0)]:
        r, k = 16 * rt + m, 32 * ks + 8 * g + j
        want = wb[r, k] if r < R else 0.0
        assert float(p[rt, ks, 16 * g + m, j]) == float(want)


def test_pack_relpos_blocks():
    from css_amd import ops
    maxlen = 1000
    pe = torch.randn(2 * maxlen, 64)
    packed, qmin, qmax = ops.pack_relpos(pe, maxlen)
    nq = qmax - qmin + 1
    blk = packed.float().view(nq, 2, 64, 8)
    peb = pe.to(torch.bfloat16).float()
    for q, ks, g, m, j in [(0, 0, 0, 0, 0), (-3, 1, 2, 5, 7), (62, 0, 3, 15, 1), (qmin, 1, 1, 1, 1), (qmax, 0, 0, 9, 2)]:
        rel = min(max(16 * q + 15 - m, -maxlen), maxlen - 1)
        assert float(blk[q - qmin, ks, 16 * g + m, j]) == float(peb[rel + maxlen, 32 * ks + 8 * g + j])
    # the end blocks are entirely clamped (kernel clamps q into [qmin, qmax])
    assert 16 * qmin + 15 <= -maxlen and 16 * qmax >= maxlen - 1


def test_model_registry_and_state_dict():
    from css_amd.models import MODELS, build_model
    keys = json.load(open(os.path.join(GOLDEN, "keys.json")))["conformer_base"]
    import argparse
    ap = argparse.ArgumentParser()
    MODELS["Conformer"].add_args(ap)
    conf = vars(ap.parse_args(["--conformer-relative-pos-emb", "true"]))
    m = build_model("Conformer", conf)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == keys
    with pytest.raises(RuntimeError, match="MI355X|cuda|CUDA"):
        m(torch.zeros(1, 8, 257))  # CPU module: must fail loudly, never fall back


def test_lr_scheduler_matches_reference_trace():
    from css_amd.trainer import LRScheduler
    tr = json.load(open(os.path.join(GOLDEN, "trainer.json")))
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    sch = LRScheduler(opt, types.SimpleNamespace(warmup=4, decay=0.1, fixed=2, min_lr=1e-9))
    got = []
    for _ in range(10):
        got.append(sch.curr_lr)
        sch.step(1.0)
    assert np.allclose(got, tr["lr_trace"], rtol=1e-12, atol=0)
    st = sch.state_dict()
    sch2 = LRScheduler(opt, types.SimpleNamespace(warmup=0, decay=0.0, fixed=0, min_lr=0.0))
    sch2.load_state_dict(st)
    assert sch2.state_dict() == st


def test_shard_range_partitions():
    from css_amd.parallel import shard_range
    for n in (0, 1, 7, 747, 1000):
        for ws in (1, 2, 3, 8):
            spans = [shard_range(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_average_parameters():
    from css_amd.parallel import average_parameters
    a = {"w": torch.tensor([1.0, 3.0]), "n": torch.tensor(4)}
    b = {"w": torch.tensor([3.0, 5.0]), "n": torch.tensor(6)}
    out = average_parameters([a, b])
    # 0-dim buffers follow combine_models.py:77-78: the LAST job's value x 1/n, truncated (golden: test_combine_models_golden)
    assert torch.equal(out["w"], torch.tensor([2.0, 4.0])) and int(out["n"]) == 3 and out["n"].dtype == torch.int64


WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["CSS_ROOT"])
from css_amd import parallel
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, ws = dist.get_rank(), dist.get_world_size()
torch.manual_seed(0)
model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
data = torch.randn(8, 8); tgt = torch.randn(8, 4)
# single-process big-batch gradient
model.zero_grad(); torch.nn.functional.mse_loss(model(data), tgt).backward()
ref = [p.grad.clone() for p in model.parameters()]
# this rank's shard, then one flat all-reduce (mean)
lo, hi = parallel.shard_range(8, rank, ws)
model.zero_grad(); torch.nn.functional.mse_loss(model(data[lo:hi]), tgt[lo:hi]).backward()
parallel.allreduce_gradients(model.parameters(), average=True)
ok = all(torch.allclose(p.grad, r, atol=1e-6) for p, r in zip(model.parameters(), ref))
# the same through an already-flat gradient buffer (what css_amd.optim.FusedAdam hands to the collective)
model.zero_grad(); torch.nn.functional.mse_loss(model(data[lo:hi]), tgt[lo:hi]).backward()
flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
parallel.allreduce_flat(flat, average=True)
ok = ok and torch.allclose(flat, torch.cat([r.reshape(-1) for r in ref]), atol=1e-6)
# bucketed (chunks issued back to back), asynchronous (handles waited for later) and per-segment forms; SyncBN-style sum
want = torch.cat([r.reshape(-1) for r in ref])
def fresh():
    model.zero_grad(); torch.nn.functional.mse_loss(model(data[lo:hi]), tgt[lo:hi]).backward()
    return torch.cat([p.grad.reshape(-1) for p in model.parameters()])
f3 = fresh(); parallel.allreduce_flat(f3, average=True, buckets=3)
ok = ok and torch.allclose(f3, want, atol=1e-6)
fa, hs = fresh(), []
parallel.allreduce_flat(fa, average=True, buckets=2, async_handles=hs)
ok = ok and len(hs) == 2
parallel.wait_all(hs, fa, average=True)
ok = ok and torch.allclose(fa, want, atol=1e-6)
fs, hs = fresh(), []
mid = fs.numel() // 2 // 4 * 4
parallel.allreduce_segment(fs, mid, fs.numel(), hs); parallel.allreduce_segment(fs, 0, mid, hs)   # last block first
parallel.wait_all(hs, fs, average=True)
ok = ok and torch.allclose(fs, want, atol=1e-6)
# the collective schedule both step forms issue (parallel.bucket_plan -> allreduce_plan): a plan of stretches in backward
# order, here rank 0 issues it in one go and rank 1 stretch by stretch with work in between (an eager rank next to a
# replaying one): same result, ranks stay paired
fp = fresh()
n = fp.numel()
plan = [(n - n // 3 // 4 * 4, n), (n // 4 // 4 * 4, n - n // 3 // 4 * 4), (0, n // 4 // 4 * 4)]
if rank == 0:
    parallel.allreduce_plan(fp, plan)
else:
    hs = []
    for lo_, hi_ in plan:
        parallel.allreduce_segment(fp, lo_, hi_, hs)
        _ = torch.randn(64, 64) @ torch.randn(64, 64)
    parallel.wait_all(hs, fp, average=True)
ok = ok and torch.allclose(fp, want, atol=1e-6)
t = torch.full((5,), float(rank + 1), dtype=torch.float64)
ok = ok and torch.equal(parallel.allreduce_sum_(t), torch.full((5,), 3.0, dtype=torch.float64))
# window gather
local = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 3)
full = parallel.gather_windows(local, 8, rank, ws)
ok = ok and torch.equal(full[:, 0], torch.arange(8, dtype=torch.float32))
dist.destroy_process_group()
sys.exit(0 if ok else 3)
"""


def test_gradient_allreduce_gloo_world2(tmp_path):
    """N-rank flat-bucket all-reduce == single-process big-batch gradient (equal shards, mean loss)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, CSS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    rcs = [p.wait(timeout=120) for p in procs]
    assert rcs == [0, 0]


def test_checkpoint_roundtrip_and_combine(tmp_path):
    """Reference on-disk format: keys of train.py:304-317, averaging of combine_models.py:55-82."""
    from css_amd.models import MODELS
    from css_amd.trainer import LRScheduler
    from css_amd.utils import checkpoint as ck
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=1, kernel_size=33,
                dropout_rate=0.1, relative_pos_emb=True)
    a, b = MODELS["Conformer"](257, 257, 2, 1, conf), MODELS["Conformer"](257, 257, 2, 1, conf)
    opt = torch.optim.Adam(a.parameters(), lr=1e-3)
    sch = LRScheduler(opt, types.SimpleNamespace(warmup=4, decay=0.1, fixed=2, min_lr=1e-9))
    pa, pb = str(tmp_path / "3.1.mdl"), str(tmp_path / "3.2.mdl")
    ck.save_checkpoint(pa, a, None, opt, sch, epoch=2, loss=0.5)
    ck.save_checkpoint(pb, b, None, None, None, epoch=2, loss=0.6)
    saved = torch.load(pa)
    assert set(saved) == {"model", "objective", "optimizer", "lr_sched", "epoch", "loss"}
    assert ck.epoch_checkpoints(str(tmp_path), 3) == [pa, pb]
    out = ck.combine_models(str(tmp_path / "3.mdl"), [pa, pb])
    avg = torch.load(out)["model"]
    k = "conformer.encoders.0.self_attn.linear_q.weight"
    assert torch.allclose(avg[k], (a.state_dict()[k] + b.state_dict()[k]) / 2)
    assert avg["conformer.encoders.0.conv.BN.num_batches_tracked"].dtype == torch.int64
    c = MODELS["Conformer"](257, 257, 2, 1, conf)
    assert ck.load_checkpoint(out, c) == 2
    assert torch.equal(c.state_dict()[k], avg[k])


def test_beamformer_spans_follow_the_reference_batch_end():
    """beamformer.py:57,104-118: the per-window test `en == T` uses en = hop * (min(first + batch, W) + 1) + hop * i, one
    window more than the last (truncated) batch holds.  Session length chosen so that the tail branch fires."""
    import types
    from css_amd.executor.beamformer import Beamformer
    hop, margin, bs = 100, 160, 4
    self = types.SimpleNamespace(proceed_margin=margin, eval_hop=hop)
    W = 6                                    # windows; batches [0..4], [4..6) (+1 beyond the end)
    for first in (0, 4):
        bend = min(first + bs, W) + 1
        count = min(bend, W) - first
        for T in (hop * (bend + 1), hop * (bend + 7)):   # T = hop * (bend + i) for i = 1 (a window of the batch) / never
            got = Beamformer._spans(self, first, count, T, batch_end=bend)
            st, en, want = hop * first, hop * bend, []
            for _ in range(count):           # the reference's loop, literally
                if st == 0:
                    want.append((0, margin))
                elif en == T:
                    want.append((margin - hop, None))
                else:
                    want.append((margin - hop, hop))
                st += hop
                en += hop
            assert got == want, (first, T)
    # the old computation (batch end = first + count) differs exactly on the truncated batch
    assert Beamformer._spans(self, 4, 2, hop * 8) != Beamformer._spans(self, 4, 2, hop * 8, batch_end=7)


def test_bucket_plan_partitions_the_flat_buffer_in_backward_order():
    """parallel.bucket_plan: the collective schedule both step forms issue - a partition of the flat gradient buffer,
    mask head first (the END of the buffer), encoder blocks in groups from the last to the first, the front last."""
    import types
    from css_amd import parallel
    from css_amd.models import MODELS
    conf = dict(attention_dim=64, attention_heads=1, linear_units=64, num_blocks=5, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True)
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    params = [p for p in m.parameters()]
    offs, n = [], 0
    for p in params:
        offs.append(n)
        n += (p.numel() + 3) // 4 * 4
    opt = types.SimpleNamespace(param_groups=[{"params": params}], _offs=offs, flat_grad=torch.zeros(n))
    for every, nseg in ((1, 7), (2, 5), (4, 4)):   # head + ceil(5 / every) block groups + the front
        plan = parallel.bucket_plan(m, opt, every)
        assert len(plan) == nseg, (every, plan)
        assert plan[0][1] == n and plan[-1][0] == 0
        assert all(plan[i][0] == plan[i + 1][1] for i in range(len(plan) - 1))   # contiguous, descending: a partition
        head_lo = min(offs[i] for i, p in enumerate(params) if any(p is q for q in m.linear.parameters()))
        assert plan[0] == (head_lo, n)
    assert getattr(m, "_grad_bucket_hook", None) is None   # planning leaves no hook behind


def test_cached_const_is_bounded_and_keeps_recent_entries():
    """ops_bwd.cached_const: least recently used entries leave once the table is full; a hit does not rebuild"""
    from css_amd import ops_bwd as ob
    saved, keep = dict(ob._CONST), set(ob._CONST_KEEP)
    ob._CONST.clear(), ob._CONST_KEEP.clear()
    try:
        built = []
        for i in range(ob._CONST_MAX + 5):
            ob.cached_const(("t", i), lambda i=i: built.append(i) or torch.full((1,), float(i)))
            ob.cached_const(("t", 0), lambda: built.append(-1) or torch.zeros(1))   # touched every round: stays
        assert len(ob._CONST) == ob._CONST_MAX and ("t", 0) in ob._CONST and ("t", 1) not in ob._CONST
        assert -1 not in built and built == list(range(ob._CONST_MAX + 5))
        assert float(ob.cached_const(("t", ob._CONST_MAX + 4), lambda: torch.zeros(1))) == ob._CONST_MAX + 4
    finally:
        ob._CONST.clear(), ob._CONST.update(saved)
        ob._CONST_KEEP.clear(), ob._CONST_KEEP.update(keep)
