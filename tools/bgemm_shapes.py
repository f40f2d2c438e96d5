"""Which batched-GEMM shapes does one Conformer training step launch, and what does each cost?
Logs every ops_bwd.bgemm call of one step, then times each distinct configuration by graph replay."""
import collections, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from css_amd import ops_bwd as ob, backward as bw, autograd as ag  # noqa: F401

dev = "cuda"
log = []
orig = ob.bgemm


def spy(a, b, **kw):
    log.append((tuple(a.shape), tuple(a.stride()), a.dtype, tuple(b.shape), tuple(b.stride()), b.dtype,
                {k: (v if not torch.is_tensor(v) else ("T", tuple(v.shape), v.dtype)) for k, v in kw.items()}))
    return orig(a, b, **kw)


ob.bgemm = spy
bw.ob.bgemm = spy
sys.argv = [sys.argv[0], "--steps", "1"]
import runpy
try:
    runpy.run_path(os.path.join(os.path.dirname(__file__), "bench_train.py"), run_name="__main__")
except SystemExit:
    pass
ob.bgemm = orig
bw.ob.bgemm = orig
per_step = len(log) // 3
calls = log[-per_step:]
cnt = collections.Counter()
first = {}
for c in calls:
    key = (c[0], c[1], c[2], c[3], c[4], c[5], tuple(sorted((k, str(v)) for k, v in c[6].items())))
    cnt[key] += 1
    first[key] = c
print(f"{per_step} bgemm calls per step, {len(cnt)} distinct")


def launch_time(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


tot = 0.0
rows = []
for key, n in cnt.items():
    ash, ast, adt, bsh, bst, bdt, kw = first[key]
    a = torch.empty_strided(ash, ast, dtype=adt, device=dev).normal_() if adt != torch.bfloat16 else torch.randn(ash, device=dev).to(adt).as_strided(ash, ast) if False else torch.empty_strided(ash, ast, dtype=adt, device=dev)
    b = torch.empty_strided(bsh, bst, dtype=bdt, device=dev)
    a.normal_(); b.normal_()
    kws = {}
    for k, v in kw.items():
        if isinstance(v, tuple) and v and v[0] == "T":
            kws[k] = torch.zeros(v[1], dtype=v[2], device=dev)
        else:
            kws[k] = v
    us = launch_time(lambda: orig(a, b, **kws))
    akm, bkm = kw.get("a_kmajor", False), kw.get("b_kmajor", False)
    M, K = (ash[-1], ash[-2]) if akm else (ash[-2], ash[-1])
    N = bsh[-1] if bkm else bsh[-2]
    batch = 1
    for d in ash[:-2]:
        batch *= d
    fl = 2.0 * M * N * K * batch
    rows.append((n * us, n, us, M, N, K, batch, akm, bkm, fl / us / 1e6, [k for k in kw if k not in ("a_kmajor", "b_kmajor")]))
    tot += n * us
rows.sort(reverse=True)
for r in rows:
    print(f"{r[0]:8.0f} us = {r[1]:3d} x {r[2]:6.1f} us  M={r[3]:6d} N={r[4]:5d} K={r[5]:6d} batch={r[6]:4d} akm={int(r[7])} bkm={int(r[8])}  {r[9]:6.1f} TF/s  {r[10]}")
print(f"total {tot / 1e3:.2f} ms per step")
