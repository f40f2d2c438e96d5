"""Which Python lines launch the torch (non-csship) kernels of one training step?  (torch.profiler, with_stack)"""
import collections, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from torch.profiler import profile, ProfilerActivity
sys.argv = [sys.argv[0], "--steps", "1"]
import runpy
ns = runpy.run_path(os.path.join(os.path.dirname(__file__), "bench_train.py"), run_name="__main__")
step = ns["step"]
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.Counter()
cnt = collections.Counter()
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue  # only top-level aten ops
    dev_t = ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
    if dev_t <= 0:
        continue
    frame = next((s for s in ev.stack if "css_amd" in s or "tools/" in s), ev.stack[0] if ev.stack else "?")
    key = (ev.name, frame.strip()[-90:])
    agg[key] += dev_t
    cnt[key] += 1
tot = sum(agg.values())
print(f"top-level aten ops with device time: {tot / 1e3:.2f} ms")
for (name, frame), t in agg.most_common(40):
    print(f"{t:8.0f} us  {cnt[(name, frame)]:4d} x  {name:28s} {frame}")
