#!/usr/bin/env python3
"""Rank the kernels of ONE replayed step of a rocprofv3 --kernel-trace CSV by start-to-start time (launch gaps included):
python tools/trace_rank.py <kernel_trace.csv> [anchor-kernel-prefix=stft_kernel] [rows=45]
The step is the shortest stretch between two launches of the anchor kernel (eager warm-up / capture steps are longer)."""
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2] if len(sys.argv) > 2 else "stft_kernel"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k[:k.find("(")] if "(" in k else k


names = [short(r["Kernel_Name"]) for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith(anchor)]
sp, j = min((st[idx[j + 1]] - st[idx[j]], j) for j in range(len(idx) - 1))
a, b = idx[j], idx[j + 1]
print(f"{len(idx)} launches of {anchor}; shortest step {sp / 1000:.1f} us, {b - a} kernels")
acc = collections.OrderedDict()
for i in range(a, b):
    d = acc.setdefault(names[i][:78], [0, 0.0])
    d[0] += 1
    d[1] += (st[i + 1] - st[i]) / 1000
tot = sum(v[1] for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[1]:8.1f} us {100 * v[1] / tot:5.1f}%  x{v[0]:3d}  avg {v[1] / v[0]:6.1f}  {k}")
