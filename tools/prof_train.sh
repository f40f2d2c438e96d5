#!/bin/bash
# rocprofv3 kernel stats of the graphed Conformer training step (on the GPU box): per-step totals of the top kernels
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_train
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python tools/bench_train.py --graph > gpurun_out/pt.log 2>&1
tail -1 gpurun_out/pt.log
python - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_train/*/*kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms/step", round(tot / 10 / 1e6, 2))
for r in rows[:${1:-24}]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{float(r['TotalDurationNs'])/10/1e3:8.0f} us {int(r['Calls'])//10:4d} x {float(r['AverageNs'])/1e3:6.1f}  {n[:90]}")
PY
