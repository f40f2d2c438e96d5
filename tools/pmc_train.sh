#!/bin/bash
# HBM traffic of the training step's kernels: FETCH_SIZE and WRITE_SIZE in their own passes (tools/pmc_train.sh [pattern])
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1)); rm -rf gpurun_out/pmct_$i
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmct_$i -- python bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmct_$i.log 2>&1
  rc=$?; echo "pmc [$set] rc=$rc"; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
done
python - "${1:-}" <<'PY'
import collections, csv, glob, sys
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmct_[0-9]*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, v in vals.items():
    if "at::" in k or "rocclr" in k: continue
    n = k.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n[:n.find("(")] if "(" in n else n
    f = sum(v.get("FETCH_SIZE", [0])) / max(1, len(v.get("FETCH_SIZE", [0])))
    w = sum(v.get("WRITE_SIZE", [0])) / max(1, len(v.get("WRITE_SIZE", [0])))
    rows.append((2 * f * 1024 / 1e6, w * 1024 / 1e6, len(v.get("WRITE_SIZE", [])), n))
for fr, wr, cnt, n in sorted(rows, key=lambda t: -(t[0] + t[1]) * t[2])[:28]:
    if sys.argv[1] and sys.argv[1] not in n: continue
    print("%-64s n=%4d  read %7.1f MB (2 x FETCH_SIZE)  write %7.1f MB per launch" % (n[:64], cnt, fr, wr))
PY
