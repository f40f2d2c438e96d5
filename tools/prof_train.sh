#!/bin/bash
# kernel stats of the training step, twice on one box: default build and with the env given as $1 (e.g. CSSHIP_BGEMM_NOGLDS=1)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for leg in B A; do
  rm -rf gpurun_out/pt_$leg
  if [ $leg = A ] && [ -n "${1:-}" ]; then export "$1"; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pt_$leg -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --mode train > gpurun_out/pt_$leg.log 2>&1
  rc=$?; echo "leg $leg rc=$rc"
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
  f=$(ls gpurun_out/pt_$leg/*/*kernel_stats.csv | tail -1)
  cp "$f" gpurun_out/pt_${leg}_kernel_stats.csv
  python - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot / 1e6)
for r in rows[:22]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-110s %6d %9.1f us %5.1f%%" % (n[:110], int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
done
