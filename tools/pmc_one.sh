#!/bin/bash
# usage: tools/pmc_one.sh <name> "<counters>" <python script + args...>   (runs on the GPU box)
name=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc_$name -- python "$@" > gpurun_out/pmc_$name.log 2>&1
echo rc=$?
python - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_$name/*/*_counter_collection.csv')
if not f: print('no counter file'); raise SystemExit
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'bgemm' not in k: continue
    print(k)
    for c,vals in v.items(): print(f'   {c:32s} n={len(vals):3d} mean={sum(vals)/len(vals):14.1f}')
PY
