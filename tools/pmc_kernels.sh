#!/bin/bash
# Hardware counters of chosen kernels of any command, on the GPU box (via gpurun): every counter set in its own pass,
# never together with a tracing domain other than --kernel-trace.  Prints the per-launch average of every counter.
#   tools/pmc_kernels.sh "<name substring>[|<substring>...]" python tools/bench_wgrad_group.py --iters 5
#   tools/pmc_kernels.sh "ffn_bwd_kernel|chain_s_kernel" python bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-train-step
filter=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf gpurun_out/pk_$i
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pk_$i -- "$@" > gpurun_out/pk_$i.log 2>&1
  rc=$?; echo "pmc [$set] rc=$rc"
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
done
FILTER="$filter" python - <<'PY'
import csv, glob, collections, os
flt = os.environ["FILTER"].split("|")
vals = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pk_[0-9]*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pk_1/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in vals.items():
    if any(s in k for s in flt):
        d = dur.get(k)
        print(k[:110], "  avg %.1f us under the counters" % (sum(d) / len(d) / 1e3) if d else "")
        for c, x in sorted(v.items()):
            print("   %-34s %14.0f  (n=%d)" % (c, sum(x) / len(x), len(x)))
PY
rm -rf gpurun_out/pk_[0-9]*
