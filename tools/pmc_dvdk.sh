#!/bin/bash
# LDS counters of csship_attn_dvdk alone (tools/bench_dvdk.py): bash tools/pmc_dvdk.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES" "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"; do
  i=$((i + 1)); rm -rf gpurun_out/pmcd_$i
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcd_$i -- python tools/bench_dvdk.py "$@" > gpurun_out/pmcd_$i.log 2>&1
  rc=$?; echo "pmc [$set] rc=$rc"; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
done
python - <<'PY'
import collections, csv, glob
vals = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcd_[0-9]*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "dvdk" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(vals.items()):
    print(f"{k:28s} per launch {sum(v) / len(v):14.0f}   ({len(v)} launches)")
PY
