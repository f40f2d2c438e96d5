cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf gpurun_out/pw_$i
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pw_$i -- python tools/bench_wgrad_group.py --iters 5 > gpurun_out/pw_$i.log 2>&1
  echo "pmc [$set] rc=$?"
done
python - <<'PY'
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pw_[0-9]*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in vals.items():
    if "group" in k or "Lb1ELi128ELb1" in k or "true, 128, true" in k:
        print(k[:80])
        for c, x in sorted(v.items()):
            print("   %-34s %14.0f  (n=%d)" % (c, sum(x)/len(x), len(x)))
PY
rm -rf gpurun_out/pw_[0-9]*
