#!/bin/bash
# The round's profile set, on the GPU box (via gpurun):  tools/profile_round.sh <tag>   e.g. r02_v3
#   1. rocprofv3 --kernel-trace --stats of `bench.py` (forward + PIT) and of `bench.py --mode train`
#   2. PMC, each counter set in its own pass (never together with the tracing domains): FETCH_SIZE, WRITE_SIZE,
#      MFMA busy, LDS
# Summaries land in gpurun_out/<tag>_*: copy the ones to be judged into profiles/.
set -u
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-step"
for mode in fwd train; do
  rm -rf gpurun_out/prof_$mode
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$mode -- $B --mode $mode > gpurun_out/${tag}_bench_$mode.log 2>&1
  rc=$?; echo "stats $mode rc=$rc"
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
  f=$(ls gpurun_out/prof_$mode/*/*kernel_stats.csv 2>/dev/null | tail -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_${mode}_kernel_stats.csv
  # the same averages over the launches of the REPLAYED graph only (back to back: < 1.5 us to both neighbours) - the
  # stats file above also counts the eager warm-up / capture / timeline-probe launches of the same command, which start
  # on a drained GPU with cold caches and read ~4-8 % longer
  t=$(ls gpurun_out/prof_$mode/*/*kernel_trace.csv 2>/dev/null | tail -1)
  [ -n "$t" ] && TRACE="$t" OUT="gpurun_out/${tag}_${mode}_replay_kernel_avg.csv" python - <<'PY'
import collections, csv, os
rows = sorted(csv.DictReader(open(os.environ["TRACE"])), key=lambda r: int(r["Start_Timestamp"]))
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
acc = collections.defaultdict(list)
for i in range(1, len(rows) - 1):
    if st[i] - en[i - 1] < 1500 and st[i + 1] - en[i] < 1500:
        acc[rows[i]["Kernel_Name"]].append(en[i] - st[i])
with open(os.environ["OUT"], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "ReplayCalls", "AverageNs", "MinNs", "MaxNs"])
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), round(sum(v) / len(v), 1), min(v), max(v)])
PY
  grep '^{' gpurun_out/${tag}_bench_$mode.log > gpurun_out/${tag}_bench_${mode}_under_rocprof.json
done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY"; do
  i=$((i + 1))
  rm -rf gpurun_out/pmc_$i
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-step > gpurun_out/pmc_$i.log 2>&1
  rc=$?; echo "pmc [$set] rc=$rc"
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
done
python - <<PY
import collections, csv, glob, json
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_[0-9]*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
def shorten(k):
    short = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return short[:short.find("(")] if "(" in short else short
dur = {}
# launch durations: the averages over the REPLAYED launches (back to back, as in the timed region); the --stats file (which also
# counts the eager warm-up / capture launches) only for kernels the replay filter did not see
for f in glob.glob("gpurun_out/${tag}_fwd_kernel_stats.csv") + glob.glob("gpurun_out/${tag}_fwd_replay_kernel_avg.csv"):
    for r in csv.DictReader(open(f)):
        dur[shorten(r["Name"])] = float(r["AverageNs"])
out = {}
for k, v in vals.items():
    if "at::" in k or "rocclr" in k or "elementwise" in k:
        continue
    short = shorten(k)
    m = {c: sum(x) / len(x) for c, x in v.items()}
    e = {"launches": max(len(x) for x in v.values())}
    e.update({c: round(x, 1) for c, x in m.items()})
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts half of wide coalesced reads (MI355X_MICROARCH.md, HBM): x2
        e["hbm_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
    if short in dur and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        # busy cycles are summed over the 1024 SIMDs (16 per 16x16x32 bf16 MFMA); wall = average launch x 2.4 GHz
        e["avg_launch_us"] = round(dur[short] / 1e3, 2)
        e["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * dur[short] * 2.4), 4)
    if m.get("SQ_WAVE_CYCLES"):
        e["valu_active_frac_of_wave_cycles"] = round(m.get("SQ_ACTIVE_INST_VALU", 0.0) / m["SQ_WAVE_CYCLES"], 4)
        e["lds_active_frac_of_wave_cycles"] = round(m.get("SQ_ACTIVE_INST_LDS", 0.0) / m["SQ_WAVE_CYCLES"], 4) if "SQ_ACTIVE_INST_LDS" in m else None
    out[short] = e
import hashlib, os
hsh = hashlib.sha1()
for fn in sorted(os.listdir("css_amd/csrc")):
    if fn.endswith((".hip", ".hpp", ".inc")) and fn != "fft_tables.inc":
        hsh.update(open(os.path.join("css_amd/csrc", fn), "rb").read())
out["_meta"] = {"csrc_sha": hsh.hexdigest()[:16], "command": "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-step",
                "note": "bench.py quotes traffic / mfma_busy from this file only while csrc_sha matches the tree"}
json.dump(out, open("gpurun_out/${tag}_pmc.json", "w"), indent=1)
for k, e in sorted(((k, e) for k, e in out.items() if k != "_meta"), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0)):
    print(f"{k[:56]:56s} n={e['launches']:4d} hbm {e.get('hbm_bytes_per_launch', 0) / 1e6:8.2f} MB  mfma_busy {e.get('mfma_busy_frac')}")
PY
