#!/bin/bash
# The first thing to run on an 8-GPU MI355X node (nothing in rounds 1-6 has run on more than one device):
#   tools/scale_ab.sh [outdir]
# 1. the scaling curve of the headline (train-mode forward + PIT: no collective on the data path) and of the training step
#    (gradient all-reduce over RCCL / xGMI) at N = 1, 2, 4, 8 - one JSON line per N (bench.py starts its own ranks);
# 2. at N = 8, the training step A/B: RCCL (ring / tree as it chooses - the line's train_step.rccl says which algorithm
#    and protocol per bucket size) against the hand-written direct all-reduce over hipIpc peer mappings (--p2p), and a
#    sweep of the number of buckets in flight (--buckets 1 2 4 8).
# Every run is bounded by `timeout`; a run that fails stops the script (no further GPU work behind a hang).
set -u
out=${1:-gpurun_out/scale_ab}
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # name, args...
  local name=$1; shift
  echo "=== $name: python bench.py $*"
  timeout -k 10 900 python bench.py "$@" > "$out/$name.json" 2> "$out/$name.err"
  local rc=$?
  echo "rc=$rc"
  [ $rc -ne 0 ] && { tail -5 "$out/$name.err"; echo "stopping"; exit 1; }
  python - "$out/$name.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d.get("train_step") or {}
print(f"  n_gpus {d['n_gpus']}: {d['metric']}: {d['value'] / 1e6:.3f} M frames/s ({d['ms_per_step']:.3f} ms/step); "
      f"training step {t.get('ms_per_step')} ms, {t.get('value', 0) / 1e6:.3f} M frames/s, collective {t.get('collective')}, "
      f"rccl {json.dumps(t.get('rccl'))[:300]}")
PY
}
for n in 1 2 4 8; do run "fwd_n$n" --gpus $n --steps 20 --warmup 5 --no-cpu-baseline; done
for n in 2 4 8; do run "train_n$n" --gpus $n --steps 20 --warmup 5 --mode train --no-cpu-baseline; done
run "train_n8_p2p" --gpus 8 --steps 20 --warmup 5 --mode train --no-cpu-baseline --p2p
for b in 1 2 8; do
  run "train_n8_buckets$b" --gpus 8 --steps 20 --warmup 5 --mode train --no-cpu-baseline --buckets $b
  run "train_n8_p2p_buckets$b" --gpus 8 --steps 20 --warmup 5 --mode train --no-cpu-baseline --p2p --buckets $b
done
echo "done: $out"
