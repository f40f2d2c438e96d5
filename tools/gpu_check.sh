#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests per file, then smoke, then a short bench.
# A step that times out (exit 124/137) stops the script: no further GPU work after a hang.
set -u
mkdir -p gpurun_out
run() {  # name timeout cmd...
  local name=$1 tmo=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/summary.log
  timeout -k 10 "$tmo" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/summary.log
  tail -n 25 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/summary.log; exit 1; fi
  return 0
}
: > gpurun_out/summary.log
for step in "$@"; do
  case $step in
    ops)   run test_ops 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider ;;
    model) run test_model 600 python -m pytest tests/test_gpu_model.py -q -m gpu -s -p no:cacheprovider ;;
    all)   run test_all 900 python -m pytest tests -q -m gpu -p no:cacheprovider ;;
    smoke) run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) run bench 600 python bench.py --steps 20 --warmup 5 ;;
    benchq) run benchq 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ;;
    prof)  cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
           run prof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline ;;
    *) echo "unknown step $step" ;;
  esac
done
cat gpurun_out/summary.log
