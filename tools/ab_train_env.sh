#!/bin/bash
# A/B of one environment switch on the graphed TRAINING step, on ONE box: A = ENVV (e.g. "CSSHIP_NO_DVDK=1") set, B = unset.
# ENVV="K=V" tools/ab_train_env.sh [bench_train args]
for i in 1 2 3; do
  echo "A ($ENVV): $(env $ENVV timeout -k 10 300 python tools/bench_train.py --graph --steps 30 "$@" 2>/dev/null | tail -1)"
  echo "B (unset): $(timeout -k 10 300 python tools/bench_train.py --graph --steps 30 "$@" 2>/dev/null | tail -1)"
done
