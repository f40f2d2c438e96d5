#!/bin/bash
# A/B of two library builds on the graphed TRAINING step, on ONE box: A = $A_LIB, B = the in-tree library.
# A_LIB=build/exp/lib_x.so tools/ab_train_lib.sh [bench_train args]
for i in 1 2 3; do
  echo "A ($A_LIB): $(timeout -k 10 300 python tools/ab_train.py "$A_LIB" --graph --steps 30 "$@" 2>/dev/null | tail -1)"
  echo "B (in-tree): $(timeout -k 10 300 python tools/bench_train.py --graph --steps 30 "$@" 2>/dev/null | tail -1)"
done
