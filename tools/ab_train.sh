#!/bin/bash
# A/B of the TRAINING step on one box: A = $A_LIB (another build of the library), B = the in-tree library.
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then cmd="python tools/ab_bench.py $A_LIB"; else cmd="python bench.py"; fi
    r=$(timeout -k 10 200 $cmd --mode train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
    echo "$v train ms/step $r"
  done
done
