#!/bin/bash
# Ceiling of cross-launch overlap: the encoder's launches with the dependencies between consecutive
# launches CUT (CSSHIP_DEBUG_NODEP=N: dealt round-robin to N streams; results are wrong, timing is what is measured),
# eager launches, against the same eager step with the dependencies in place.  One box, alternating.
for i in 1 2; do
  for n in 0 1 2 3 6; do
    r=$(CSSHIP_DEBUG_NODEP=$n timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-step --no-graph 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
    echo "streams=$n (0 = dependencies kept, one stream) eager ms/step $r"
  done
done
