#!/bin/bash
# A/B of the training step under environment overrides: tools/ab_train_env.sh "VAR=val" ["VAR2=val2" ...]; the first leg is the default
for i in 1 2; do
  for e in "X_=0" "$@"; do
    r=$(env $e timeout -k 10 200 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
    echo "[$e] train ms/step $r"
  done
done
