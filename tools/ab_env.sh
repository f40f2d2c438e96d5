#!/bin/bash
# A/B of one environment switch on ONE box: A = ENVV (e.g. "CSSHIP_HEADPIT_ROWS=64") set, B = unset.  ENVV="K=V" tools/ab_env.sh [bench args]
common="--steps 30 --warmup 5 --no-cpu-baseline --no-train-step $*"
show='import json,sys
d=json.loads(sys.stdin.readline()); r=d["roofline"]
print("%s ms/step %.4f  chain %.2f attn %.2f oc %.2f stft %.2f pit %.2f (head %.2f)" % (sys.argv[1], d["ms_per_step"], r["avg_launch_us"], r["attention"]["avg_launch_us"], r["outproj_conv"]["avg_launch_us"], r["stft"]["avg_launch_us"], r["pit"]["avg_launch_us"], r["pit"].get("head_pit_us", 0)))'
for i in 1 2 3; do
  env $ENVV timeout -k 10 200 python bench.py $common 2>/dev/null | python -c "$show" A
  timeout -k 10 200 python bench.py $common 2>/dev/null | python -c "$show" B
done
