#!/bin/bash
# A/B of one bench.py switch on ONE box: A = bench.py $FLAG, B = bench.py.   FLAG="--no-w2-chunked" tools/ab_flag.sh [bench args]
common="--steps 30 --warmup 5 --no-cpu-baseline --no-train-step $*"
show='import json,sys
d=json.loads(sys.stdin.readline()); r=d["roofline"]
print("%s ms/step %.4f  chain %.2f attn %.2f oc %.2f stft %.2f pit %.2f" % (sys.argv[1], d["ms_per_step"], r["avg_launch_us"], r["attention"]["avg_launch_us"], r["outproj_conv"]["avg_launch_us"], r["stft"]["avg_launch_us"], r["pit"]["avg_launch_us"]))'
for i in 1 2 3; do
  timeout -k 10 200 python bench.py $common $FLAG 2>/dev/null | python -c "$show" A
  timeout -k 10 200 python bench.py $common 2>/dev/null | python -c "$show" B
done
