#!/bin/bash
# A/B on ONE box (devices differ by up to 12 %: never compare numbers across gpurun calls): alternates
#   A = python bench.py <common> $A_ARGS     B = python bench.py <common> $B_ARGS
# three times each and prints ms/step + the in-step kernel timeline of every run.
#   A_ARGS="--no-ln-fold" B_ARGS="" tools/ab.sh [common bench args]
common="--steps 30 --warmup 5 --no-cpu-baseline --no-train-step $*"
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then extra="$A_ARGS"; else extra="$B_ARGS"; fi
    timeout -k 10 200 python ${AB_PY:-bench.py} $common $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$v [$extra] ms/step %.4f  chain %.2f attn %.2f oc %.2f stft %.2f pit %.2f head/us %s' % (d['ms_per_step'], r['avg_launch_us'], r['attention']['avg_launch_us'], r['outproj_conv']['avg_launch_us'], r['stft']['avg_launch_us'], r['pit']['avg_launch_us'], r['step_kernel_us']))"
  done
done
