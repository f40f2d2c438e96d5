#!/bin/bash
# What each store stream of the attention backward's one-pass kernel costs (on the GPU box, via gpurun).  Needs the
# diagnostic builds of tools/attn_ablate.sh: `tools/attn_ablate.sh build 64 128 256 448` (bit 6: no skewed-dS store,
# bit 7: no P' store, bit 8: no row-major dS store; wrong results by construction - only the kernel's rocprofv3 average
# under the graphed training step means anything).  Run with CSSHIP_NO_POS_FROM_DS=1 to have the skewed store in the path.
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for m in base 64 128 256 448; do
  rm -rf gpurun_out/prof_t
  if [ $m = base ]; then cmd="python tools/bench_train.py --graph --steps 6"; else cmd="python tools/with_lib.py build/abl/lib_$m.so tools/bench_train.py --graph --steps 6"; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t -- $cmd > /dev/null 2>&1
  f=$(ls gpurun_out/prof_t/*/*kernel_stats.csv | tail -1)
  echo "A2_ABL=$m: $(grep -E 'mha_fwd_kernel<8, true, true, true' $f | awk -F, '{print $(NF-4)}')"
done
rm -rf gpurun_out/prof_t
