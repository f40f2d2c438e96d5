#!/bin/bash
# rocprofv3 kernel stats of the Conformer-large forward (tools/bench_large.py, 18 blocks): gpurun_out/<tag>_large_kernel_stats.csv
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_large
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_large -- python tools/bench_large.py > gpurun_out/${tag}_bench_large.log 2>&1
echo "rc=$?"
f=$(ls gpurun_out/prof_large/*/*kernel_stats.csv 2>/dev/null | tail -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_large_kernel_stats.csv
