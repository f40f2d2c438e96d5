cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_train2
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train2 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --mode train > gpurun_out/pt2.log 2>&1
f=$(ls gpurun_out/prof_train2/*/*kernel_stats.csv | tail -1); cp $f gpurun_out/train2_kernel_stats.csv
