#!/bin/bash
# kernel trace of the HEADLINE product only (the secondary rows launch the same kernel template on other matrices: 2-D 16384^2, strips)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o h --output-format csv -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary > $OUT/r05_bench_headline_under_rocprof.log 2>&1
cp /tmp/prof_h/h_kernel_stats.csv $OUT/r05_bench_kernel_stats.csv
head -4 $OUT/r05_bench_kernel_stats.csv | cut -c1-90,250-400; tail -c 600 $OUT/r05_bench_headline_under_rocprof.log | cut -c1-600
