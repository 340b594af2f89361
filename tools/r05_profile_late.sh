#!/bin/bash
# Round-5 evidence, second half of the round (fp32 plane product, pipelined one-pass build): the driver's bench command un-profiled
# and under the kernel trace, the fp32 product under the kernel trace, the set-up trace.  Outputs under gpurun_out/, copied to profiles/.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_bench_n1_late.log 2> $OUT/r05_bench_n1_late.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b --output-format csv -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary > $OUT/r05_bench_under_rocprof_late.log 2>&1
cp /tmp/prof_b/b_kernel_stats.csv $OUT/r05_bench_kernel_stats_late.csv 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o f --output-format csv -- python $ROOT/tools/r05_fp32.py > $OUT/r05_fp32_under_rocprof.log 2>&1
cp /tmp/prof_f/f_kernel_stats.csv $OUT/r05_fp32_kernel_stats.csv 2>/dev/null
cd $ROOT
VEXHIP_SETUP_TRACE=1 timeout 100 python tools/r03_setup_profile.py 512 > $OUT/r05_setup_trace_512_late.log 2>&1
tail -c 600 $OUT/r05_bench_n1_late.log; echo; head -4 $OUT/r05_bench_kernel_stats_late.csv | cut -c1-220; head -4 $OUT/r05_fp32_kernel_stats.csv | cut -c1-220
