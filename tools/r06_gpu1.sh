#!/bin/bash
# round 6, call 1: the yardstick the review asked for (rocprim::radix_sort_keys on the bench's 1e9 keys, timed and traced per kernel),
# our sort beside it, and the bench line on this box before anything changes.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 tools/build/r06_rocprim_sort 1e9 5 > $OUT/r06_sort_rocprim.json 2> $OUT/r06_sort_rocprim.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_rp -o rp --output-format csv -- $ROOT/tools/build/r06_rocprim_sort 1e9 3 > $OUT/r06_sort_rocprim_under_rocprof.log 2>&1
cp /tmp/prof_rp/rp_kernel_stats.csv $OUT/r06_sort_rocprim_kernel_stats.csv 2>/dev/null
cd $ROOT
timeout 600 python tools/r05_sort_time.py 1e9 > $OUT/r06_sort_time_before.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_n1_before.log 2> $OUT/r06_bench_n1_before.err
cat $OUT/r06_sort_rocprim.json; head -12 $OUT/r06_sort_rocprim_kernel_stats.csv | cut -c1-220; tail -3 $OUT/r06_sort_time_before.log | cut -c1-600; tail -c 1500 $OUT/r06_bench_n1_before.log
