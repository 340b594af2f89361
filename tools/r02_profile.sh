#!/bin/bash
# Round 2 evidence: rocprofv3 kernel trace of the bench command (the same process prints the bench line) and of the
# C++ front end; copies go to profiles/r02_*.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-sustained > $OUT/bench_under_rocprof.log 2>&1
echo "bench trace exit $?"; tail -c 1500 $OUT/bench_under_rocprof.log
rocprofv3 --kernel-trace --stats -d $OUT/trace_cpp -o cpp --output-format csv -- $ROOT/examples/build/spmv_headline 512 50 > $OUT/cpp_under_rocprof.log 2>&1
echo "cpp trace exit $?"; grep "^{" $OUT/cpp_under_rocprof.log
find $OUT -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -12 "$f" | cut -c1-220; done
