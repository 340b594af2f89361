#!/bin/bash
# Round-2 evidence in one gpurun call: the bench line (un-profiled), the same command under rocprofv3 --kernel-trace,
# the C++ front end under the trace, SQ counters of the round-2 kernels, the sort trace.
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py > $O/r02_bench_n1.log 2>&1; echo "bench exit $?"; tail -c 600 $O/r02_bench_n1.log
bash tools/r02_profile.sh > $O/r02_profile.log 2>&1; echo "profile exit $?"
bash tools/r02_sq.sh > $O/r02_sq.log 2>&1; echo "sq exit $?"
bash tools/r02_sort_trace.sh > $O/r02_sort_trace.log 2>&1; echo "sort exit $?"; tail -8 $O/r02_sort_trace.log
timeout 600 ./examples/build/roofline > $O/r02_roofline_cpp.log 2>&1; echo "roofline exit $?"; grep "^{" $O/r02_roofline_cpp.log | cut -c1-200
