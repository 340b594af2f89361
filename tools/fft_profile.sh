#!/bin/bash
# rocprofv3 kernel trace of tools/fft_bench.py (per-pass durations of fft_lines_kernel), then a PMC pass for the HBM
# bytes of the 2^24-point fp64 transform.  Output under gpurun_out/prof_fft; copy the summaries into profiles/.
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_fft
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o fft --output-format csv -- python $ROOT/tools/fft_bench.py > $OUT/trace.log 2>&1
echo "trace exit $?"
cd $ROOT
find $OUT -name "*kernel_stats.csv" | head -3
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f"
