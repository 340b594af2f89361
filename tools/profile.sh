#!/bin/bash
# rocprofv3 passes for the bench command: kernel-trace stats, then PMC passes
# (FETCH_SIZE and WRITE_SIZE need separate passes: TCC slots, MI355X_MICROARCH.md).
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary $BENCH_ARGS"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- $CMD > $OUT/trace.log 2>&1
echo "trace exit $?"
CMD2="python $ROOT/tools/pmc_target.py"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc --output-format csv -- $CMD2 > $OUT/pmc_fetch.log 2>&1
echo "pmc fetch exit $?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc --output-format csv -- $CMD2 > $OUT/pmc_write.log 2>&1
echo "pmc write exit $?"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_l2 -o pmc --output-format csv -- $CMD2 > $OUT/pmc_l2.log 2>&1
echo "pmc l2 exit $?"
cd $ROOT
find $OUT -name "*.csv" | head -30
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt | tail -40
