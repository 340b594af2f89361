#!/bin/bash
# HBM bytes of the FFT row kernel from the PMC counters: separate passes for FETCH_SIZE and WRITE_SIZE
# (MI355X_MICROARCH.md, section HBM), counters only with --kernel-trace.  Summary: tools/fft_pmc_summary.py.
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_fft
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o pmc --output-format csv -- python $ROOT/tools/fft_pmc_target.py > $OUT/fetch.log 2>&1
echo "fetch exit $?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o pmc --output-format csv -- python $ROOT/tools/fft_pmc_target.py > $OUT/write.log 2>&1
echo "write exit $?"
cd $ROOT
python tools/fft_pmc_summary.py $OUT
