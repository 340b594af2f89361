#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SORT_MODES=${SORT_MODES:--1,8,9} timeout 900 python tools/r06_sort_chain.py > $OUT/r06_sort_chain.log 2>&1; echo "exit $?"; tail -7 $OUT/r06_sort_chain.log | cut -c1-300
