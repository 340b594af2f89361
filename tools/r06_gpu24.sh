#!/bin/bash
# round 6, call 24: runs of three diagonals in wide value-coded slices (27-point): probe + tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
WIDEN=s27 timeout 900 python tools/r06_widen_probe.py > gpurun_out/r06_gpu24_probe.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_spmv.py -q -x -m gpu -k "wide_value or vector_added or sell8" 2>&1 | tail -5 > gpurun_out/r06_gpu24_tests.log
grep "^{" gpurun_out/r06_gpu24_probe.log | cut -c1-500; tail -3 gpurun_out/r06_gpu24_probe.log | cut -c1-300; cat gpurun_out/r06_gpu24_tests.log
