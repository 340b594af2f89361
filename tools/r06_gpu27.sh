#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_spmv.py -q -x -m gpu -k "wide_value or vector_added or sell8" 2>&1 | tail -4 > gpurun_out/r06_gpu27_tests.log
timeout 900 python tools/r06_stencil19.py > gpurun_out/r06_gpu27.log 2>&1
cat gpurun_out/r06_gpu27_tests.log; grep "^{" gpurun_out/r06_gpu27.log | cut -c1-400 | head -4; tail -2 gpurun_out/r06_gpu27.log | cut -c1-300
