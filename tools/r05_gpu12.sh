#!/bin/bash
# round 5: long grid lines in one segment with many short walks -- the grid tests, then the sizes
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_spmv.py -x -q -m gpu 2>&1 | tail -4
SWEEP_SEGS=1024 SWEEP_DIVS=0 timeout 600 python tools/r05_grid640.py 384 500 576 640 700 768 800 900 1024 2>&1 | grep -v amdgpu
