#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_spmv.py -q -x -m gpu --timeout=1200 -k "two_dimensional or grid_product or plane_product or storage_by_grid" > gpurun_out/r05_2d_tests.log 2>&1; echo "pytest exit $?"; tail -25 gpurun_out/r05_2d_tests.log
