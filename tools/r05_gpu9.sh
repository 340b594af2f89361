#!/bin/bash
# round 5: the fp32 plane product on both stored forms -- the spmv suite, the 512^3 time
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_spmv.py tests/test_cpp_api.py -x -q -m gpu 2>&1 | tail -8
timeout 300 python tools/r05_fp32.py 2>&1 | grep "^f"
VEXHIP_SPMAT_NO_GRID_BUILD=1 VEXHIP_NO_GRID_BUILD=1 timeout 300 python tools/r05_fp32.py 2>&1 | grep "^f32"
