#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_distributed.py -q -x -m gpu 2>&1 | tail -3
