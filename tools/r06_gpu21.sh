#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_distributed.py -q -x -m gpu 2>&1 | tail -25 > gpurun_out/r06_gpu21_tests.log
tail -25 gpurun_out/r06_gpu21_tests.log
