#!/bin/bash
# round 6, call 15: the pull step between processes (plane + general strips), the stand-in step for the variable-coefficient strip
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_distributed.py -q -x -m gpu 2>&1 | tail -15 > gpurun_out/r06_gpu15_tests.log
DIST_VARIABLE=1 DIST_ONLY=pull,events,parts DIST_OUT=gpurun_out/r06_dist_step_variable_512.json timeout 600 python tools/r06_dist_step.py > gpurun_out/r06_gpu15_step.log 2>&1
tail -5 gpurun_out/r06_gpu15_tests.log; tail -12 gpurun_out/r06_gpu15_step.log
