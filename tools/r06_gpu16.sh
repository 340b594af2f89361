#!/bin/bash
# round 6, call 16: the pair product's HALO role counting only the edge workgroups
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
DIST_VARIABLE=1 DIST_ONLY=pull,events DIST_OUT=gpurun_out/r06_dist_step_variable_512.json timeout 600 python tools/r06_dist_step.py > gpurun_out/r06_gpu16_step.log 2>&1
timeout 900 python -m pytest tests/test_gpu_distributed.py -q -x -m gpu 2>&1 | tail -15 > gpurun_out/r06_gpu16_tests.log
(cd tests/cpp && timeout 600 ./build/spmv_tests 2>&1 | tail -4) > gpurun_out/r06_gpu16_cpp.log
VEXCL_LOGICAL_DEVICES=2 timeout 600 examples/build/spmv_headline 512 20 --devices 1 --check > gpurun_out/r06_gpu16_headline2.log 2>&1
tail -3 gpurun_out/r06_gpu16_tests.log; grep -v "^{" gpurun_out/r06_gpu16_step.log | tail -8; cat gpurun_out/r06_gpu16_cpp.log; tail -30 gpurun_out/r06_gpu16_headline2.log
