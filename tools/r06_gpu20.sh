#!/bin/bash
# round 6, call 20: flat grid plans -- SpMV tests, distributed tests, C++ spmv tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_distributed.py -q -x -m gpu 2>&1 | tail -25 > gpurun_out/r06_gpu20_tests.log
(cd tests/cpp && timeout 900 ./build/spmv_tests 2>&1 | grep -v "^\[ ok" | tail -6) > gpurun_out/r06_gpu20_cpp.log
tail -25 gpurun_out/r06_gpu20_tests.log; cat gpurun_out/r06_gpu20_cpp.log
