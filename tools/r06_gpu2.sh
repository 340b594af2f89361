#!/bin/bash
# round 6, call 2: the one-launch step behind vex::SpMat (C++ test + the headline example on two logical devices), the stand-in step
# (push against pull), the Infinity Cache probe, the distributed GPU tests.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export VEXHIP_IPC_TIMEOUT_MS=5000
timeout 300 tests/cpp/build/spmv_tests > $OUT/r06_cpp_spmv_tests.log 2>&1; echo "spmv_tests rc $?" | tee -a $OUT/r06_cpp_spmv_tests.log
timeout 600 python tools/r06_dist_step.py > $OUT/r06_dist_step.log 2>&1; echo "dist_step rc $?"
grep -E "device_us|equals|timed_out" $OUT/r06_dist_step.log | head -30
VEXCL_LOGICAL_DEVICES=2 timeout 600 examples/build/spmv_headline 512 100 --devices 1 --check > $OUT/r06_headline_2dev.log 2>&1; echo "headline rc $?"
cut -c1-700 $OUT/r06_headline_2dev.log
timeout 200 tools/build/r06_mall_probe > $OUT/r06_mall_probe.json 2> $OUT/r06_mall_probe.err; cat $OUT/r06_mall_probe.json
timeout 1200 python -m pytest tests/test_gpu_distributed.py tests/test_cpp_api.py -m gpu -x -q --timeout=900 > $OUT/r06_gputests_dist.log 2>&1; tail -5 $OUT/r06_gputests_dist.log
