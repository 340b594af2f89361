#!/bin/bash
# round 6, call 22: C++ spmv tests on two logical devices (2-D strips), the 2-D python test, bench on two ranks sharing the GPU
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(cd tests/cpp && VEXCL_LOGICAL_DEVICES=2 timeout 900 ./build/spmv_tests 2>&1 | grep -v "^\[ ok" | tail -8) > gpurun_out/r06_gpu22_cpp.log
timeout 900 python -m pytest tests/test_gpu_spmv.py -q -x -m gpu -k "two_dimensional" 2>&1 | tail -5 > gpurun_out/r06_gpu22_py.log
export VEXHIP_IPC_TIMEOUT_MS=5000
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --one-device > gpurun_out/r06_bench_n2_one_device.log 2> gpurun_out/r06_bench_n2_one_device.err
echo "bench n2 exit $?"
cat gpurun_out/r06_gpu22_cpp.log gpurun_out/r06_gpu22_py.log; grep "^{" gpurun_out/r06_bench_n2_one_device.log | cut -c1-1500; tail -3 gpurun_out/r06_bench_n2_one_device.err
