#!/bin/bash
# round 6, call 37: DistSpMat's set-up through the library's split -- the distributed GPU tests, bench on two and on FOUR ranks sharing the GPU
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export VEXHIP_IPC_TIMEOUT_MS=5000 BENCH_DUMP_AFTER=240
timeout 1200 python -m pytest tests/test_gpu_distributed.py -q -x -m gpu 2>&1 | tail -3
for w in 2 4; do
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port 2955$w bench.py --gpus $w --steps 20 --warmup 5 --one-device --backend gloo --no-secondary --no-cpu-baseline --no-pmc > gpurun_out/r06_bench_n${w}_one_device_split.log 2> gpurun_out/r06_bench_n${w}_one_device_split.err
echo "bench n$w exit $?"
grep "^{" gpurun_out/r06_bench_n${w}_one_device_split.log | cut -c1-260; grep "transports, fastest\|Timeout\|rror" gpurun_out/r06_bench_n${w}_one_device_split.err | tail -3 | cut -c1-300
done
