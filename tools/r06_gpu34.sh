#!/bin/bash
# round 6, call 34: bench on FOUR ranks sharing the GPU (middle ranks with two neighbours; transports that time out must be skipped, never fatal)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export VEXHIP_IPC_TIMEOUT_MS=5000
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 4 --steps 20 --warmup 5 --one-device --backend gloo > gpurun_out/r06_bench_n4_one_device.log 2> gpurun_out/r06_bench_n4_one_device.err
echo "bench n4 exit $?"
grep "^{" gpurun_out/r06_bench_n4_one_device.log | cut -c1-600; grep "transports" gpurun_out/r06_bench_n4_one_device.err | tail -4 | cut -c1-400
