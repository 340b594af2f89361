#!/bin/bash
# round 6, call 23: bench on two ranks sharing the GPU (the N > 1 path end to end: partition, transports, validation, timing)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export VEXHIP_IPC_TIMEOUT_MS=5000
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --one-device --backend gloo > gpurun_out/r06_bench_n2_one_device.log 2> gpurun_out/r06_bench_n2_one_device.err
echo "bench n2 exit $?"
grep "^{" gpurun_out/r06_bench_n2_one_device.log | cut -c1-2500; grep -v "amdgpu.ids" gpurun_out/r06_bench_n2_one_device.err | tail -5
