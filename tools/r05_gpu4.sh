#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
VEXHIP_IPC_TIMEOUT_MS=5000 timeout 600 python tools/r05_dist_step.py > gpurun_out/r05_dist_step.log 2>&1; grep -E "device_us" gpurun_out/r05_dist_step.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_distributed.py -q -x -m gpu --timeout=800 2>&1 | tail -3
