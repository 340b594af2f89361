#!/bin/bash
# round 5: the fp32 plane product -- the spmv suite, the 512^3 time, the kernel trace
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_spmv.py tests/test_cpp_api.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python tools/r05_fp32.py 2>&1 | grep "^f"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof32 -o p32 -- python /root/repo/tools/r05_fp32.py > /dev/null 2>&1
f=$(ls /tmp/prof32/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -6 "$f" > /root/repo/gpurun_out/r05_fp32_kernel_stats.csv && head -4 "$f"
