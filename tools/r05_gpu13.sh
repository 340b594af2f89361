#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
echo "== 168 registers"; timeout 300 python tools/r05_fp32_sizes.py 384 500 640 2>&1 | grep "f32"
echo "== 199 registers"; VEXHIP_LIBRARY=/root/repo/gpurun_in/libvexhip_G32U.so timeout 300 python tools/r05_fp32_sizes.py 384 500 640 2>&1 | grep "f32"
done
