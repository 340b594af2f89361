#!/bin/bash
# round 5: set-up of the storage by grid line -- workgroups per CU, rows staged at a time (A/B build in gpurun_in/)
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { timeout 300 python tools/r05_setup.py 2>&1 | grep -E "^(poisson|diffusion)" | sed 's/setup_ms_all.*csr_bytes/.../'; }
echo "== piece 512, default"; run
for w in 2 3; do echo "== piece 512, VEXHIP_GRID_BUILD_WGS=$w"; VEXHIP_GRID_BUILD_WGS=$w run; done
export VEXHIP_LIBRARY=/root/repo/gpurun_in/libvexhip_P256.so
for w in 4 6 7; do echo "== piece 256, VEXHIP_GRID_BUILD_WGS=$w"; VEXHIP_GRID_BUILD_WGS=$w run; done
