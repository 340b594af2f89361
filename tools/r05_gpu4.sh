#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { echo "== $*"; env "$@" VEXHIP_IPC_TIMEOUT_MS=5000 timeout 600 python tools/r05_dist_step.py 2>&1 | grep -E "^halo: |^local part alone|Traceback|Error" | cut -c1-300; }
run A=1
run VEXHIP_LIBRARY=$PWD/vexcl_amd/lib/libvexhip_halo4.so
run VEXHIP_LIBRARY=$PWD/vexcl_amd/lib/libvexhip_halo4.so VEXHIP_HALO_DEPTH=24
run VEXHIP_LIBRARY=$PWD/vexcl_amd/lib/libvexhip_halo4.so VEXHIP_HALO_DEPTH=24 VEXHIP_HALO_EDGE_PLANES=4
run VEXHIP_LIBRARY=$PWD/vexcl_amd/lib/libvexhip_halo4.so VEXHIP_HALO_DEPTH=16
run A=1
