#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { echo "== $*"; env "$@" VEXHIP_IPC_TIMEOUT_MS=5000 timeout 300 python tools/r05_halo_timeline.py 2>&1 | grep -E "^device|lower|main" | cut -c1-330; }
run A=1
run VEXHIP_HALO_TWO_PASS=1
run A=1
run VEXHIP_HALO_TWO_PASS=1
