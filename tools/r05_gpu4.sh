#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { echo "== $*"; env "$@" VEXHIP_IPC_TIMEOUT_MS=5000 timeout 600 python tools/r05_dist_step.py 2>&1 | grep -E "^halo: |^local part|Traceback|Error" | cut -c1-300; python -c "
import json; d=json.load(open('gpurun_out/r05_dist_step.json')); print({k:v for k,v in d.items() if 'equals' in k})"; }
run A=1
run VEXHIP_HALO_EDGE_PLANES=4
run VEXHIP_HALO_EDGE_PLANES=12
run VEXHIP_HALO_EDGE_PLANES=16
run VEXHIP_HALO_NO_PUSH=1
