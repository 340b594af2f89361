#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for d in 250 125 100 84 72 63 50 36; do echo -n "500 depth $d: "; VEXHIP_GRID32_DEPTH=$d timeout 300 python tools/r05_fp32_sizes.py 500 2>&1 | grep "f32" | sed "s/.*'ms': \([0-9.]*\).*/\1/"; done
for d in 128 96 77 64 55 48 39 32; do echo -n "384 depth $d: "; VEXHIP_GRID32_DEPTH=$d timeout 300 python tools/r05_fp32_sizes.py 384 2>&1 | grep "f32" | sed "s/.*'ms': \([0-9.]*\).*/\1/"; done
for d in 256 128 86 64; do echo -n "512 depth $d: "; VEXHIP_NO_PLANE512=1 VEXHIP_GRID32_DEPTH=$d timeout 300 python tools/r05_fp32_sizes.py 512 2>&1 | grep "f32" | sed "s/.*'ms': \([0-9.]*\).*/\1/"; done
