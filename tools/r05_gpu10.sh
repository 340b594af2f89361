#!/bin/bash
# round 5: the one-pass build with the next piece's entries requested one trip ahead -- parity, then the set-up times
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_spmv.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/r05_setup.py 2>&1 | grep -E "^(poisson|diffusion)" | sed 's/setup_ms_all.*csr_bytes/.../'
for w in 2 3; do echo "== VEXHIP_GRID_BUILD_WGS=$w"; VEXHIP_GRID_BUILD_WGS=$w timeout 300 python tools/r05_setup.py 2>&1 | grep -E "^poisson" | sed 's/setup_ms_all.*csr_bytes/.../'; done
