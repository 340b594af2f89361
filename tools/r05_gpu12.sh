#!/bin/bash
# round 5: the plane product's walks on planes that are not one tile per CU -- parity, the shapes, the headline
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_spmv.py -x -q -m gpu 2>&1 | tail -3
SWEEP_DIVS=0 timeout 600 python tools/r05_plane_shapes.py 512x640x640 512x768x512 512x384x768 512x1024x256 512x320x1024 512x512x512 512x256x1024 512x512x256 512x512x64 512x128x2048 2>&1 | grep -v amdgpu
