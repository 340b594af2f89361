#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3 4; do
timeout 1200 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu 2>&1 | tail -3
done
