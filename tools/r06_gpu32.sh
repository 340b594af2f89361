#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python tools/r06_contig.py 2>&1 | grep "^{" | cut -c1-300
timeout 900 python tools/r06_stencil19.py 2>&1 | grep "^{" | cut -c1-260 | head -4
timeout 1800 python -m pytest tests/test_gpu_spmv.py -q -x -m gpu 2>&1 | tail -3
