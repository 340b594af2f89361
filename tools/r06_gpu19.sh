#!/bin/bash
# round 6, call 19: 2-D rows of any length through the grid product on virtual lines (flat plan)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python tools/r06_2d.py > gpurun_out/r06_gpu19_2d.log 2>&1
tail -40 gpurun_out/r06_gpu19_2d.log | cut -c1-700
