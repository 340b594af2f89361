#!/bin/bash
# round 6, call 26: the runs product (entries decoded at set-up): the SpMV suite, smoke, the 27-point rows
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_spmv.py -q -x -m gpu 2>&1 | tail -12 > gpurun_out/r06_gpu26_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > gpurun_out/r06_gpu26_smoke.log
WIDEN=s27 timeout 900 python tools/r06_widen_probe.py > gpurun_out/r06_gpu26_probe.log 2>&1
cp gpurun_out/r06_widen_probe.json gpurun_out/r06_widen_probe_runs.json
tail -12 gpurun_out/r06_gpu26_tests.log; cat gpurun_out/r06_gpu26_smoke.log | cut -c1-200; grep "^{" gpurun_out/r06_gpu26_probe.log | cut -c1-330
