#!/bin/bash
# round 6, call 28: the library as built from the final sources -- smoke, the SpMV and distributed suites
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
timeout 2400 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_distributed.py tests/test_cpp_api.py -q -x -m gpu 2>&1 | tail -3
