#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_gputests_late.log; head -2 gpurun_out/r05_gputests_late.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
