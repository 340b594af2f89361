#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_gputests_late.log; head -2 gpurun_out/r05_gputests_late.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_of_measured_copy'], [ (k[:28], v.get('fp32',{}).get('ms')) for k,v in d['secondary'].items() if '7-point' in k])"
