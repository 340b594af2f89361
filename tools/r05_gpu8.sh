#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for d in 512 256 171 128 512; do
  VEXHIP_PLANE_DEPTH=$d timeout 300 python bench.py --gpus 1 --steps 40 --warmup 10 --no-secondary --no-pmc --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $d', d['ms_per_step'], d['roofline']['frac'], d['sustained']['ms_per_step'], d['roofline'].get('plane',{}).get('depth'))"
done
