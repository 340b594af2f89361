#!/bin/bash
# occupancy / run-length sensitivity of the march product (v8): LDS floor per workgroup => workgroups per CU
for lds in 0 53000 65000; do
  echo "== VEXHIP_MARCH_LDS=$lds"; VEXHIP_MARCH_LDS=$lds timeout 200 python tools/r03_march2_ab.py 2>/dev/null
done
for run in 16 64; do
  echo "== VEXHIP_MARCH_RUN=$run"; VEXHIP_MARCH_RUN=$run timeout 200 python tools/r03_march2_ab.py 2>/dev/null
done
