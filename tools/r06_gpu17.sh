#!/bin/bash
# round 6, call 17: by-key look-back with the value folded serially from the anchor (reproducible floating-point carries)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(cd tests/cpp && timeout 900 ./build/primitives_tests 2>&1 | tail -12) > gpurun_out/r06_gpu17_prim.log
timeout 600 examples/build/roofline 1000000000 k > gpurun_out/r06_gpu17_roofline_k.log 2>&1
for p in scan_by_key reduce_by_key; do (timeout 300 oracle/_ref/$p 2>&1 | tail -2) >> gpurun_out/r06_gpu17_ref.log; done
cat gpurun_out/r06_gpu17_prim.log; cat gpurun_out/r06_gpu17_roofline_k.log | cut -c1-400; cat gpurun_out/r06_gpu17_ref.log
