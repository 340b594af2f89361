#!/bin/bash
# round 6, call 18: by-key reproducibility A/B (old header against the serial value fold), long-run rows
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(cd tests/cpp && timeout 900 ./build/primitives_tests_old 2>&1 | grep -v "^\[ ok" | tail -12) > gpurun_out/r06_gpu18_prim_old.log
(cd tests/cpp && timeout 900 ./build/primitives_tests 2>&1 | grep -v "^\[ ok" | tail -12) > gpurun_out/r06_gpu18_prim.log
timeout 600 examples/build/roofline 1000000000 k > gpurun_out/r06_gpu18_roofline_k.log 2>&1
echo OLD; cat gpurun_out/r06_gpu18_prim_old.log; echo NEW; cat gpurun_out/r06_gpu18_prim.log; grep "^{" gpurun_out/r06_gpu18_roofline_k.log | cut -c1-300
