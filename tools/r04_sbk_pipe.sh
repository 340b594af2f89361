#!/bin/bash
# by-key single pass as a pipeline (VEXCL_SBK_PIPELINE=1, scan_by_key.hpp vexcl_sbk_pipe): exact-arithmetic test, then the 1e8 rows
# of examples/roofline (section k) with and without it.  Every process is bounded: a look-back that never ends must not hold the box.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/r04_sbk_pipe.log
: > $out
export VEXCL_SBK_PIPELINE=1
echo "== test (pipeline ${VEXCL_SBK_PIPE_WAVES:-7} x ${VEXCL_SBK_PIPE_ROWS:-4})" >> $out
timeout 25 tests/cpp/build/primitives_tests by_key_single_pass_against_three_phases >> $out 2>&1; echo "rc $?" >> $out
echo "== roofline k, pipeline" >> $out
timeout 25 examples/build/roofline 1000000000 k >> $out 2>&1; echo "rc $?" >> $out
if [ -n "$SBK_BASELINE" ]; then
  unset VEXCL_SBK_PIPELINE
  echo "== roofline k, one tile per workgroup (default)" >> $out
  timeout 25 examples/build/roofline 1000000000 k >> $out 2>&1; echo "rc $?" >> $out
fi
cat $out
