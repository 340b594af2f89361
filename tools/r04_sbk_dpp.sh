#!/bin/bash
# by-key single pass with the wave scan on DPP (VEXCL_SBK_DPP=1): exact-arithmetic test, the 1e8 rows of examples/roofline (section k)
# with it and, on the same box, without it.  Every process is bounded.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/r04_sbk_dpp.log
: > $out
echo "== test, VEXCL_SBK_DPP=1" >> $out
VEXCL_SBK_DPP=1 timeout 20 tests/cpp/build/primitives_tests by_key_single_pass_against_three_phases >> $out 2>&1; echo "rc $?" >> $out
echo "== roofline k, VEXCL_SBK_DPP=1" >> $out
VEXCL_SBK_DPP=1 timeout 20 examples/build/roofline 1000000000 k 2>&1 | grep -v "^$" >> $out; echo "rc $?" >> $out
echo "== roofline k, VEXCL_SBK_DPP=0" >> $out
VEXCL_SBK_DPP=0 timeout 20 examples/build/roofline 1000000000 k 2>&1 | grep -v "^$" >> $out; echo "rc $?" >> $out
cut -c1-200 $out
