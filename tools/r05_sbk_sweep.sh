#!/bin/bash
# by-key single pass: geometry sweep (waves per workgroup x rows per wave, wave scan on DPP or shuffles) on section k of examples/roofline
# (1e8 (int, double) pairs), with the exact-arithmetic test in front of each geometry.  One process per point (the knobs are read once).
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_sbk_sweep.log
: > $out
make -C tests/cpp -s build/primitives_tests >> $out 2>&1
for cfg in "16 4 0" "16 4 1" "8 4 1" "8 4 0" "8 8 1" "8 6 1" "4 8 1" "12 4 1" "16 2 1" "8 2 1" "16 3 1" "4 4 1"; do
  set -- $cfg
  export VEXCL_SBK_WAVES=$1 VEXCL_SBK_ROWS=$2 VEXCL_SBK_DPP=$3
  t=$(timeout 60 tests/cpp/build/primitives_tests by_key_single_pass_against_three_phases 2>&1 | grep -E "failures" | tail -1)
  r=$(timeout 60 examples/build/roofline 1000000000 k 2>&1 | grep -E '"row"' | grep -v tree | sed -E 's/.*"row": "([a-z_]*).*"ms": ([0-9.]*).*/\1 \2/' | tr '\n' ' ')
  echo "waves $1 rows $2 dpp $3 | test: $t | $r" >> $out
done
cat $out
