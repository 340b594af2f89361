#!/bin/bash
# Round 5: Reductor order `tagged` (default): exact stress + reduce rows, beside `relaxed` on the same box; then the whole GPU suite.
mkdir -p gpurun_out; export TMPDIR=/tmp
o=gpurun_out/r05_gpu2.log; : > $o
make -C tests/cpp -s build/vector_tests >> $o 2>&1
for m in tagged relaxed tagged relaxed; do
  echo "-- VEXCL_REDUCTOR_ORDER=$m" >> $o
  VEXCL_REDUCTOR_ORDER=$m timeout 300 examples/build/roofline 1000000000 e 2>&1 | grep -i "reduce" | sed "s/^/[$m] /" >> $o
done
VEXCL_REDUCTOR_ORDER=tagged VEX_TEST_REDUCE_STRESS=100000 timeout 600 tests/cpp/build/vector_tests 2>&1 | grep -E "reduction|failures|FAIL|CHECK" >> $o
cat $o
timeout 2400 python -m pytest tests -m gpu -q -x --timeout=1500 > gpurun_out/r05_gputests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r05_gputests.log
tail -5 gpurun_out/r05_gputests.log
