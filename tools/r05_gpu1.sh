#!/bin/bash
# Round 5, first GPU call: (1) the sort's rank schemes (exact test + 1e9 timing), (2) Reductor order modes: exact stress in each,
# then the reduce rows of examples/roofline in each, (3) the GPU test files touched so far.  Every process under timeout.
mkdir -p gpurun_out; export TMPDIR=/tmp
o=gpurun_out/r05_gpu1.log; : > $o
echo "== sort rank schemes: exact" >> $o
timeout 600 python -m pytest tests/test_gpu_primitives.py -q -x -m gpu -k "sort" --timeout=500 >> $o 2>&1; echo "pytest sort exit $?" >> $o
echo "== sort timing" >> $o
timeout 600 python tools/r05_sort_time.py 1e9 >> $o 2>&1; echo "sort time exit $?" >> $o
echo "== reductor order modes" >> $o
make -C tests/cpp -s build/vector_tests >> $o 2>&1
for m in release relaxed two_launch; do
  echo "-- VEXCL_REDUCTOR_ORDER=$m vector_tests" >> $o
  VEXCL_REDUCTOR_ORDER=$m VEX_TEST_REDUCE_STRESS=100000 timeout 600 tests/cpp/build/vector_tests 2>&1 | grep -E "reduction|failures|FAIL|CHECK" >> $o
  for rep in 1 2; do
    VEXCL_REDUCTOR_ORDER=$m timeout 300 examples/build/roofline 1000000000 e 2>&1 | grep -i "reduce" | sed "s/^/[$m] /" >> $o
  done
done
tail -60 $o
