#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_spmv.py -m gpu -q -x --timeout=900 -k "vector_added or plane or grid" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_cpp_api.py -m gpu -q -x --timeout=900 2>&1 | tail -4
timeout 600 examples/build/roofline 100000000 h > $OUT/r06_examples_roofline_inline2.log 2>&1; grep -E "make_inline|b - A|library product" $OUT/r06_examples_roofline_inline2.log | cut -c1-260
