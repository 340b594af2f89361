#!/bin/bash
# round 6, call 5: sort A/B (store policy of the scatter, loads in flight of the histogram); the headline with the library's placement
# of x and y; the C++ rows (vex::vector allocated through vexhip_malloc); the whole GPU suite on the new allocator.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export VEXHIP_IPC_TIMEOUT_MS=5000
for v in "A=0" "VEXHIP_SORT_STORE_AUX=2" "VEXHIP_SORT_STORE_AUX=18" "VEXHIP_SORT_STORE_AUX=17" "VEXHIP_SORT_HIST_UNROLL=12" "VEXHIP_SORT_HIST_UNROLL=3" "VEXHIP_SORT_HIST_UNROLL=12 VEXHIP_SORT_STORE_AUX=2"; do
  env $v timeout 300 python tools/r06_sort_ab.py 1e9 2>&1 | tail -1
done | tee $OUT/r06_sort_ab.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_n1_mid.log 2> $OUT/r06_bench_n1_mid.err; tail -c 3000 $OUT/r06_bench_n1_mid.log | head -c 1200; echo
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_n1_mid.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_general','value_csr_stream') if k in d}, d['roofline']['frac'], d['config'].get('vector_placement'), d['blocks']['ms_per_step'])
PY
timeout 300 examples/build/spmv_headline 512 200 2>&1 | cut -c1-330
VEXHIP_MALLOC_STAGGER=0 timeout 300 examples/build/spmv_headline 512 200 2>&1 | cut -c1-330
timeout 600 ./examples/build/roofline 1000000000 escipk > $OUT/r06_examples_roofline_cpp.log 2>&1; grep -c row $OUT/r06_examples_roofline_cpp.log; cut -c1-250 $OUT/r06_examples_roofline_cpp.log | head -40
timeout 3000 python -m pytest tests -m gpu -q --timeout=1500 -x > $OUT/r06_gputests_mid.log 2>&1; echo "pytest exit $?" >> $OUT/r06_gputests_mid.log; tail -6 $OUT/r06_gputests_mid.log
