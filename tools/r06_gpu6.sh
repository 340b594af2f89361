#!/bin/bash
# round 6, call 6: the inline terminal as straight-line code (tests + roofline rows), the bench line with the new unstructured rows
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 tests/cpp/build/spmv_tests > $OUT/r06_cpp_spmv_tests.log 2>&1; echo "spmv_tests rc $?"; grep -v "^\[ ok" $OUT/r06_cpp_spmv_tests.log | head -20
timeout 600 ./examples/build/roofline 1000000000 ik > $OUT/r06_roofline_inline.log 2>&1; cut -c1-200 $OUT/r06_roofline_inline.log
timeout 900 python -m pytest tests/test_reference_suite.py -m gpu -q -x -k "spmv or sparse" > $OUT/r06_ref_spmv.log 2>&1; tail -3 $OUT/r06_ref_spmv.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_n1_mid2.log 2> $OUT/r06_bench_n1_mid2.err; echo "bench rc $?"; tail -3 $OUT/r06_bench_n1_mid2.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_n1_mid2.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_general','value_csr_stream') if k in d}, d['roofline']['frac'])
for k,v in d['secondary'].items():
    if 'unstructured' in k:
        r=v.get('roofline',{})
        print(k[:60], v.get('ms'), v.get('storage'), 'frac', r.get('frac'), 'frac_alg', r.get('frac_of_algorithmic_bytes'), 'traffic/bytes', r.get('traffic_over_bytes_per_launch'), 'traffic/alg', r.get('traffic_over_algorithmic_bytes'), r.get('bound_ms'), v.get('error'))
PY
