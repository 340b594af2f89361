#!/bin/bash
# round 6, call 7: the one-launch step through the grid product (384^3 strips) and in float; make_inline through the library product;
# the unstructured rows with the slices dealt by XCD; the GPU suite.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export VEXHIP_IPC_TIMEOUT_MS=5000
timeout 300 tests/cpp/build/spmv_tests > $OUT/r06_cpp_spmv_tests.log 2>&1; echo "spmv_tests rc $?"; grep -v "^\[ ok" $OUT/r06_cpp_spmv_tests.log | head -20
run() { local label=$1; shift
  env "$@" DIST_OUT=$OUT/r06_dist_step_$label.json timeout 300 python tools/r06_dist_step.py > $OUT/r06_dist_step_$label.log 2>&1
  echo "== $label: $(grep -E 'device_us' $OUT/r06_dist_step_$label.log | sed 's/halo //' | cut -c1-140 | tr '\n' '|')"; grep -o '"[a-z_]*equals[a-z_]*": [a-z]*' $OUT/r06_dist_step_$label.log | tr '\n' ' '; tail -2 $OUT/r06_dist_step_$label.log | grep -i "error\|assert" | head -3; echo
}
run f64_512 DIST_ONLY=pull,events,parts
run f32_512 DIST_DTYPE=f32 DIST_ONLY=pull,events,parts
run f64_384 DIST_GRID=384 DIST_ONLY=pull,events,parts
run f64_640 DIST_GRID=640 DIST_ONLY=pull,events,parts
timeout 600 ./examples/build/roofline 1000000000 k > $OUT/r06_roofline_inline.log 2>&1; cut -c1-200 $OUT/r06_roofline_inline.log
VEXCL_LOGICAL_DEVICES=2 timeout 600 examples/build/spmv_headline 384 100 --devices 1 --check 2>&1 | cut -c1-420
timeout 3000 python -m pytest tests -m gpu -q --timeout=1500 -x > $OUT/r06_gputests_mid2.log 2>&1; echo "pytest exit $?" >> $OUT/r06_gputests_mid2.log; grep -E "passed|failed|exit" $OUT/r06_gputests_mid2.log | tail -3
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_n1_mid3.log 2> $OUT/r06_bench_n1_mid3.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_n1_mid3.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_general','value_csr_stream') if k in d}, d['roofline']['frac'])
for k,v in d['secondary'].items():
    if 'unstructured' in k:
        r=v.get('roofline',{})
        print(k[:60], v.get('ms'), 'rr', v.get('slices_dealt_round_robin_ms'), v.get('storage'), 'frac', r.get('frac'), 'traffic/bytes', r.get('traffic_over_bytes_per_launch'), r.get('bound_ms'), r.get('traffic_source'), v.get('error'))
PY
