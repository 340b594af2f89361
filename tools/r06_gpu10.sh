#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export VEXHIP_IPC_TIMEOUT_MS=5000
run() { local label=$1; shift
  env "$@" DIST_OUT=$OUT/r06_dist_step_$label.json timeout 300 python tools/r06_dist_step.py > $OUT/r06_dist_step_$label.log 2>&1
  echo "== $label: $(grep -E 'device_us' $OUT/r06_dist_step_$label.log | sed 's/halo //' | cut -c1-160 | tr '\n' '|')"; grep -o '"[a-z_]*equals[a-z_]*": [a-z]*' $OUT/r06_dist_step_$label.log | tr '\n' ' '; echo
}
run f64_640 DIST_GRID=640 DIST_ONLY=events,parts
run f64_512 DIST_ONLY=pull,events,parts
timeout 3000 python -m pytest tests -m gpu -q --timeout=1500 > $OUT/r06_gputests_mid5.log 2>&1; echo "pytest exit $?" >> $OUT/r06_gputests_mid5.log; grep -E "passed|failed|exit|FAILED" $OUT/r06_gputests_mid5.log | tail -6
VEXHIP_DEBUG=1 timeout 300 examples/build/spmv_headline 256 20 2>&1 | grep -E "vexhip\]|selection" | cut -c1-420
timeout 900 bash tools/r06_sq_bykey.sh > $OUT/r06_sq_bykey.log 2>&1; grep -E "^==|^  ->" $OUT/r06_sq_bykey.txt | cut -c1-330
