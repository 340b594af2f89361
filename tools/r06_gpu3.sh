#!/bin/bash
# round 6, call 3: the step with the signal folded into the launch's last workgroup; A/B of the acquire behind the flag, of the
# second launch, of the planes per edge chunk; the C++ test of the one-launch step on two logical devices; the headline example.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export VEXHIP_IPC_TIMEOUT_MS=5000
timeout 300 tests/cpp/build/spmv_tests > $OUT/r06_cpp_spmv_tests.log 2>&1; echo "spmv_tests rc $?" | tee -a $OUT/r06_cpp_spmv_tests.log; grep -v "^\[ ok" $OUT/r06_cpp_spmv_tests.log | head -20
run() { # label, env...
  local label=$1; shift
  env "$@" DIST_ONLY=${ONLY:-push,pull} DIST_OUT=$OUT/r06_dist_step_$label.json timeout 300 python tools/r06_dist_step.py > $OUT/r06_dist_step_$label.log 2>&1
  echo "== $label: $(grep -E 'device_us' $OUT/r06_dist_step_$label.log | sed 's/halo //' | cut -c1-150 | tr '\n' '|')"; grep -o '"[a-z_]*equals[a-z_]*": [a-z]*' $OUT/r06_dist_step_$label.log | tr '\n' ' '; echo
}
ONLY=push,pull,events,parts run default A=1
run two_launches VEXHIP_HALO_TWO_LAUNCHES=1
ONLY=pull run acquire0 VEXHIP_HALO_ACQUIRE=0
ONLY=pull run acquire1 VEXHIP_HALO_ACQUIRE=1
for e in 2 4 16; do ONLY=pull run edge$e VEXHIP_HALO_EDGE_PLANES=$e; done
ONLY=pull run edge4_acq0 VEXHIP_HALO_EDGE_PLANES=4 VEXHIP_HALO_ACQUIRE=0
VEXCL_LOGICAL_DEVICES=2 timeout 600 examples/build/spmv_headline 512 100 --devices 1 --check > $OUT/r06_headline_2dev.log 2>&1; echo "headline rc $?"
cut -c1-560 $OUT/r06_headline_2dev.log
timeout 300 examples/build/spmv_headline 512 200 2>&1 | cut -c1-400
