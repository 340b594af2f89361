#!/bin/bash
# round 6, call 4: the pull step with the flag raised by a relaxed store; timeline per workgroup; acquire A/B; gap sweep of the headline
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export VEXHIP_IPC_TIMEOUT_MS=5000
run() { local label=$1; shift
  env "$@" DIST_ONLY=${ONLY:-push,pull} DIST_OUT=$OUT/r06_dist_step_$label.json timeout 300 python tools/r06_dist_step.py > $OUT/r06_dist_step_$label.log 2>&1
  echo "== $label: $(grep -E 'device_us' $OUT/r06_dist_step_$label.log | sed 's/halo //' | cut -c1-150 | tr '\n' '|')"; grep -o '"[a-z_]*equals[a-z_]*": [a-z]*' $OUT/r06_dist_step_$label.log | tr '\n' ' '; echo
}
ONLY=push,pull,events,parts run default A=1
ONLY=pull run acquire0 VEXHIP_HALO_ACQUIRE=0
ONLY=pull run acquire1 VEXHIP_HALO_ACQUIRE=1
ONLY=pull run two_launches VEXHIP_HALO_TWO_LAUNCHES=1
for e in 2 4 16; do ONLY=pull run edge$e VEXHIP_HALO_EDGE_PLANES=$e; done
DIST_MODE=pull timeout 300 python tools/r06_halo_timeline.py > $OUT/r06_halo_timeline_pull.log 2>&1; cat $OUT/r06_halo_timeline_pull.log | cut -c1-400
DIST_MODE=push timeout 300 python tools/r06_halo_timeline.py > $OUT/r06_halo_timeline_push.log 2>&1; cat $OUT/r06_halo_timeline_push.log | cut -c1-400
timeout 600 python tools/r06_xy_gap.py > $OUT/r06_xy_gap.log 2>&1; tail -4 $OUT/r06_xy_gap.log | cut -c1-600
