#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export VEXHIP_IPC_TIMEOUT_MS=5000
timeout 600 python tools/r06_sort_ab.py 1e9 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x --timeout=900 2>&1 | tail -3
run() { local label=$1; shift
  env "$@" DIST_OUT=$OUT/r06_dist_step_$label.json timeout 300 python tools/r06_dist_step.py > $OUT/r06_dist_step_$label.log 2>&1
  echo "== $label: $(grep -E 'device_us' $OUT/r06_dist_step_$label.log | sed 's/halo //' | cut -c1-140 | tr '\n' '|')"; grep -o '"[a-z_]*equals[a-z_]*": [a-z]*' $OUT/r06_dist_step_$label.log | tr '\n' ' '; echo
}
run f64_640 DIST_GRID=640 DIST_ONLY=pull,events,parts
python - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch, bench
from vexcl_amd import ops
import unstructured as U
dev = torch.device("cuda:0"); m = 20000000
x = ops.fill_hash(torch.empty(m, dtype=torch.float64, device=dev), 42); y = torch.empty_like(x)
out = {}
for name in ("random16", "banded16"):
    p, c, v = U.MAKERS[name](m, dev)
    for xl in (None, "nt"):
        if xl: os.environ["VEXHIP_SELL_XLOAD"] = xl
        else: os.environ.pop("VEXHIP_SELL_XLOAD", None)
        A = ops.SpMat(p, c, v)
        A.apply(x, y)
        out["%s, gathers of x %s" % (name, xl or "plain")] = round(min(bench.timed_events(torch, lambda: A.apply(x, y), 10) for _ in range(3)), 4)
        del A
    del p, c, v
os.environ.pop("VEXHIP_SELL_XLOAD", None)
print(json.dumps(out)); json.dump(out, open("gpurun_out/r06_sell_xload_ab.json", "w"), indent=1)
PY
timeout 3000 python -m pytest tests -m gpu -q --timeout=1500 -x > $OUT/r06_gputests_mid4.log 2>&1; echo "pytest exit $?" >> $OUT/r06_gputests_mid4.log; grep -E "passed|failed|exit" $OUT/r06_gputests_mid4.log | tail -3
