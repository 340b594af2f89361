#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_distributed.py -q -x -m gpu --timeout=1200 > gpurun_out/r05_dist_tests.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/r05_dist_tests.log
timeout 900 python bench.py --gpus 2 --one-device --steps 20 --warmup 5 --no-secondary > gpurun_out/r05_bench_n2.json 2> gpurun_out/r05_bench_n2.err; echo "bench n2 exit $?"
tail -3 gpurun_out/r05_bench_n2.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05_bench_n2.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], d["config"].get("exchange_transport", "")[:60])
    print(json.dumps(d["distributed"]["transports_tried"])[:1500])
    print(d["roofline"].get("per_gpu"), d["roofline"].get("aggregate"))
except Exception as e:
    print("parse failed", e)
PY
