#!/bin/bash
# full GPU suite + smoke + bench on the current tree
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export VEXHIP_IPC_TIMEOUT_MS=5000
timeout 3000 python -m pytest tests -m gpu -q --timeout=1500 > $OUT/r06_gputests_a.log 2>&1; echo "pytest exit $?" >> $OUT/r06_gputests_a.log; grep -E "passed|failed|exit|FAILED" $OUT/r06_gputests_a.log | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python bench.py > $OUT/r06_bench_n1_a.log 2> $OUT/r06_bench_n1_a.err; echo "bench exit $?"; tail -c 600 $OUT/r06_bench_n1_a.err
python - <<PY
import json
for l in open("$OUT/r06_bench_n1_a.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("avg_launch_ms"))
        for s in d.get("secondary", []):
            if isinstance(s, dict): print("  ", (s.get("row") or s.get("name") or "?")[:70], s.get("ms"), s.get("frac"))
PY
