#!/bin/bash
# Round 5: the whole GPU suite (by-key on DPP, tagged Reductor, lean sort) and the bench line with the new rows.
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout=1500 > gpurun_out/r05_gputests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r05_gputests.log
tail -4 gpurun_out/r05_gputests.log
timeout 900 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; echo "bench exit $?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05_bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "general", d.get("value_general"))
    for k, v in d.get("secondary", {}).items():
        if "unstructured" in k or "sort" in k or k == "error": print(k, json.dumps(v)[:900])
except Exception as e:
    print("bench parse failed", e)
PY
tail -5 gpurun_out/r05_bench.err
