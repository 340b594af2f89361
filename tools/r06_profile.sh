#!/bin/bash
# Round-6 evidence: the driver's bench command un-profiled and under the kernel trace (the dominant kernel's average must agree with
# roofline.avg_launch_ms), the GPU suite, smoke.  Outputs under gpurun_out/, copied to profiles/ by hand.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export VEXHIP_IPC_TIMEOUT_MS=5000
timeout 3000 python -m pytest tests -m gpu -q --timeout=1500 > $OUT/r06_gputests_final.log 2>&1; echo "pytest exit $?" >> $OUT/r06_gputests_final.log; grep -E "passed|failed|exit|FAILED" $OUT/r06_gputests_final.log | tail -6
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/r06_smoke.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_n1_final.log 2> $OUT/r06_bench_n1_final.err; echo "bench exit $?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b --output-format csv -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary > $OUT/r06_bench_under_rocprof.log 2>&1
cp /tmp/prof_b/b_kernel_stats.csv $OUT/r06_bench_kernel_stats.csv 2>/dev/null
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o a --output-format csv -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r06_bench_all_rows_under_rocprof.log 2>&1
cp /tmp/prof_a/a_kernel_stats.csv $OUT/r06_bench_all_rows_kernel_stats.csv 2>/dev/null
cd $ROOT
python - <<PY
import json
for l in open("$OUT/r06_bench_n1_final.log"):
    if l.startswith("{"):
        d = json.loads(l); print("bench:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
        for k, v in d.get("secondary", {}).items():
            if "one pass" in k or "sort" in k: print("  ", k[:80], {a: v[a] for a in ("ms", "frac", "one_pass", "bits_equal_b_minus_product", "gkeys_per_s") if a in v})
PY
head -3 $OUT/r06_bench_kernel_stats.csv | cut -c1-200
