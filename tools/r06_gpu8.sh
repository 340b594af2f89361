#!/bin/bash
# round 6, call 8: fixes of call 7 (grid walks of the plan, plane size from the strips), make_inline rows, sort with the in-kernel
# second ranking, PMC counters of the banded matrix, the GPU suite
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export VEXHIP_IPC_TIMEOUT_MS=5000
timeout 300 tests/cpp/build/spmv_tests > $OUT/r06_cpp_spmv_tests.log 2>&1; echo "spmv_tests rc $?"; grep -v "^\[ ok" $OUT/r06_cpp_spmv_tests.log | head -20
run() { local label=$1; shift
  env "$@" DIST_OUT=$OUT/r06_dist_step_$label.json timeout 300 python tools/r06_dist_step.py > $OUT/r06_dist_step_$label.log 2>&1
  echo "== $label: $(grep -E 'device_us' $OUT/r06_dist_step_$label.log | sed 's/halo //' | cut -c1-140 | tr '\n' '|')"; grep -o '"[a-z_]*equals[a-z_]*": [a-z]*' $OUT/r06_dist_step_$label.log | tr '\n' ' '; tail -2 $OUT/r06_dist_step_$label.log | grep -i "error\|assert" | head -3; echo
}
run f64_640 DIST_GRID=640 DIST_ONLY=pull,events,parts
run f64_768 DIST_GRID=768 DIST_ONLY=pull,events,parts
timeout 600 ./examples/build/roofline 1000000000 i > $OUT/r06_roofline_inline.log 2>&1; cut -c1-200 $OUT/r06_roofline_inline.log
timeout 600 python tools/r06_sort_ab.py 1e9 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x --timeout=900 2>&1 | tail -3
# L2 / TCP counters of the banded matrix's product (what bounds it once x comes from the caches)
cd /tmp
for grp in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE"; do
  for m in banded16 random16; do
    PMC_ONLY=$m UNSTRUCTURED_ROWS=2e7 GRID=8 timeout 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$m -o pmc --output-format csv -- python $ROOT/tools/pmc_headline.py > /tmp/pmc_$m.log 2>&1
    python - "$m" "$grp" <<'PY'
import csv, glob, sys, collections
m, grp = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("/tmp/pmc_%s/**/*counter_collection.csv" % m, recursive=True):
    for r in csv.DictReader(open(f)):
        if "sell_kernel" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print(m, {k: round(v[1] / v[0], 1) for k, v in agg.items()}, flush=True)
PY
    rm -rf /tmp/pmc_$m
  done
done 2>&1 | tee $OUT/r06_unstructured_counters.log
cd $ROOT
timeout 3000 python -m pytest tests -m gpu -q --timeout=1500 -x > $OUT/r06_gputests_mid3.log 2>&1; echo "pytest exit $?" >> $OUT/r06_gputests_mid3.log; grep -E "passed|failed|exit" $OUT/r06_gputests_mid3.log | tail -3
