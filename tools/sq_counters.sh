#!/bin/bash
# SQ / TCP counters of the headline kernel (diagnostic): one rocprofv3 --pmc pass per group
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- $CMD > $OUT/g$i.log 2>&1
  echo "group $i ($grp) exit $?"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sell8v_kernel" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg): print("%-34s %.4g per launch" % (k, agg[k][1] / agg[k][0]))
PY
