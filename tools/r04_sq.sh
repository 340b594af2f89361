#!/bin/bash
# SQ / TCP / TA / TCC counters of the plane (GRID=512) or grid product, the one-pass set-up kernel and the pair product (diagnostic; GRID env): one rocprofv3 --pmc pass per group
# over tools/r03_pmc_target.py; per-kernel averages to gpurun_out/r04_sq_summary_${GRID:-512}.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sq4_${GRID:-512}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/r03_pmc_target.py"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- $CMD > $OUT/g$i.log 2>&1
  echo "group $i ($grp) exit $?"
done
python - <<PY > $ROOT/gpurun_out/r04_sq_summary_${GRID:-512}.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for tag in ("sell8_plane_kernel", "sell8_grid_kernel", "grid_build_kernel", "sell8_march_kernel", "sell8_pair_kernel", "reduce_stage1"):
            if tag in k:
                a = agg[tag][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
                break
for tag in agg:
    c = {k: v[1] / v[0] for k, v in agg[tag].items()}
    print("==", tag)
    for k in sorted(c): print("  %-36s %.6g per launch (%d launches)" % (k, c[k], agg[tag][k][0]))
    if "SQ_WAVES" in c and "GRBM_GUI_ACTIVE" in c and "SQ_WAVE_CYCLES" in c:
        w, cyc = c["SQ_WAVES"], c["GRBM_GUI_ACTIVE"] / 8
        print("  -> per wave: VMEM %.1f  VALU %.0f  SALU %.0f  LDS %.0f  SMEM %.1f  TCP accesses %.0f | occupancy %.1f waves/CU | wave time: parked %.0f%%, issue stall %.0f%%, issuing %.0f%% | TA busy %.0f%% | L2 hit %.0f%% | HBM read %.2f GB | LDS idx active/wave %.0f conflicts/wave %.0f" % (
            (c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]) / w, c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_LDS"] / w, c["SQ_INSTS_SMEM"] / w,
            c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / w, c["SQ_WAVE_CYCLES"] * 4 / (cyc * 256), 100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
            100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c.get("TA_BUSY_avr", 0) / cyc,
            100 * c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)), c.get("TCC_EA0_RDREQ_sum", 0) * 128 / 1e9,
            c.get("SQ_LDS_IDX_ACTIVE", 0) / w, c.get("SQ_LDS_BANK_CONFLICT", 0) / w))
PY
cat $ROOT/gpurun_out/r04_sq_summary_${GRID:-512}.txt
