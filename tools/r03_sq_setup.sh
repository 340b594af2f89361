#!/bin/bash
# SQ counters of the set-up kernels (tools/r03_setup_profile.py) -> gpurun_out/r03_sq_setup.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sqs; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/r03_setup_profile.py"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- $CMD > $OUT/g$i.log 2>&1
  echo "group $i exit $?"
done
python - <<PY > $ROOT/gpurun_out/r03_sq_setup.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for tag in ("sell8v_fill_kernel", "analyze_fused_kernel"):
            if tag in k:
                a = agg[tag + " vgpr=" + r.get("VGPR_Count", "?") + " lds=" + r.get("LDS_Block_Size", "?")][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
                break
for tag in sorted(agg):
    c = {k: v[1] / v[0] for k, v in agg[tag].items()}
    print("==", tag)
    for k in sorted(c): print("  %-36s %.6g per launch (%d launches)" % (k, c[k], agg[tag][k][0]))
    if "SQ_WAVES" in c and "GRBM_GUI_ACTIVE" in c and "SQ_WAVE_CYCLES" in c:
        w, cyc = c["SQ_WAVES"], c["GRBM_GUI_ACTIVE"] / 8
        print("  -> per wave: VMEM %.1f  VALU %.0f  SALU %.0f  LDS %.0f | occupancy %.1f waves/CU | wave time: parked %.0f%%, issue stall %.0f%%, issuing %.0f%% | VALU busy %.0f%% | LDS busy %.0f%% conflicts %.0f%% of it | cycles %.0f" % (
            (c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]) / w, c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_LDS"] / w,
            c["SQ_WAVE_CYCLES"] * 4 / (cyc * 256), 100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
            100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
            100 * c.get("SQ_ACTIVE_INST_VALU", 0) / (cyc * 256), 100 * c.get("SQ_LDS_IDX_ACTIVE", 0) / (cyc * 256), 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 1)), cyc))
PY
cat $ROOT/gpurun_out/r03_sq_setup.txt
