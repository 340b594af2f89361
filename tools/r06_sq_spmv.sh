#!/bin/bash
# SQ / TCP / TCC counters of the SpMV products the review asks about -> gpurun_out/r06_sq_spmv.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sqs6; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/r06_sq_spmv_driver.py"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- $CMD > $OUT/g$i.log 2>&1
  echo "group $i exit $?"
done
python - <<PY > $ROOT/gpurun_out/r06_sq_spmv.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
tags = ("sell8_plane_kernel", "sell8_grid_kernel", "sell8_grid_f32_kernel", "sell8v_runs_kernel")
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for tag in tags:
            if tag + "<" in k or tag + "(" in k:
                key = tag + " grid=" + r.get("Grid_Size", "?") + " wg=" + r.get("Workgroup_Size", "?") + " vgpr=" + r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?")) + " lds=" + r.get("LDS_Block_Size", "?")
                a = agg[key][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
                break
for tag in sorted(agg):
    c = {k: v[1] / v[0] for k, v in agg[tag].items()}
    print("==", tag)
    for k in sorted(c): print("  %-36s %.6g per launch (%d launches)" % (k, c[k], agg[tag][k][0]))
    if "SQ_WAVES" in c and "GRBM_GUI_ACTIVE" in c and "SQ_WAVE_CYCLES" in c:
        w, cyc = c["SQ_WAVES"], c["GRBM_GUI_ACTIVE"] / 8
        print("  -> per wave: VMEM %.1f  VALU %.0f  SALU %.0f  SMEM %.0f  LDS %.0f | occupancy %.1f waves/CU | wave time: parked %.0f%%, issue stall %.0f%%, issuing %.0f%% | VALU busy %.0f%% | HBM read %.2f GB written %.2f GB | L1 accesses %.3g, L1->L2 reads %.3g | cycles %.0f" % (
            (c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]) / w, c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c.get("SQ_INSTS_SMEM", 0) / w, c["SQ_INSTS_LDS"] / w,
            c["SQ_WAVE_CYCLES"] * 4 / (cyc * 256), 100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
            100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
            100 * c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (cyc * 256 * 4), c.get("TCC_EA0_RDREQ_sum", 0) * 128 / 1e9, c.get("TCC_EA0_WRREQ_sum", 0) * 64 / 1e9,
            c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0), c.get("TCP_TCC_READ_REQ_sum", 0), cyc))
PY
cat $ROOT/gpurun_out/r06_sq_spmv.txt | grep -E "^==|->"
