#!/bin/bash
# SQ / TCP / TA / TCC counters of the SpMV kernels at 512^3 (diagnostic): one rocprofv3 --pmc pass per group over
# tools/pmc_headline.py (default SpMat on the Poisson and the variable-coefficient matrix, 32-bit columns, CSR); per-kernel averages to gpurun_out/r02_sq_summary.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
CMD="python $ROOT/tools/pmc_headline.py"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- $CMD > $OUT/g$i.log 2>&1
  echo "group $i ($grp) exit $?"
done
python - <<PY > $ROOT/gpurun_out/r02_sq_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sell8_pair_kernel" in k:
            targs = [a.strip() for a in k.split("sell8_pair_kernel<")[1].split(">")[0].split(",")]          # <V, W, VCODED, DICT>
            k = ("sell8_pair_kernel_vcoded" if targs[2] in ("true", "1") else "sell8_pair_kernel_values") + ("_dict" if targs[3] in ("true", "1") else "")
        for tag in ("sell8_pair_kernel_vcoded_dict", "sell8_pair_kernel_values_dict", "sell8_pair_kernel_vcoded", "sell8_pair_kernel_values", "sell_pair_kernel", "csr_stream2_kernel", "sell8v_kernel", "sell8_kernel",
                    "sell_kernel", "csr_stream_kernel", "hell_kernel", "reduce_stage1"):
            if tag in k:
                a = agg[tag][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
                break
for tag in agg:
    c = {k: v[1] / v[0] for k, v in agg[tag].items()}
    print("==", tag)
    for k in sorted(c): print("  %-36s %.6g per launch (%d launches)" % (k, c[k], agg[tag][k][0]))
    if "SQ_WAVES" in c and "GRBM_GUI_ACTIVE" in c and "SQ_WAVE_CYCLES" in c:
        w, cyc = c["SQ_WAVES"], c["GRBM_GUI_ACTIVE"] / 8
        print("  -> per wave: VMEM %.1f  VALU %.0f  SALU %.0f  LDS %.0f  TCP accesses %.0f | occupancy %.1f waves/CU | wave time: parked %.0f%%, issue stall %.0f%%, issuing %.0f%% | TA busy %.0f%% | L2 hit %.0f%% | HBM read %.2f GB" % (
            (c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]) / w, c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_LDS"] / w,
            c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / w, c["SQ_WAVE_CYCLES"] * 4 / (cyc * 256), 100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
            100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c.get("TA_BUSY_avr", 0) / cyc,
            100 * c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)), c.get("TCC_EA0_RDREQ_sum", 0) * 128 / 1e9))
PY
cat $ROOT/gpurun_out/r02_sq_summary.txt
