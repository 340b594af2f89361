#!/bin/bash
# SQ counters of the radix-sort kernels (1e9 u32 keys): what bounds histogram and scatter (diagnostic)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sortsq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/sort_once.py <<PY
import sys
sys.path.insert(0, "$ROOT")
import torch
from vexcl_amd import ops
k = ops.fill_hash(torch.empty(10**9, dtype=torch.int32, device="cuda:0"), 42)
ops.sort(k, unsigned=True); torch.cuda.synchronize()
PY
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- python /tmp/sort_once.py > $OUT/g$i.log 2>&1
  echo "group $i exit $?"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for tag in ("radix_hist_kernel", "radix_scatter_kernel", "lookback_scan_kernel"):
            if tag in k:
                a = agg[tag][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for tag, c in agg.items():
    g = {k: v[1] / v[0] for k, v in c.items()}
    w = g.get("SQ_WAVES", 1); cyc = g.get("GRBM_GUI_ACTIVE", 0) / 8
    print("==", tag, "waves %.3g cycles/XCD %.3g" % (w, cyc))
    for k in sorted(g): print("   %-28s %.5g   per wave %.4g" % (k, g[k], g[k] / w))
    if "SQ_WAVE_CYCLES" in g:
        print("   occupancy %.1f waves/CU  wait_any %.0f%% wait_inst %.0f%% active %.0f%% valu_active %.0f%% lds_active %.0f%%" % (
            g["SQ_WAVE_CYCLES"] * 4 / (cyc * 256), 100 * g["SQ_WAIT_ANY"] / g["SQ_WAVE_CYCLES"], 100 * g["SQ_WAIT_INST_ANY"] / g["SQ_WAVE_CYCLES"],
            100 * g["SQ_ACTIVE_INST_ANY"] / g["SQ_WAVE_CYCLES"], 100 * g["SQ_ACTIVE_INST_VALU"] / g["SQ_WAVE_CYCLES"], 100 * g["SQ_ACTIVE_INST_LDS"] / g["SQ_WAVE_CYCLES"]))
PY
