#!/bin/bash
# kernel trace of one sort of 1e9 u32 keys (per-kernel averages) + the timing tool
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sorttrace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/sort_once.py <<PY
import sys
sys.path.insert(0, "$ROOT")
import torch
from vexcl_amd import ops
k = ops.fill_hash(torch.empty(10**9, dtype=torch.int32, device="cuda:0"), 42)
ops.sort(k, unsigned=True); torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d $OUT -o s --output-format csv -- python /tmp/sort_once.py > $OUT/log.txt 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "radix" in r["Name"] or "scan" in r["Name"]:
            print(r["Name"][:70], r["Calls"], "avg %.1f us  total %.2f ms" % (float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
cd $ROOT && python tools/sort_bench.py 2>&1 | grep -v amdgpu
