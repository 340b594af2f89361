#!/bin/bash
# per-kernel times of radix sorts, rocprofv3 kernel trace ($1 = tag for the output directory)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sortprof_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/sort_once.py <<PY
import sys
sys.path.insert(0, "$ROOT")
import torch
from vexcl_amd import ops
k = ops.fill_hash(torch.empty(10**9, dtype=torch.int32, device="cuda:0"), 42)
for _ in range(3): ops.sort(k, unsigned=True); torch.cuda.synchronize()
m = 250_000_000
k2 = ops.fill_hash(torch.empty(m, dtype=torch.int32, device="cuda:0"), 43)
v2 = torch.arange(m, dtype=torch.int32, device="cuda:0")
for _ in range(3): ops.sort_by_key(k2, v2, unsigned=True); torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d $OUT -o sort --output-format csv -- python /tmp/sort_once.py > $OUT/log.txt 2>&1
echo "exit $?"
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "radix" in n or "lookback" in n:
        print(n.split("(")[0][-70:], r["Calls"], "avg ms %.4f" % (float(r["AverageNs"]) / 1e6))
PY
