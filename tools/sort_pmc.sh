#!/bin/bash
# HBM bytes of the radix-sort kernels (1e9 u32 keys): separate FETCH_SIZE / WRITE_SIZE passes
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sortpmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/sort_once.py <<PY
import sys
sys.path.insert(0, "$ROOT")
import torch
from vexcl_amd import ops
k = ops.fill_hash(torch.empty(10**9, dtype=torch.int32, device="cuda:0"), 42)
ops.sort(k, unsigned=True); torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o pmc --output-format csv -- python /tmp/sort_once.py > $OUT/$c.log 2>&1
  echo "$c exit $?"
done
python - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True)
    if not f: print(c, "no csv"); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"].split("(")[0][-60:]
        agg[name][0] += 1; agg[name][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        print(c, k, "calls", n, "per-call raw", v / n)
PY
