#!/bin/bash
# HBM bytes of the CCSR product at 512^3 (examples/build/roofline), separate FETCH_SIZE / WRITE_SIZE passes
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ccsrpmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o pmc --output-format csv -- $ROOT/examples/build/roofline > $OUT/$c.log 2>&1
  echo "$c exit $?"
done
python - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        if "ccsr" in n or "sell8" in n or "stencil" in n.lower() or "conv" in n:
            k = n.split("(")[0][-50:]
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        print(c, k, "calls", n, "GB/call", round(v / n * 1024 / 1e9 * (2 if c == "FETCH_SIZE" else 1), 3))
PY
