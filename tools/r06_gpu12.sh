#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp; SORT_CHECKS=0 SORT_MODES=${SORT_MODES:-8} timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/sortchain_prof -o t --output-format csv -- python $ROOT/tools/r06_sort_chain.py > $OUT/r06_sort_chain_prof.log 2>&1
tail -3 $OUT/r06_sort_chain_prof.log | cut -c1-200
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$OUT/sortchain_prof/t_kernel_trace.csv")))
agg=collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"][:90],r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    print(k, len(v), "avg %.1f us min %.1f max %.1f"%(sum(v)/len(v), min(v), max(v)))
PY
