#!/bin/bash
# Timeline of one rank's product step (tools/r02_dist_step.py) under rocprofv3 --kernel-trace: which kernels overlap
ROOT=$(pwd); OUT=$ROOT/gpurun_out/disttrace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o d --output-format csv -- python $ROOT/tools/r02_dist_step.py > $OUT/log.txt 2>&1
echo "exit $?"
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
# find the 300-step "direct issue" region: look for a window of steps in the middle
names = [r[2] for r in rows]
idx = [i for i, nm in enumerate(names) if "sell8_pair" in nm]
mid = idx[len(idx) // 3]
t0 = rows[mid][0]
for r in rows[mid - 1: mid + 14]:
    print("%9.1f us .. %9.1f us  (%6.1f us)  q=%s s=%s  %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[4], r[2]))
PY
