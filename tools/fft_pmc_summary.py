#!/usr/bin/env python
"""Per-launch HBM bytes of fft_lines_kernel from the two PMC passes of tools/fft_pmc.sh.  FETCH_SIZE is calibrated on
the 2 GiB stream read by reduce_stage1 in the same pass (gfx950 tallies 128-byte requests at 64 B); units: KiB."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
def rows(sub):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r
res = {}
for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    per = defaultdict(list)
    for r in rows(sub):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        key = "reduce_stage1" if "reduce_stage1" in name else ("fft_lines_kernel grid=%s" % r.get("Grid_Size", "?")) if "fft_lines_kernel" in name else None
        if key:
            per[key].append(float(r["Counter_Value"]))
    res[counter] = {k: {"launches": len(v), "avg_KiB": sum(v) / len(v)} for k, v in per.items()}
cal = res["FETCH_SIZE"].get("reduce_stage1", {}).get("avg_KiB")
factor = (2 * 1024 * 1024) / cal if cal else None          # 2 GiB in KiB over what the counter reported
summary = {"fetch_calibration_factor": factor, "kernels": {}}
for k, v in res["FETCH_SIZE"].items():
    if k.startswith("fft"):
        w = res["WRITE_SIZE"].get(k, {}).get("avg_KiB")
        summary["kernels"][k] = {"fetched_bytes": v["avg_KiB"] * 1024 * (factor or 1), "written_bytes": (w or 0) * 1024, "launches": v["launches"]}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
