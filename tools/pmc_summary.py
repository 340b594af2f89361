#!/usr/bin/env python
"""Parses the rocprofv3 CSVs written by tools/profile.sh into
profiles-ready JSON: per-kernel average duration (kernel trace) and HBM bytes
per launch from FETCH_SIZE / WRITE_SIZE, with the gfx950 correction of
MI355X_MICROARCH.md (section HBM): FETCH_SIZE counts 128-byte requests of a wide
coalesced stream at 64 bytes, so it is calibrated on a stream of known size
(reduce_stage1 over 2 GiB) captured in the same pass."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def rows(pattern):
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


def short(name):
    for k in ("spmm_sell8_kernel", "spmm_sell_kernel", "sell8v_kernel", "sell8_kernel", "sell_kernel", "hell_kernel", "csr_stream_kernel", "reduce_stage1", "reduce_stage2", "poisson_kernel",
              "hell_fill_kernel", "fill_hash_kernel"):
        if k in name:
            return k
    return name[:60]


def main():
    out = sys.argv[1]
    res = {}
    # kernel trace durations
    dur = defaultdict(list)
    for r in rows(os.path.join(out, "trace", "**", "*kernel_trace.csv")):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    res["kernel_trace"] = {k: dict(calls=len(v), avg_us=sum(v) / len(v) / 1e3, min_us=min(v) / 1e3, max_us=max(v) / 1e3)
                           for k, v in dur.items()}
    # counters
    ctr = defaultdict(lambda: defaultdict(list))
    for sub in ("pmc_fetch", "pmc_write", "pmc_l2"):
        for r in rows(os.path.join(out, sub, "**", "*counter_collection.csv")):
            ctr[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    avg = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in ctr.items()}
    res["counters_avg_per_dispatch"] = avg
    cal_bytes = (1 << 28) * 8
    if "reduce_stage1" in avg and "FETCH_SIZE" in avg["reduce_stage1"]:
        raw = avg["reduce_stage1"]["FETCH_SIZE"] * 1024        # FETCH_SIZE is in KiB
        res["fetch_calibration"] = dict(known_bytes=cal_bytes, reported_bytes=raw, factor=cal_bytes / raw)
        f = cal_bytes / raw
        for k in ("sell8v_kernel", "sell8_kernel", "sell_kernel", "hell_kernel", "csr_stream_kernel"):
            if k in avg and "FETCH_SIZE" in avg[k]:
                rd = avg[k]["FETCH_SIZE"] * 1024 * f
                wr = avg[k].get("WRITE_SIZE", 0.0) * 1024
                res[k] = dict(hbm_read_bytes=rd, hbm_write_bytes_uncalibrated=wr, hbm_bytes_per_launch=rd + wr,
                              l2_hit_rate=(avg[k].get("TCC_HIT_sum", 0) /
                                           max(1.0, avg[k].get("TCC_HIT_sum", 0) + avg[k].get("TCC_MISS_sum", 0))))
    print(json.dumps(res, indent=1))
    json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
