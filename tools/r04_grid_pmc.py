"""HBM bytes per launch of the grid product (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE through bench.measure_traffic) at the given grid sizes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out = {}
for g in [int(a) for a in sys.argv[1:]] or [500, 640]:
    tr, how = bench.measure_traffic(g, only="auto")
    N = g ** 3
    row = {"how": how, "priced_bytes": 16 * N}
    if tr:
        for k, v in tr.items():
            if isinstance(v, dict):
                row[k] = dict(v, over_priced=round(v["total"] / (16.0 * N), 4), read_over_x=round(v["read"] / (8.0 * N), 4))
        row["fetch_calibration_factor"] = tr.get("fetch_calibration_factor")
    out[str(g)] = row
    print(g, json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r04_grid_pmc.json", "w"), indent=1)
