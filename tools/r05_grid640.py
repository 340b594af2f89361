"""Round 5: the grid product at 640^3 / 700^3 over segment length and walk depth (VEXHIP_GRID_SEGMENT, VEXHIP_PLANE_DEPTH)."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vexcl_amd import ops
import bench
dev = torch.device("cuda:0")
out = {}
for g in [int(a) for a in sys.argv[1:]] or [640, 700]:
    N = g ** 3
    p, c, v = ops.poisson3d(g, dev)
    x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 7); y = torch.empty_like(x)
    segs = os.environ.get("SWEEP_SEGS", "512,1024").split(",")
    divs = [int(d) for d in os.environ.get("SWEEP_DIVS", "0,1,2,3,4,6,8").split(",")]
    for seg in segs:
        for depth in [None if d == 0 else (g + d - 1) // d for d in divs]:
            os.environ["VEXHIP_GRID_SEGMENT"] = seg
            if depth is None: os.environ.pop("VEXHIP_PLANE_DEPTH", None)
            else: os.environ["VEXHIP_PLANE_DEPTH"] = str(depth)
            A = ops.SpMat(p, c, v)
            A.apply(x, y)
            t = min(bench.timed_events(torch, lambda: A.apply(x, y), 10) for _ in range(2))
            key = "%d seg %s depth %s" % (g, seg, depth)
            out[key] = {"ms": round(t, 4), "frac": round((A.matrix_bytes() + 16 * N) / t / 1e6 / 8000, 4), "plan": {k: A.grid[k] for k in ("segments", "segment_rows", "threads", "depth")} if A.grid else None}
            print(key, out[key], flush=True)
            del A
    del p, c, v, x, y; torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r05_grid640.json", "w"), indent=1)
