"""Round 5: the plane product (512-point lines) on grids whose planes are NOT 512 lines -- tiles per CU 1.25 / 1.5 / 0.75 ... --
over the walk depth (VEXHIP_PLANE_DEPTH; None = the plan's choice)."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from vexcl_amd import ops
import bench
from test_gpu_distributed import _stencil_strip
dev = torch.device("cuda:0")
out = {}
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(512, 640, 640), (512, 768, 512), (512, 384, 768), (512, 1024, 256), (512, 320, 1024)]
divs = [int(d) for d in os.environ.get("SWEEP_DIVS", "0,1,2,3,4,6,8,12").split(",")]
for (nx, ny, nz) in shapes:
    N = nx * ny * nz
    p, c, v = _stencil_strip(torch, nx, ny, nz, 0, N, dev)
    x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 7); y = torch.empty_like(x)
    for depth in [None if d == 0 else (nz + d - 1) // d for d in divs]:
        if depth is None: os.environ.pop("VEXHIP_PLANE_DEPTH", None)
        else: os.environ["VEXHIP_PLANE_DEPTH"] = str(depth)
        A = ops.SpMat(p, c, v)
        A.apply(x, y)
        t = min(bench.timed_events(torch, lambda: A.apply(x, y), 10) for _ in range(2))
        key = "%dx%dx%d depth %s" % (nx, ny, nz, depth)
        out[key] = {"ms": round(t, 4), "frac": round((A.matrix_bytes() + 16 * N) / t / 1e6 / 8000, 4), "plane": {k: A.plane[k] for k in ("lines_per_plane", "planes", "depth", "tile")} if A.plane else None}
        print(key, out[key], flush=True)
        del A
    del p, c, v, x, y; torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r05_plane_shapes.json", "w"), indent=1)
