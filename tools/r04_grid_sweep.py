"""Grid product (grid.hip) at 384^3 / 500^3 (/ 640^3): time per product against the pair product of the same storage, walk depths and
store policies (VEXHIP_PLANE_DEPTH / VEXHIP_PLANE_STORE), bit-identity.  Writes gpurun_out/r04_grid_sweep.json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops

dev = torch.device("cuda:0")


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {}
grids = [int(a) for a in sys.argv[1:]] or [384, 500]
for g in grids:
    N = g ** 3
    p, c, v = ops.poisson3d(g, dev)
    x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 7)
    y = torch.empty(N, dtype=torch.float64, device=dev); yb = torch.empty_like(y)
    t0 = time.time(); A = ops.SpMat(p, c, v); torch.cuda.synchronize(); setup_s = time.time() - t0
    B = ops.SpMat(p, c, v, march=False)
    M = ops.SpMat(p, c, v, plane=False)
    row = {"storage": A.storage, "dictionary_blocks": A.dictionary_blocks, "grid": A.grid, "plane": A.plane, "march_of_no_plane": M.march, "setup_s_first": round(setup_s, 4)}
    tb = min(timed(lambda: B.apply(x, yb)) for _ in range(2))
    row["pair_ms"] = round(tb, 5)
    if M.march:
        row["march_ms"] = round(min(timed(lambda: M.apply(x, y)) for _ in range(2)), 5)
        row["march_equal"] = bool(torch.equal(y, yb))
    del M
    if A.grid:
        ta = min(timed(lambda: A.apply(x, y)) for _ in range(3))
        row["grid_ms"] = round(ta, 5); row["equal"] = bool(torch.equal(y, yb))
        row["frac_of_8TBps"] = round((A.matrix_bytes() + 16 * N) / ta / 1e6 / 8000.0, 4)
        sweep = {}
        nz = A.grid["planes"]
        for depth in sorted({max(8, nz // 8), max(8, nz // 4), max(8, nz // 3), nz // 2, nz}):
            for store in (0, 1, 2):
                os.environ["VEXHIP_PLANE_DEPTH"] = str(depth); os.environ["VEXHIP_PLANE_STORE"] = str(store)
                C = ops.SpMat(p, c, v)
                t = min(timed(lambda: C.apply(x, y), 10) for _ in range(2))
                sweep["depth %d store %d" % (depth, store)] = [round(t, 5), bool(torch.equal(y, yb))]
                del C
        os.environ.pop("VEXHIP_PLANE_DEPTH"); os.environ.pop("VEXHIP_PLANE_STORE")
        row["sweep"] = sweep
    out[str(g)] = row
    print(g, json.dumps(row), flush=True)
    del A, B, p, c, v, x, y, yb
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r04_grid_sweep.json", "w"), indent=1)
