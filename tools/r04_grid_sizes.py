"""Grid product at several grid sizes with the plan as the library chooses it: time, pair product, fraction of 8 TB/s, bit-identity."""
import sys, os, torch, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vexcl_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
out = {}
for g in [int(a) for a in sys.argv[1:]] or [384, 500, 640, 700, 256, 168]:
    N = g**3
    p, c, v = ops.poisson3d(g, dev)
    A = ops.SpMat(p, c, v); B = ops.SpMat(p, c, v, march=False)
    x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 7); y = torch.empty_like(x); yb = torch.empty_like(x)
    ta = min(timed(lambda: A.apply(x, y)) for _ in range(3)); tb = min(timed(lambda: B.apply(x, yb)) for _ in range(2))
    out[str(g)] = {"product": "grid" if A.grid else ("plane" if A.plane else "march" if A.march else "pair"), "plan": A.grid or A.plane, "ms": round(ta, 5), "pair_ms": round(tb, 5),
                   "frac_of_8TBps": round((A.matrix_bytes() + 16 * N) / ta / 1e6 / 8000, 4), "equal": bool(torch.equal(y, yb))}
    print(g, json.dumps(out[str(g)]), flush=True)
    del A, B, p, c, v, x, y, yb; torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r04_grid_sizes.json", "w"), indent=1)
