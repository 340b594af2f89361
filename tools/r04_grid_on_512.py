import sys, os, torch
sys.path.insert(0, ".")
from vexcl_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
g = 512; N = g**3
p, c, v = ops.poisson3d(g, dev)
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 7); y = torch.empty_like(x); yb = torch.empty_like(x)
A = ops.SpMat(p, c, v)
print("plane", A.plane, min(timed(lambda: A.apply(x, yb), 40) for _ in range(5)))
for depth in (None, 512, 256, 128):
    os.environ["VEXHIP_NO_PLANE512"] = "1"
    if depth: os.environ["VEXHIP_PLANE_DEPTH"] = str(depth)
    G = ops.SpMat(p, c, v)
    print("grid", G.grid and (G.grid["depth"], G.grid["threads"]), G.plane, min(timed(lambda: G.apply(x, y), 40) for _ in range(5)), bool(torch.equal(y, yb)))
    del G
    os.environ.pop("VEXHIP_NO_PLANE512"); os.environ.pop("VEXHIP_PLANE_DEPTH", None)
