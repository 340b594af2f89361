"""The plane product at 512^3: time of 5 x 40 products, bit-identity with the pair product ('=' and '+= alpha'), walk depths.
Written for the A/B of a variant that requested every line one more step ahead (VEXHIP_PLANE_DEEP=1 selected it; DESIGN.md 3.0c:
bit-identical, 0.423 against 0.397 ms) -- the variant was removed after the measurement, the switch no longer exists."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vexcl_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=40):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
g = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = g ** 3
p, c, v = ops.poisson3d(g, dev)
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 7); y = torch.empty_like(x); yb = torch.empty_like(x)
B = ops.SpMat(p, c, v, march=False); B.apply(x, yb)
out = {"deep": os.environ.get("VEXHIP_PLANE_DEEP", "0")}
for depth in (None, 256, 128):
    if depth: os.environ["VEXHIP_PLANE_DEPTH"] = str(depth)
    A = ops.SpMat(p, c, v)
    t = sorted(timed(lambda: A.apply(x, y)) for _ in range(5))
    ya = torch.full_like(y, 3.0); yc = torch.full_like(y, 3.0)
    A.apply(x, ya, -0.5, True); B.apply(x, yc, -0.5, True)
    out["depth %s" % (A.plane["depth"])] = {"ms_min_med_max": [round(t[0], 5), round(t[2], 5), round(t[4], 5)], "equal": bool(torch.equal(y, yb)), "append_equal": bool(torch.equal(ya, yc))}
    del A
    os.environ.pop("VEXHIP_PLANE_DEPTH", None)
print(json.dumps(out))
