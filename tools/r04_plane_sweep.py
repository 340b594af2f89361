"""Sweep of the plane product at 512^3: tile height x depth x store policy (VEXHIP_PLANE_TILE / _DEPTH / _STORE at plan time),
interleaved in one process with the march product and torch's copy; bit-identity against the march product asserted.
Usage: python tools/r04_plane_sweep.py [TILExDEPTHxSTORE ...]   JSON on stdout."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vexcl_amd import ops
dev = torch.device("cuda:0"); n = 512; N = n ** 3
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42)
y = torch.empty(N, dtype=torch.float64, device=dev)
yref = torch.empty(N, dtype=torch.float64, device=dev)
def timed(fn, reps=30, rounds=3):
    best = 1e30
    for _ in range(rounds):
        for _ in range(30): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
p, c, v = ops.poisson3d(n, dev)
mats = {}
cfgs = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(2, 128, 0), (2, 128, 1), (2, 256, 1), (2, 512, 1), (4, 128, 2), (4, 256, 0), (4, 256, 2), (4, 256, 3)]
for tile, d, st in cfgs:
    os.environ["VEXHIP_PLANE_DEPTH"] = str(d); os.environ["VEXHIP_PLANE_TILE"] = str(tile); os.environ["VEXHIP_PLANE_STORE"] = str(st)
    A = ops.SpMat(p, c, v)
    assert A.plane is not None and (A.plane["tile"], A.plane["depth"], A.plane["store_policy"]) == (tile, d, st), A.plane
    mats["plane tile %d depth %d store %d" % (tile, d, st)] = A
for k in ("VEXHIP_PLANE_DEPTH", "VEXHIP_PLANE_TILE", "VEXHIP_PLANE_STORE"): os.environ.pop(k)
mats["march"] = ops.SpMat(p, c, v, plane=False)
del p, c, v
for A in mats.values(): A.ptr = A.col = A.val = None
torch.cuda.empty_cache()
mats["march"].apply(x, yref)
out = {}
for rnd in range(3):
    for k, A in mats.items():
        if rnd == 0:
            y.zero_(); A.apply(x, y)
            assert torch.equal(y, yref), k
        out.setdefault(k, []).append(round(timed(lambda: A.apply(x, y)), 4))
    out.setdefault("copy", []).append(round(timed(lambda: y.copy_(x)), 4))
print(json.dumps(out))
