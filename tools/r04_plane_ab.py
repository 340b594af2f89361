"""A/B of the plane product (round 4) against the march and pair products, 512^3 Poisson, one process, interleaved; bit-identity
asserted.  Also small banded cases through the forced plan (VEXHIP_PLANE_FORCE=1) against the pair product.
Usage: python tools/r04_plane_ab.py  -> JSON on stdout (profiles/r04_plane_ab.json)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vexcl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
out = {"small": [], "ms": {}}


def band(n, offsets, seed, constant=True):
    rows = np.arange(n, dtype=np.int64)[:, None]
    cols = rows + np.array(offsets, dtype=np.int64)[None, :]
    ok = (cols >= 0) & (cols < n)
    vals = np.stack([np.full(n, 0.5 + k) for k in range(len(offsets))], axis=1)
    ptr = np.zeros(n + 1, dtype=np.int32)
    ptr[1:] = np.cumsum(ok.sum(axis=1))
    return ptr, cols[ok].astype(np.int32), vals[ok]


up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
os.environ["VEXHIP_PLANE_FORCE"] = "1"
for ny, nz, extra in ((8, 12, 0), (4, 40, 0), (16, 9, 3 * 512), (6, 33, 0)):
    P = 512 * ny
    m = P * nz + extra
    ptr, col, val = band(m, (-P, -512, -1, 0, 1, 512, P), 5)
    A = ops.SpMat(up(ptr), up(col), up(val))
    B = ops.SpMat(up(ptr), up(col), up(val), march=False)
    x = torch.rand(m, dtype=torch.float64, device=dev)
    y0 = torch.rand(m, dtype=torch.float64, device=dev)
    res = {"ny": ny, "nz": nz, "rows": m, "plane": A.plane, "dict": A.dictionary_blocks}
    if A.plane is not None:
        for alpha, append in ((1.0, False), (-0.75, True)):
            ya, yb = y0.clone(), y0.clone()
            A.apply(x, ya, alpha, append); B.apply(x, yb, alpha, append)
            res["same a=%g" % alpha] = bool(torch.equal(ya, yb))
            if not torch.equal(ya, yb):
                bad = (ya != yb).nonzero().flatten()
                res["first_bad"] = [int(v) for v in bad[:8]]
                res["nbad"] = int(bad.numel())
    out["small"].append(res)
    print(json.dumps(res), file=sys.stderr, flush=True)
os.environ.pop("VEXHIP_PLANE_FORCE")

n = 512
N = n ** 3
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42)
y = torch.empty(N, dtype=torch.float64, device=dev)
yref = torch.empty_like(y)


def timed(fn, reps=30, rounds=3):
    best = 1e30
    for _ in range(rounds):
        for _ in range(40):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


p, c, v = ops.poisson3d(n, dev)
mats = {"plane": ops.SpMat(p, c, v), "march": ops.SpMat(p, c, v, plane=False), "pair": ops.SpMat(p, c, v, march=False)}
out["plane_plan"] = mats["plane"].plane
assert mats["plane"].plane is not None and mats["march"].plane is None and mats["march"].march is not None
del p, c, v
for A in mats.values():
    A.ptr = A.col = A.val = None
torch.cuda.empty_cache()
mats["pair"].apply(x, yref)
for rnd in range(3):
    for name, A in mats.items():
        y.zero_()
        A.apply(x, y)
        assert torch.equal(y, yref), name
        out["ms"].setdefault(name, []).append(round(timed(lambda: A.apply(x, y)), 4))
y2 = torch.rand(N, dtype=torch.float64, device=dev)
ya, yb = y2.clone(), y2.clone()
mats["plane"].apply(x, ya, -0.5, True); mats["pair"].apply(x, yb, -0.5, True)
out["append_same"] = bool(torch.equal(ya, yb))
out["copy_ms"] = round(timed(lambda: y.copy_(x)), 4)
print(json.dumps(out))
