"""A/B of the two forms of the march product and the pair product, 512^3 Poisson, one process, interleaved; bit-identity
against the pair product asserted.  The library under test: VEXHIP_LIBRARY (default: the built one).
Usage: python tools/r03_march2_ab.py [grid=512]  -> JSON on stdout (profiles/r03_march2_ab*.json)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vexcl_amd import ops  # noqa: E402
from vexcl_amd._capi import lib  # noqa: E402

L = lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
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
A = ops.SpMat(p, c, v)
assert A.march is not None
del p, c, v
A.ptr = A.col = A.val = None
torch.cuda.empty_cache()
out = {"grid": n, "library": os.environ.get("VEXHIP_LIBRARY", "default"), "march": A.march, "ms": {}}
L.spmv_sell8_set_variant(2)
A.apply(x, yref)
names = {2: "pair", 0: "march"}
for rnd in range(3):
    for var in (2, 0):
        L.spmv_sell8_set_variant(var)
        y.zero_()
        A.apply(x, y)
        assert torch.equal(y, yref), names[var]
        # y += 0.5 A x against the same through the pair kernel
        out["ms"].setdefault(names[var], []).append(round(timed(lambda: A.apply(x, y)), 4))
L.spmv_sell8_set_variant(0)
print(json.dumps(out))
