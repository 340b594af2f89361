"""A/B of the march products against the pair products they replace, 512^3, same process, interleaved:
value-coded Poisson (headline) and the variable-coefficient operator (stored values).  Bit-identity is asserted.
Usage: python tools/r03_march_ab.py [grid=512] [runs=8,16,32,64]  -> JSON on stdout
(profiles/r03_march_ab_v*.json: one file per version of the kernel, see the comment above sell8_march_kernel)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vexcl_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
runs = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "8,16,32,64").split(",")]

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


out = {"grid": n, "rows": N}
for label, gen in (("poisson_value_codes", ops.poisson3d),):
    p, c, v = gen(n, dev)
    mats = {"pair": ops.SpMat(p, c, v, march=False)}
    for r in runs:
        os.environ["VEXHIP_MARCH_RUN"] = str(r)
        A = ops.SpMat(p, c, v)
        if A.march is None:
            continue
        mats["march_run%d" % A.march["run"]] = A
    os.environ.pop("VEXHIP_MARCH_RUN", None)
    del p, c, v
    for A in mats.values():
        A.ptr = A.col = A.val = None
    torch.cuda.empty_cache()
    mats["pair"].apply(x, yref)
    res = {}
    for waves in ("4",):
      for rnd in range(2):
        for k, A in mats.items():
            A.apply(x, y)
            assert torch.equal(y, yref), k
            ms = timed(lambda: A.apply(x, y))
            res[k] = min(res.get(k, 1e30), ms)
    out[label] = {k: round(v, 5) for k, v in res.items()}
    out[label + "_plan"] = {k: A.march for k, A in mats.items()}
    del mats
    torch.cuda.empty_cache()
print(json.dumps(out))
