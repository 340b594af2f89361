"""Timing-only ablations of the march kernel (results are WRONG for ablate != 0): which part of a slice costs what.
bits: 1 no far gathers, 2 no y store, 4 no ring reads, 8 no barrier, 16 no value look-ups, 32 no window loads."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vexcl_amd import ops
n = 512; dev = torch.device("cuda:0"); N = n ** 3
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42); y = torch.empty(N, dtype=torch.float64, device=dev)
p, c, v = ops.poisson3d(n, dev)
os.environ["VEXHIP_MARCH_PF"] = "0"
A = ops.SpMat(p, c, v); B = ops.SpMat(p, c, v, march=False)
del p, c, v
def timed(fn, reps=30, rounds=3):
    best = 1e30
    for _ in range(rounds):
        for _ in range(40): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
out = {"pair": round(timed(lambda: B.apply(x, y)), 5)}
z = torch.empty_like(y)
out["copy_16B_per_lane_torch"] = round(timed(lambda: z.copy_(x)), 5)
for ab in (0, 1, 2, 4, 8, 16, 32, 1 | 4, 1 | 2, 1 | 2 | 4, 1 | 4 | 16, 1 | 2 | 4 | 16, 1 | 2 | 4 | 8 | 16, 1 | 2 | 4 | 8 | 16 | 32, 0):
    os.environ["VEXHIP_MARCH_ABLATE"] = str(ab)
    out["march_ablate_%d" % ab] = round(timed(lambda: A.apply(x, y)), 5)
print(json.dumps(out))
