"""Round 6: WHERE must y lie relative to x for the headline product to run at its fast end (round 5 found 0.379 - 0.399 ms over eleven
gaps on one box, no rule)?  A fine sweep: x and y as views into ONE allocation, y - x = N * 8 + gap bytes, gap from 0 to 64 MiB in
steps of 256 KiB, then a few fine steps (4 KiB) around the best and the worst; every point = best of 3 x 40 products; the whole sweep
twice (is a gap's time a property of the gap?).  Output: gpurun_out/r06_xy_gap.json."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops
import bench
dev = torch.device("cuda:0")
n = 512; N = n ** 3
ptr, col, val = ops.poisson3d(n, dev)
A = ops.SpMat(ptr, col, val)
del ptr, col, val; A.ptr = A.col = A.val = None
torch.cuda.empty_cache()
SL = 128 << 20
big = torch.empty(2 * N + SL // 8, dtype=torch.float64, device=dev)
ops.fill_hash(big[:N], 42)
def t_of(gap_bytes, reps=40):
    xv = big[:N]; yv = big[N + gap_bytes // 8:2 * N + gap_bytes // 8]
    A.apply(xv, yv)
    return round(min(bench.timed_events(torch, lambda: A.apply(xv, yv), reps) for _ in range(3)), 5)
for _ in range(100): t_of(0, 10)          # warm
out = {"base_mod_2MiB": big.data_ptr() % (2 << 20), "base_mod_1GiB": big.data_ptr() % (1 << 30), "sweeps": []}
step = 256 << 10
for rep in range(2):
    sw = {}
    for g in range(0, (64 << 20) + 1, step):
        sw[g] = t_of(g)
    out["sweeps"].append(sw)
    ts = sorted(sw.values())
    print("sweep", rep, "min", ts[0], "median", ts[len(ts) // 2], "max", ts[-1], flush=True)
s0, s1 = out["sweeps"]
both = {g: max(s0[g], s1[g]) for g in s0}
best = min(both, key=both.get); worst = max(both, key=lambda g: min(s0[g], s1[g]))
out["best_gap"] = {"gap": best, "ms": [s0[best], s1[best]]}; out["worst_gap"] = {"gap": worst, "ms": [s0[worst], s1[worst]]}
out["correlation_of_the_two_sweeps"] = float(torch.corrcoef(torch.tensor([[s0[g] for g in s0], [s1[g] for g in s0]]))[0, 1])
fine = {}
for centre in (best, worst):
    for d in range(-16, 17):
        g = centre + d * 4096
        if 0 <= g <= SL - 4096: fine[g] = t_of(g)
out["fine_4KiB_steps_around_best_and_worst"] = fine
print(json.dumps({k: out[k] for k in ("best_gap", "worst_gap", "correlation_of_the_two_sweeps")}))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r06_xy_gap.json", "w"), indent=1)
