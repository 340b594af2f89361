"""Round 5: does the box-to-box spread of the headline (0.372 - 0.402 ms) come from where x and y lie relative to each other?
The 512^3 product with x and y as separate allocations (what bench.py does) and as views into ONE allocation at several gaps."""
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
out = {}
def t_of(x, y):
    A.apply(x, y)
    return round(min(bench.timed_events(torch, lambda: A.apply(x, y), 30) for _ in range(3)), 5)
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42); y = torch.empty_like(x)
out["separate allocations"] = {"ms": t_of(x, y), "y_minus_x_bytes": y.data_ptr() - x.data_ptr(), "x_mod_2MiB": x.data_ptr() % (2 << 20)}
print("separate", out["separate allocations"], flush=True)
del y
big = torch.empty(2 * N + (64 << 20) // 8, dtype=torch.float64, device=dev)
big[:N].copy_(x); del x
for gap in (0, 512, 4096, 32768, 65536, 262144, 786432, 1048576, 1572864, 4194304):
    xv = big[:N]; yv = big[N + gap:2 * N + gap]
    out["one allocation, gap %d elements" % gap] = {"ms": t_of(xv, yv), "y_minus_x_bytes": yv.data_ptr() - xv.data_ptr()}
    print(gap, out["one allocation, gap %d elements" % gap], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r05_xy_placement.json", "w"), indent=1)
