"""Round 2: per-launch duration of the headline product after idle gaps and after other kernels (clock / power transient).\nDiagnostic behind the order of operations in bench.py."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from vexcl_amd import ops
dev = torch.device("cuda:0")
n = 512; N = n**3
ptr, col, val = ops.poisson3d(n, device=dev)
A = ops.SpMat(ptr, col, val)
del ptr, col, val
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 1); y = torch.empty_like(x)
torch.cuda.synchronize()
def series(k, label):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
    ev[0].record()
    for i in range(k):
        A.apply(x, y); ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(k)]
    print(label, " ".join("%.3f" % m for m in ms), flush=True)
series(40, "cold start      ")
series(40, "right after     ")
for gap in (0.0005, 0.002, 0.005, 0.02, 0.05):
    for _ in range(200): A.apply(x, y)
    torch.cuda.synchronize(); time.sleep(gap)
    series(30, "200 products, %4.1f ms idle" % (gap * 1e3))
# a burst of other (streaming) kernels, then the product at once
for _ in range(200): A.apply(x, y)
z = torch.empty_like(x)
for _ in range(20): torch.add(x, y, out=z)
series(30, "200 products, 20 torch adds ")
for _ in range(20): torch.add(x, y, out=z)
torch.cuda.synchronize()
series(30, "20 torch adds, sync         ")
m = float((z - y).abs().max())
series(30, "reduction with host readback")

# what bench.py does before its warm-up: the independent stencil evaluation, queued without a host synchronisation
sys.path.insert(0, "/root/repo")
import bench
for rep in range(2):
    time.sleep(0.05)
    A.apply(x, y)
    yref, bound = bench.independent_product(torch, x, n, lazy=True)
    chk = ((y - yref).abs().max(), yref.sum(dtype=torch.float64), bound)
    del yref
    series(30, "bench check queued, then    ")
    print("   check:", [float(v) for v in chk])
# long stream of plain adds (continuous HBM load ~40 ms), then the product
for _ in range(70): torch.add(x, y, out=z)
series(30, "70 torch adds (40 ms)       ")
