import os, sys, time
sys.path.insert(0, ".")
flags = set(sys.argv[1:])
import torch
if "dist" in flags:
    import torch.distributed as dist
    from vexcl_amd.distributed import DistReductor, DistSpMat, partition
from vexcl_amd import lib, ops
if "avail" in flags:
    assert torch.cuda.is_available()
if "count" in flags:
    assert torch.cuda.device_count() >= 1
if "setdev" in flags:
    torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
L = lib()
n = 512; N = n ** 3
if "nnz" in flags:
    nnz_total = L.poisson3d_nnz(n)
ptr, col, val = ops.poisson3d(n, dev, rows=(0, N)) if "rows" in flags else ops.poisson3d(n, dev)
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), (42 + 0 * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
y = torch.zeros(N, dtype=torch.float64, device=dev)
for _ in range(30):
    y.copy_(x)
y.zero_()
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info(dev)[0]
t0 = time.perf_counter()
A = ops.SpMat(ptr, col, val, fmt="auto", dictionary=True, march=True, plane=True, direct=True) if "kw" in flags else ops.SpMat(ptr, col, val)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(sorted(flags), "create %.3f ms, sync %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
