"""CSR product with 64-bit row pointers whose VALUES lie beyond 2^31 while the matrix is small (debug)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops
dev = torch.device("cuda:0")
n = 100
N = n ** 3
p, c, v = ops.poisson3d(n, dev)
nnz = c.numel()
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 11)
yref = torch.empty(N, dtype=torch.float64, device=dev)
ops.SpMat(p, c, v, fmt="csr").apply(x, yref)
for off in [int(a) for a in sys.argv[1:]] or [0, 2 ** 31 - 4096, 2 ** 31 + 4096, 2200000000]:
    cb = torch.empty(off + nnz, dtype=torch.int32, device=dev)
    vb = torch.empty(off + nnz, dtype=torch.float64, device=dev)
    cb[off:] = c; vb[off:] = v
    p64 = p.to(torch.int64) + off
    A = ops.SpMat(p64, cb, vb, fmt="csr")
    y = torch.zeros(N, dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); t0 = time.time()
    A.apply(x, y)
    torch.cuda.synchronize(); t1 = time.time()
    print("offset", off, "apply %.4f s" % (t1 - t0), "equal", bool(torch.equal(y, yref)), flush=True)
    del A, cb, vb, y
    torch.cuda.empty_cache()
