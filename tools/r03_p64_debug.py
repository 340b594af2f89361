"""CSR product with 64-bit row pointers at 700^3 (2.38e9 entries): timing + agreement with the value-coded storage (debug)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 700
N = n ** 3
dp, dc, dv = ops.poisson3d(n, dev, ptr64=True)
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 11)
ys = {}
for fmt in ("auto", "csr"):
    A = ops.SpMat(dp, dc, dv, fmt=fmt)
    y = torch.empty(N, dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); t0 = time.time()
    A.apply(x, y)
    torch.cuda.synchronize(); t1 = time.time()
    print(n, fmt, A.storage, "apply %.4f s" % (t1 - t0), flush=True)
    ys[fmt] = y
print("equal", bool(torch.equal(ys["auto"], ys["csr"])))
