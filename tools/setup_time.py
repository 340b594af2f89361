import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from vexcl_amd import ops
dev = torch.device("cuda:0")
# general matrix: 20M rows x 16 random columns, random values -> no codes at all
n, w = 20_000_000, 16
g = torch.Generator(device=dev); g.manual_seed(1)
col = torch.randint(0, n, (n * w,), device=dev, dtype=torch.int32, generator=g)
col, _ = torch.sort(col.view(n, w), dim=1); col = col.reshape(-1).contiguous()
ptr = torch.arange(0, n * w + 1, w, device=dev, dtype=torch.int32)
val = torch.rand(n * w, device=dev, dtype=torch.float64, generator=g)
torch.cuda.synchronize(); t0 = time.perf_counter()
S = ops.SlicedELL(ptr, col, val)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("general matrix %d x %d nnz: set-up %.3f s, ndeltas %d nvalues %d" % (n, w, t1 - t0, S.ndeltas, S.nvalues))
# banded with random values: diagonal codes only
offs = torch.tensor([-5000, -70, -1, 0, 1, 70, 5000], device=dev)
rows = torch.arange(n, device=dev).view(-1, 1)
c2 = (rows + offs).clamp(0, n - 1).to(torch.int32).reshape(-1).contiguous()
ptr2 = torch.arange(0, n * 7 + 1, 7, device=dev, dtype=torch.int32)
v2 = torch.rand(n * 7, device=dev, dtype=torch.float64, generator=g)
torch.cuda.synchronize(); t0 = time.perf_counter()
S2 = ops.SlicedELL(ptr2, c2, v2)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("banded, random values: set-up %.3f s, ndeltas %d nvalues %d" % (t1 - t0, S2.ndeltas, S2.nvalues))
