#!/usr/bin/env python
"""Small fixed workload for the PMC passes: a calibration copy of known size
(16-byte coalesced float4-style stream) followed by a few 512^3 products with
both kernels.  Counter values are read per dispatch from rocprofv3's CSV."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops

dev = torch.device("cuda:0")
n = int(os.environ.get("GRID", "512"))
N = n ** 3
ptr, col, val = ops.poisson3d(n, dev)
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42)
y = torch.zeros(N, dtype=torch.float64, device=dev)
A_csr = ops.SpMat(ptr, col, val, fmt="csr")
A_ell = ops.SpMat(ptr, col, val, fmt="hell")
A_sell = ops.SpMat(ptr, col, val, fmt="sell")            # SELL8V on this matrix (diagonal + value codes)
A_sell8 = ops.SpMat(ptr, col, val, fmt="sell8")          # diagonal codes, values as they are
A_sell32 = ops.SpMat(ptr, col, val, fmt="sell32")        # 32-bit columns
# calibration stream: 2 GiB read by the reduction kernel (16-byte loads), known byte count
cal = torch.empty(1 << 28, dtype=torch.float64, device=dev).normal_()
r = ops.Reductor("SUM")
torch.cuda.synchronize()
for _ in range(3):
    r.device_result(cal)
for _ in range(3):
    A_ell.apply(x, y)
for _ in range(3):
    A_csr.apply(x, y)
for _ in range(3):
    A_sell.apply(x, y)
for _ in range(3):
    A_sell8.apply(x, y)
for _ in range(3):
    A_sell32.apply(x, y)
torch.cuda.synchronize()
print("done")
