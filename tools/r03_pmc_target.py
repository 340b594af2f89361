#!/usr/bin/env python
"""Workload of the round-3 counter passes (tools/r03_sq.sh): the value-coded 512^3 Poisson matrix through the pair product
and the march product, plus a calibration stream of known size."""
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
A = ops.SpMat(ptr, col, val)
B = ops.SpMat(ptr, col, val, march=False)
cal = torch.empty(1 << 28, dtype=torch.float64, device=dev).normal_()
r = ops.Reductor("SUM")
torch.cuda.synchronize()
for _ in range(3):
    r.device_result(cal)
for _ in range(4):
    B.apply(x, y)
for _ in range(4):
    A.apply(x, y)
torch.cuda.synchronize()
print("done", A.march)
