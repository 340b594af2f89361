#!/usr/bin/env python
"""Workload of the round-4 counter passes (tools/r04_sq.sh): the value-coded 512^3 Poisson matrix through the plane product
(round 4), the march product (round 3) and a calibration stream of known size (2 GiB read by the Reductor)."""
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
B = ops.SpMat(ptr, col, val, plane=False)
assert A.plane is not None and B.plane is None and B.march is not None, (A.plane, B.march)
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
print("done", A.plane)
