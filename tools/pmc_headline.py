#!/usr/bin/env python
"""Workload of bench.py's counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): a calibration stream of known
size (2 GiB read with 16-byte loads by the reduction kernel), then three products with each storage of the 512^3
matrices bench.py times -- the default SpMat on the Poisson matrix (value codes), on the variable-coefficient matrix
(diagonal codes + fp64 values), 32-bit columns, and the CSR arrays; PMC_ONLY=fp32: the Poisson matrix in float.  Counter values are read per dispatch from
rocprofv3's CSV by bench.py (measure_traffic)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops

dev = torch.device("cuda:0")
n = int(os.environ.get("GRID", "512"))
N = n ** 3
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42)
y = torch.zeros(N, dtype=torch.float64, device=dev)
cal = torch.empty(1 << 28, dtype=torch.float64, device=dev).normal_()
r = ops.Reductor("SUM")
torch.cuda.synchronize()
for _ in range(3):
    r.device_result(cal)
del cal
only = [f for f in os.environ.get("PMC_ONLY", "").split(",") if f]
unstructured = [f for f in only if f in ("random16", "powerlaw", "banded16", "stencil27")]
if unstructured:
    # round 5: the unstructured rows of bench.py (tools/unstructured.py); UNSTRUCTURED_ROWS rows
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import unstructured as U
    del x, y
    m = int(float(os.environ.get("UNSTRUCTURED_ROWS", "2e7")))
    xu = ops.fill_hash(torch.empty(m, dtype=torch.float64, device=dev), 42)
    yu = torch.zeros(m, dtype=torch.float64, device=dev)
    for name in unstructured:
        mm = U.ROWS_OF.get(name, m)
        if mm != m:
            xu = ops.fill_hash(torch.empty(mm, dtype=torch.float64, device=dev), 42); yu = torch.zeros(mm, dtype=torch.float64, device=dev)
        p_, c_, v_ = U.MAKERS[name](mm, dev)
        A = ops.SpMat(p_, c_, v_)
        torch.cuda.synchronize()
        for _ in range(3):
            A.apply(xu, yu)
        torch.cuda.synchronize()
        del A, p_, c_, v_
        torch.cuda.empty_cache()
    print("done")
    sys.exit(0)
if "fp32" in only:
    # round 5: the headline operator in float (bench.py's fp32 row): three products with the default storage
    del x, y
    ptr, col, val = ops.poisson3d(n, dev)
    v32 = val.to(torch.float32)
    del val
    xf = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42).to(torch.float32)
    yf = torch.zeros(N, dtype=torch.float32, device=dev)
    A = ops.SpMat(ptr, col, v32)
    torch.cuda.synchronize()
    for _ in range(3):
        A.apply(xf, yf)
    torch.cuda.synchronize()
    print("done")
    sys.exit(0)
ptr, col, val = ops.poisson3d(n, dev)
for fmt in ("auto", "sell32", "csr"):
    if only and fmt not in only:
        continue
    A = ops.SpMat(ptr, col, val, fmt=fmt)
    for _ in range(3):
        A.apply(x, y)
    torch.cuda.synchronize()
    del A
del ptr, col, val
torch.cuda.empty_cache()
if not only or "diffusion" in only:
    ptr, col, val = ops.diffusion3d(n, dev)
    A = ops.SpMat(ptr, col, val)
    for _ in range(3):
        A.apply(x, y)
    torch.cuda.synchronize()
print("done")
