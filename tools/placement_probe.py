"""Is the SpMV time spread a property of the allocation's pages?  Correlates a plain
streaming read (Reductor SUM) of each matrix allocation with the SpMV time on it. (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops, lib
dev = torch.device("cuda:0")
n = 512; N = n ** 3
def t(fn, k=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / k, 4)
ptr, col, val = ops.poisson3d(n, dev)
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42); y = torch.zeros(N, dtype=torch.float64, device=dev)
S0 = ops.SlicedELL(ptr, col, val)
del ptr, col, val
red = ops.Reductor("SUM")
keep = []
for trial in range(8):
    v = S0.sell.clone()
    S = ops.SlicedELL.__new__(ops.SlicedELL); S.__dict__.update(S0.__dict__); S.sell = v
    print(trial, "spmv ms", t(lambda: S.mul(x, y)), "plain", t(lambda: S.mul(x, y, tiled=False)), flush=True)
    keep.append(v)
