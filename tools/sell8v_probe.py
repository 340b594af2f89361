"""SELL8 vs SELL8V (value codes) on the 512^3 Poisson matrix: time and bit-identity."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ptr, col, val = ops.poisson3d(n, device=dev)
N = n ** 3
A8 = ops.SlicedELL(ptr, col, val, value_codes=False)
AV = ops.SlicedELL(ptr, col, val)
print("deltas", AV.ndeltas, "values", AV.nvalues, "sell8 bytes %.2f GB sell8v bytes %.2f GB" % (A8.sell.numel() / 1e9, AV.sell.numel() / 1e9))
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 1); y8 = torch.empty_like(x); yv = torch.empty_like(x)
def timed(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rep in range(2):
    t8 = timed(lambda: A8.mul(x, y8)); tv = timed(lambda: AV.mul(x, yv))
    print("sell8 %.3f ms  sell8v %.3f ms  (%.0f GFLOP/s)  identical %s" % (t8, tv, 2e-6 * col.numel() / tv, torch.equal(y8, yv)))
