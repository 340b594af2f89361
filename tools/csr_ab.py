import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops
dev = torch.device("cuda:0")
ptr, col, val = ops.poisson3d(512, device=dev)
A = ops.SpMat(ptr, col, val, fmt="csr")
print("trav grid", A.csr_trav.grid_blocks, "chunk", A.csr_trav.chunk, "planes", A.csr_trav.planes, "plane_blocks", A.csr_trav.plane_blocks)
N = 512 ** 3
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 1); y = torch.empty_like(x); y2 = torch.empty_like(x)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rep in range(2):
    t1 = timed(lambda: ops.spmv_csr(ptr, col, val, x, y, traversal=A.csr_trav))
    t0 = timed(lambda: ops.spmv_csr(ptr, col, val, x, y2))
    print("ordered %.3f ms  plain %.3f ms  equal %s" % (t1, t0, torch.equal(y, y2)))
