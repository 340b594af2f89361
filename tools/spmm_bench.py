"""Multi-right-hand-side SpMV (SpMat * multivector) on the 3-D Poisson matrix:
time of ONE fused launch for k vectors against k single-vector products."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
ptr, col, val = ops.poisson3d(n, device=dev)
A = ops.SpMat(ptr, col, val)
N, nnz = n ** 3, col.numel()
print("grid %d^3 rows %d nnz %d format %s codes %s" % (n, N, nnz, A.fmt, A.hell.deltas is not None))
del ptr, col, val
torch.cuda.empty_cache()

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for k in (1, 2, 3, 4, 8):
    xs = [ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), seed=k * 10 + i) for i in range(k)]
    ys = [torch.empty(N, dtype=torch.float64, device=dev) for _ in range(k)]
    t_sep = timed(lambda: [A.apply(x, y) for x, y in zip(xs, ys)])
    t_fused = timed(lambda: A.apply_multi(xs, ys))
    ref = [torch.empty(N, dtype=torch.float64, device=dev) for _ in range(k)]
    for x, y in zip(xs, ref): A.apply(x, y)
    same = all(torch.equal(a, b) for a, b in zip(ys, ref))
    alg = 9 * nnz + k * 16 * N if A.hell.deltas is not None else 12 * nnz + k * 16 * N
    print("k=%d separate %.3f ms fused %.3f ms speed-up %.2fx  %.1f GFLOP/s  stored-bytes rate %.2f TB/s  identical %s"
          % (k, t_sep, t_fused, t_sep / t_fused, 2e-6 * nnz * k / t_fused, alg / t_fused * 1e-9, same))
    del xs, ys, ref
