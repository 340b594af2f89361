"""4 right-hand sides in one pass (SpMat * multivector<double,4>) at 512^3: variable-coefficient matrix (diagonal codes + fp64
values) and Poisson (value codes); per-RHS results compared with the single-vector product.  Output: gpurun_out/r02_spmm.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops
dev = torch.device("cuda:0"); n = 512; N = n ** 3
out = {}
for name, gen in (("variable", ops.diffusion3d), ("poisson", ops.poisson3d)):
    ptr, col, val = gen(n, device=dev)
    nnz = col.numel()
    A = ops.SpMat(ptr, col, val); del ptr, col, val
    xs = [ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 10 + k) for k in range(4)]
    ys = [torch.empty(N, dtype=torch.float64, device=dev) for _ in range(4)]
    yref = torch.empty(N, dtype=torch.float64, device=dev)
    A.apply_multi(xs, ys)
    same = True
    for k in range(4):
        A.apply(xs[k], yref); same &= bool(torch.equal(ys[k], yref))
    for _ in range(12): A.apply_multi(xs, ys)           # ~40 ms of the same launches ahead of the timed ones
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): A.apply_multi(xs, ys)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    for _ in range(40): A.apply(xs[0], yref)
    e0.record()
    for _ in range(40): A.apply(xs[0], yref)
    e1.record(); torch.cuda.synchronize()
    ms1 = e0.elapsed_time(e1) / 40
    moved = A.matrix_bytes() + 4 * 16 * N
    out[name] = {"storage": A.storage, "ms_4rhs": round(ms, 4), "gflops": round(8.0 * nnz / ms / 1e6, 1), "ms_single": round(ms1, 4),
                 "bytes_moved": moved, "moved_tbps": round(moved / ms / 1e9, 3), "identical_to_single_products": same}
    print(name, out[name], flush=True)
    del A, xs, ys, yref; torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r02_spmm.json", "w"), indent=1)
