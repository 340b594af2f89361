"""Driver of tools/r06_sq_spmv.sh: a few launches of the products whose rates the round's review asks about -- the fp32 / fp64 grid products at
384^3, the headline plane product, the flat grid product on 12 000^2, the runs product on the 27-point operator at 320^3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from vexcl_amd import ops
import unstructured as U
dev = torch.device("cuda:0")
def run(p, c, v, reps=6):
    n = p.numel() - 1
    x = ops.fill_hash(torch.empty(n, dtype=torch.float64, device=dev), 42).to(v.dtype); y = torch.empty_like(x)
    A = ops.SpMat(p, c, v)
    for _ in range(reps): A.apply(x, y)
    torch.cuda.synchronize()
    print(n, A.product, flush=True)
p, c, v = ops.poisson3d(384, dev); run(p, c, v); run(p, c, v.float()); del p, c, v
p, c, v = ops.poisson3d(512, dev); run(p, c, v); del p, c, v; torch.cuda.empty_cache()
p, c, v, _ = U.stencil2d(12000, 12000, dev); run(p, c, v); del p, c, v; torch.cuda.empty_cache()
p, c, v = U.stencil27_const(320, dev); run(p, c, v)
