import os, sys, time
sys.path.insert(0, ".")
import torch
from vexcl_amd import ops
dev = torch.device("cuda:0")
mode = sys.argv[1]
p, c, v = ops.poisson3d(512, dev)
torch.cuda.synchronize()
if mode == "readfirst":
    s1 = int(c.sum()); s2 = float(v.sum())
elif mode == "small_first":
    pp, cc, vv = ops.poisson3d(256, dev); A0 = ops.SpMat(pp, cc, vv); torch.cuda.synchronize(); del A0
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); A = ops.SpMat(p, c, v); torch.cuda.synchronize()
    print(mode, "setup %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True); del A
