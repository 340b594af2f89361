"""First set-up in a process under different preludes (what makes the first creation slow?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
n = 512
p, c, v = ops.poisson3d(n, dev)
torch.cuda.synchronize()
if mode in ("xy", "copies", "meminfo", "all"):
    x = ops.fill_hash(torch.empty(n ** 3, dtype=torch.float64, device=dev), 42); y = torch.zeros(n ** 3, dtype=torch.float64, device=dev)
if mode in ("copies", "all"):
    for _ in range(30):
        y.copy_(x)
    y.zero_()
torch.cuda.synchronize()
if mode in ("meminfo", "all"):
    free0 = torch.cuda.mem_get_info(dev)[0]
for rep in range(2):
    t0 = time.perf_counter(); A = ops.SpMat(p, c, v); torch.cuda.synchronize()
    print(mode, "setup %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True); del A
