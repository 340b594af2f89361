"""Set-up of the headline matrix only (for rocprofv3 --kernel-trace --stats): CSR arrays in HBM -> vexhip_spmat."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
p, c, v = ops.poisson3d(n, dev)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    A = ops.SpMat(p, c, v)
    torch.cuda.synchronize()
    print("setup %.3f ms" % ((time.perf_counter() - t0) * 1e3), A.storage, A.dictionary_blocks, A.march, flush=True)
    del A
