"""Round 5: set-up time of the default storage for the unstructured bench matrices (CSR arrays resident)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from vexcl_amd import ops
import unstructured as U
dev = torch.device("cuda:0")
m = int(float(os.environ.get("UNSTRUCTURED_ROWS", "2e7")))
out = {}
for name in ("random16", "powerlaw"):
    p, c, v = U.MAKERS[name](m, dev)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        A = ops.SpMat(p, c, v)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        st = A.storage; del A
    nbytes = c.numel() * 12 + p.numel() * p.element_size()
    out[name] = {"setup_ms": [round(t, 2) for t in ts], "storage": st, "csr_bytes": nbytes, "nnz": int(c.numel())}
    print(name, out[name], flush=True)
    del p, c, v; torch.cuda.empty_cache()
json.dump(out, open("gpurun_out/r05_setup_unstructured.json", "w"), indent=1)
