"""Round 5: the set-up of the default vexhip_spmat on the benchmark's matrices (CSR arrays resident): wall time from the call to
the synchronised return, best of several (the first builds warm the code objects)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops
dev = torch.device("cuda:0")
n = int(os.environ.get("GRID", "512"))
out = {}
for name, make, dt in (("poisson", ops.poisson3d, torch.float64), ("poisson fp32", ops.poisson3d, torch.float32), ("diffusion", ops.diffusion3d, torch.float64)):
    ptr, col, val = make(n, dev)
    val = val.to(dt)
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        A = ops.SpMat(ptr, col, val)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        direct = A.direct; del A
    csr_bytes = col.numel() * (4 + val.element_size()) + ptr.numel() * ptr.element_size()
    out[name] = {"setup_ms_best": round(min(ts), 3), "setup_ms_all": [round(t, 3) for t in ts], "direct": bool(direct), "csr_bytes": csr_bytes,
                 "csr_read_tbps_at_best": round(csr_bytes / min(ts) / 1e9, 3)}
    print(name, out[name], flush=True)
    del ptr, col, val
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r05_setup%s.json" % os.environ.get("TAG", ""), "w"), indent=1)
