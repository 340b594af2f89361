"""Round 2 A/B of the CSR kernel at 512^3 (Poisson and variable coefficients): first form (x gathered in stream order, products staged in LDS;
variant word bit 2 set) against the second form ((col, val) staged in LDS, every lane gathers along its own row).
Interleaved, bit-identity checked.  Output: gpurun_out/r02_csr_ab.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops, lib
L = lib(); dev = torch.device("cuda:0")
n = 512; N = n ** 3
out = {}
for name, gen in (("poisson", ops.poisson3d), ("variable", ops.diffusion3d)):
    ptr, col, val = gen(n, device=dev)
    nnz = col.numel()
    alg = nnz * 12 + (N + 1) * 4 + 16 * N
    x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 1)
    tr = ops.csr_traversal(ptr, col)
    yref = torch.empty_like(x); y = torch.empty_like(x)
    L.spmv_csr_set_variant(4); ops.spmv_csr(ptr, col, val, x, yref, traversal=tr); torch.cuda.synchronize()
    res = {}
    for rnd in range(3):
        for v, label in ((4, "first form"), (0, "second form"), (0, "second form, row order")):
            L.spmv_csr_set_variant(v)
            t = None if "row order" in label else tr
            ops.spmv_csr(ptr, col, val, x, y, traversal=t); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.spmv_csr(ptr, col, val, x, y, traversal=t)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            r = res.setdefault(label, {"ms": [], "identical": True})
            r["ms"].append(round(ms, 4)); r["identical"] &= bool(torch.equal(y, yref))
    for label, r in res.items():
        m = min(r["ms"]); r["best_ms"] = m; r["alg_tbps"] = round(alg / m / 1e9, 3); r["frac_of_8TBps"] = round(alg / m / 1e9 / 8, 4)
        print("%-9s %-34s %s best %.4f ms  %.2f TB/s algorithmic = %.3f of 8 TB/s  identical %s" % (name, label, r["ms"], m, r["alg_tbps"], r["frac_of_8TBps"], r["identical"]), flush=True)
    out[name] = res
    del ptr, col, val
    torch.cuda.empty_cache()
L.spmv_csr_set_variant(-1)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r02_csr_ab.json", "w"), indent=1)
