"""Round 2 A/B of the CSR kernel at 512^3 (Poisson and variable coefficients): first form (x gathered in stream order, products staged in LDS;
variant word bit 2 set) against the second form ((col, val) staged in LDS with all loads of a tile in flight, every lane gathers along
its own row).  Interleaved, bit-identity checked.  The variants tried on the way (tile sizes, G row blocks per workgroup with the next
tile prefetched) are recorded in profiles/r02_csr_ab_variants.log and in the comment above csr_stream2_kernel.  Output: gpurun_out/r02_csr_ab.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops, lib
L = lib(); dev = torch.device("cuda:0")
n = 512; N = n ** 3
out = {}
# the second form on irregular matrices first (empty rows, rows longer than a tile, ragged last block), against the first form: empty rows, rows longer than a tile, ragged last block
g = torch.Generator(device="cpu"); g.manual_seed(7)
for rows, longrow in ((1, 0), (255, 0), (1000, 5000), (70001, 2500), (300000, 0)):
    cnt = torch.randint(0, 12, (rows,), generator=g); cnt[torch.rand(rows, generator=g) < 0.2] = 0
    if longrow: cnt[rows // 2] = longrow
    p_ = torch.zeros(rows + 1, dtype=torch.int64); p_[1:] = torch.cumsum(cnt, 0)
    nz = int(p_[-1]); m = rows + 13
    c_ = torch.randint(0, m, (nz,), generator=g).to(torch.int32).to(dev); v_ = torch.randn(nz, generator=g, dtype=torch.float64).to(dev)
    p_ = p_.to(torch.int32).to(dev); x_ = torch.randn(m, generator=g, dtype=torch.float64).to(dev)
    for app in (False, True):
        ya = torch.full((rows,), 3.0, dtype=torch.float64, device=dev)
        L.spmv_csr_set_variant(4); ops.spmv_csr(p_, c_, v_, x_, ya, alpha=0.5, append=app)
        yb = torch.full((rows,), 3.0, dtype=torch.float64, device=dev)
        L.spmv_csr_set_variant(0); ops.spmv_csr(p_, c_, v_, x_, yb, alpha=0.5, append=app)
        ok = torch.equal(ya, yb)
        print("irregular rows=%d long=%d append=%s identical=%s" % (rows, longrow, app, ok), flush=True)
        assert ok
L.spmv_csr_set_variant(-1)
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
