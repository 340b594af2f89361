"""Round 2 A/B: one 8-byte gather per entry (variant 1, round 1) against the pair kernels (variant 0: one 16-byte gather per
lane and ELL column) for the three SELL storages of the 512^3 Poisson matrix.  Interleaved samples, bit-identity checked.
Writes gpurun_out/r02_spmv_ab.json.  (Diagnostic; not part of the product path.)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops, lib

L = lib(); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = n ** 3
ptr, col, val = ops.poisson3d(n, device=dev)
nnz = col.numel()
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 1)
out = {"grid": n, "rows": N, "nnz": nnz, "results": {}}


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, kw, bytes_entry in (("sell8v", dict(), 2), ("sell8", dict(value_codes=False), 9), ("sell32", dict(codes=False), 12)):
    S = ops.SlicedELL(ptr, col, val, **kw)
    moved = S.sell.numel() + 16 * N            # storage + x once + y once
    yref = torch.empty_like(x); y = torch.empty_like(x)
    L.spmv_sell8_set_variant(1)
    S.mul(x, yref); torch.cuda.synchronize()
    # variants (vexhip_spmv_sell8_set_variant): 1 = one 8-byte gather per entry (round 1), 0 = pair kernels (16-byte gathers)
    configs = [(1, 0), (0, 0)]
    best = {}
    for rnd in range(3):
        for v, b in configs:
            L.spmv_sell8_set_variant(v)
            ms = timed(lambda: S.mul(x, y), 30)
            same = bool(torch.equal(y, yref))
            k = "v%d_b%d" % (v, b)
            best.setdefault(k, {"ms": [], "identical": True})
            best[k]["ms"].append(round(ms, 4)); best[k]["identical"] &= same
    for k, r in best.items():
        m = min(r["ms"])
        r["best_ms"] = m; r["gflops"] = round(2.0 * nnz / m / 1e6, 1); r["moved_tbps"] = round(moved / m / 1e9, 3)
        print("%-7s %-8s %s  best %.4f ms  %.0f GFLOP/s  %.2f TB/s of stored bytes  identical %s" % (name, k, r["ms"], m, r["gflops"], r["moved_tbps"], r["identical"]), flush=True)
    # += and alpha through the looping kernel
    L.spmv_sell8_set_variant(0)
    ya = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 9); yb = ya.clone()
    S.mul(x, ya, -0.75, True)
    L.spmv_sell8_set_variant(1)
    S.mul(x, yb, -0.75, True)
    out["results"][name] = {"storage_bytes": S.sell.numel(), "moved_bytes": moved, "append_identical": bool(torch.equal(ya, yb)), "configs": best}
    print(name, "append identical", out["results"][name]["append_identical"], flush=True)
    del S
    torch.cuda.empty_cache()
L.spmv_sell8_set_variant(0)

# compressed-stencil product (vex::SpMatCCSR): 2 unique rows; pair form (0) against the first form with 2 rows per lane
import ctypes
r_in = (n // 2) * (n * n + n + 1)
b0, e0 = int(ptr[r_in]), int(ptr[r_in + 1])
offs = torch.cat([col[int(ptr[0]):int(ptr[1])].long() - 0, col[b0:e0].long() - r_in]).to(torch.int32).contiguous()
vals = torch.cat([val[int(ptr[0]):int(ptr[1])], val[b0:e0]]).contiguous()
rowt = torch.tensor([0, int(ptr[1]) - int(ptr[0]), int(ptr[1]) - int(ptr[0]) + e0 - b0], dtype=torch.int32, device=dev)
idx = ((ptr[1:] - ptr[:-1]) > 1).to(torch.int32).contiguous()
yref = torch.empty_like(x); y = torch.empty_like(x)
ops.SpMat(ptr, col, val).apply(x, yref)
pp = lambda t: ctypes.c_void_p(t.data_ptr())
cur = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def ccsr(yy): L.spmv_ccsr_f64(0, cur, N, 1.0, 0, pp(idx), 2, pp(rowt), pp(offs), pp(vals), int(rowt[2]), n * n, pp(x), pp(yy))
best = {}
for rnd in range(3):
    for rpl in (2, 0):
        L.spmv_ccsr_set_rows_per_lane(rpl)
        ms = timed(lambda: ccsr(y), 30)
        k = "rows_per_lane_%d" % rpl if rpl else "pair_form"
        best.setdefault(k, {"ms": [], "identical": True})
        best[k]["ms"].append(round(ms, 4)); best[k]["identical"] &= bool(torch.equal(y, yref))
for k, r in best.items():
    r["best_ms"] = min(r["ms"])
    print("ccsr    %-16s %s best %.4f ms identical %s" % (k, r["ms"], r["best_ms"], r["identical"]), flush=True)
out["results"]["ccsr"] = best
L.spmv_ccsr_set_rows_per_lane(0)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r02_spmv_ab.json", "w"), indent=1)
