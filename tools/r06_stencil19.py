"""Round 6: a constant-coefficient 19-point operator (no corners: five triples and four single columns per row) through the default
vexhip_spmat (the runs product, sell8.hip) against the any-width kernel it replaced (variant 1), bit-identity between the two."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, bench
from vexcl_amd import ops, lib
dev = torch.device("cuda:0")
out = []
def stencil_const(g, points):
    N = g ** 3
    r = torch.arange(N, device=dev, dtype=torch.int32)
    ix, iy, iz = r % g, (r // g) % g, r // (g * g)
    inner = (ix > 0) & (ix < g - 1) & (iy > 0) & (iy < g - 1) & (iz > 0) & (iz < g - 1)
    del ix, iy, iz
    offs = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if points == 27 or abs(dx) + abs(dy) + abs(dz) <= 2]
    ptr64 = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.where(inner, len(offs), 1), 0, out=ptr64[1:])
    nnz = int(ptr64[-1])
    col = torch.empty(nnz, dtype=torch.int32, device=dev); val = torch.empty(nnz, dtype=torch.float64, device=dev)
    b = ptr64[:-1]; bi, ri = b[inner], r[inner]
    for k, (dx, dy, dz) in enumerate(offs):
        col[bi + k] = ri + (dz * g * g + dy * g + dx)
        val[bi + k] = float(len(offs) - 1) if (dx, dy, dz) == (0, 0, 0) else -1.0
    col[b[~inner]] = r[~inner]; val[b[~inner]] = 1.0
    return ptr64.to(torch.int32), col, val
for points in (19, 27):
    for g in (256, 320):
        p, c, v = stencil_const(g, points)
        n = p.numel() - 1
        x = ops.fill_hash(torch.empty(n, dtype=torch.float64, device=dev), 42); y = torch.empty_like(x); y1 = torch.empty_like(x)
        A = ops.SpMat(p, c, v)
        A.apply(x, y)
        ms = min(bench.timed_events(torch, lambda: A.apply(x, y), 20) for _ in range(3))
        prod = A.product
        lib().spmv_sell8_set_variant(1)
        A.apply(x, y1)
        ms1 = min(bench.timed_events(torch, lambda: A.apply(x, y1), 20) for _ in range(3))
        prod1 = A.product
        lib().spmv_sell8_set_variant(0)
        r = {"row": "%d-point constant coefficients %d^3" % (points, g), "rows": n, "product": prod, "ms": round(ms, 4), "codes_path": prod1, "codes_path_ms": round(ms1, 4), "same_bits": bool(torch.equal(y, y1)),
             "frac_of_8TBps_x_and_y": round(16.0 * n / ms / 1e6 / 8000, 3)}
        print(json.dumps(r), flush=True); out.append(r)
        del A, p, c, v, x, y, y1; torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r06_stencil19.json", "w"), indent=1)
