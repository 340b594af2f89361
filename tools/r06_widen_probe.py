"""Round 6: what the structured matrices OUTSIDE the grid storage's pattern get today: constant-coefficient 27-point on g^3, 2-D 5-point on
rows of any length.  Storage / product chosen, time, bytes moved (stored matrix + x once + y once), fraction of 8 TB/s."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, bench
from vexcl_amd import ops
import unstructured as U
dev = torch.device("cuda:0")
out = []

def stencil27_const(g):
    N = g ** 3
    r = torch.arange(N, device=dev, dtype=torch.int32)
    ix, iy, iz = r % g, (r // g) % g, r // (g * g)
    inner = (ix > 0) & (ix < g - 1) & (iy > 0) & (iy < g - 1) & (iz > 0) & (iz < g - 1)
    del ix, iy, iz
    ptr64 = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.where(inner, 27, 1), 0, out=ptr64[1:])
    nnz = int(ptr64[-1])
    col = torch.empty(nnz, dtype=torch.int32, device=dev); val = torch.empty(nnz, dtype=torch.float64, device=dev)
    b = ptr64[:-1]; bi, ri = b[inner], r[inner]
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                col[bi + k] = ri + (dz * g * g + dy * g + dx)
                val[bi + k] = 26.0 if (dx, dy, dz) == (0, 0, 0) else -1.0
                k += 1
    col[b[~inner]] = r[~inner]; val[b[~inner]] = 1.0
    return ptr64.to(torch.int32), col, val

def row(name, p, c, v):
    n = p.numel() - 1
    x = ops.fill_hash(torch.empty(n, dtype=v.dtype, device=dev), 42); y = torch.empty_like(x)
    A = ops.SpMat(p, c, v)
    yr, mag = U.reference_product(p, c, v, x)
    A.apply(x, y)
    err = float(((y - yr).abs() / mag.clamp_min(1e-300)).max())
    del yr, mag
    ms = min(bench.timed_events(torch, lambda: A.apply(x, y), 20) for _ in range(3))
    moved = int(A.info.matrix_bytes) + 2 * x.element_size() * n
    csr = 12 * c.numel() + 4 * (n + 1) + 16 * n
    r = {"row": name, "rows": n, "nnz": int(c.numel()), "storage": A.storage, "product": A.product, "reason": A.reason, "ms": round(ms, 4), "matrix_bytes": int(A.info.matrix_bytes),
         "moved_gbps": round(moved / ms / 1e6, 1), "frac_of_8TBps_moved": round(moved / ms / 1e6 / 8000, 3), "csr_bytes_frac": round(csr / ms / 1e6 / 8000, 3), "max_rel_err": err}
    print(json.dumps(r), flush=True); out.append(r)

which = os.environ.get("WIDEN", "s27,2d").split(",")
if "s27" in which:
    for g in (256, 320):
        row("27-point constant coefficients %d^3" % g, *stencil27_const(g)); torch.cuda.empty_cache()
if "2d" in which:
    for W, H in ((12000, 12000), (10000, 16384), (16384, 16384), (4097, 30000)):
        p, c, v, _ = U.stencil2d(W, H, dev)
        row("5-point 2-D %d x %d" % (W, H), p, c, v); del p, c, v; torch.cuda.empty_cache()
json.dump(out, open("gpurun_out/r06_widen_probe.json", "w"), indent=1)
