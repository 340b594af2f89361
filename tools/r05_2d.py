"""Round 5: 5-point operators on 2-D grids through the default vexhip_spmat (virtual 512-point lines, plane product) against the pair
product of the SELL-512 storage: time, bytes moved, bit-identity; the check against torch slicing of the grid."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import unstructured as U
from vexcl_amd import ops
import bench
dev = torch.device("cuda:0")
out = {}
for W, H in ((16384, 16384), (10000, 10000), (12000, 9000), (7000, 20000)):
    ptr, col, val, h2i = U.stencil2d(W, H, dev)
    n, nnz = W * H, int(col.numel())
    x = ops.fill_hash(torch.empty(n, dtype=torch.float64, device=dev), 42); y = torch.empty_like(x)
    A = ops.SpMat(ptr, col, val)
    A.apply(x, y)
    yr, mag = U.stencil2d_reference(x, W, H, h2i)
    bad = int(((y - yr).abs() > 1e-10 * mag).sum())
    del yr, mag
    t = min(bench.timed_events(torch, lambda: A.apply(x, y), 20) for _ in range(3))
    moved = A.matrix_bytes() + 16 * n
    row = {"rows": n, "nnz": nnz, "storage": A.storage, "plane": A.plane, "grid": A.grid, "ms": round(t, 5), "gflops": round(2.0 * nnz / t / 1e6, 1),
           "bytes_moved": moved, "frac_of_8TBps": round(moved / t / 1e6 / 8000.0, 4), "rows_outside_tolerance": bad}
    del A
    torch.cuda.empty_cache()
    B = ops.SpMat(ptr, col, val, march=False)
    yb = torch.empty_like(y)
    tb = min(bench.timed_events(torch, lambda: B.apply(x, yb), 10) for _ in range(2))
    row["pair_product_ms"] = round(tb, 5); row["bit_identical_to_pair_product"] = bool(torch.equal(y, yb))
    out["%d x %d" % (W, H)] = row
    print(W, H, row, flush=True)
    del B, ptr, col, val, x, y, yb
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r05_2d.json", "w"), indent=1)
