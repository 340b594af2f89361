"""Round 6: 5-point operators on 2-D grids whose rows are NOT multiples of 512 points through the grid product on virtual lines (a FLAT plan:
the walk requests no neighbour lines) against what they took until now (VEXHIP_GRID_2D_LINE=0: the march / pair product of the SELL-512
storage); sweep of the virtual line length; bit-identity with the pair product, the check against torch slicing of the grid."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import unstructured as U
from vexcl_amd import ops
import bench
dev = torch.device("cuda:0")
out = {}
shapes = [(12000, 12000, (None, 0, 1000, 750, 600, 500, 400, 250)), (10000, 10000, (None, 0, 500)), (12000, 9000, (None, 0)), (7000, 20000, (None, 0)), (9999, 9999, (None,))]
if os.environ.get("SHAPES"): shapes = shapes[:int(os.environ["SHAPES"])]
for W, H, lines in shapes:
    ptr, col, val, h2i = U.stencil2d(W, H, dev)
    n, nnz = W * H, int(col.numel())
    x = ops.fill_hash(torch.empty(n, dtype=torch.float64, device=dev), 42); y = torch.empty_like(x)
    B = ops.SpMat(ptr, col, val, march=False)
    yb = torch.empty_like(y)
    B.apply(x, yb)
    yr, mag = U.stencil2d_reference(x, W, H, h2i)
    badb = int(((yb - yr).abs() > 1e-10 * mag).sum())
    del yr, mag
    tb = min(bench.timed_events(torch, lambda: B.apply(x, yb), 10) for _ in range(2))
    del B
    rows = {"pair_product_ms": round(tb, 5), "pair_rows_outside_tolerance": badb}
    for ln in lines:
        if ln is None: os.environ.pop("VEXHIP_GRID_2D_LINE", None)
        else: os.environ["VEXHIP_GRID_2D_LINE"] = str(ln)
        A = ops.SpMat(ptr, col, val)
        y.fill_(-1.0)
        A.apply(x, y)
        t = min(bench.timed_events(torch, lambda: A.apply(x, y), 20) for _ in range(3))
        moved = A.matrix_bytes() + 16 * n
        g = A.grid
        row = {"storage": A.storage, "product": A.product, "grid": None if not g else {k: g[k] for k in ("nx", "lines_per_plane", "planes", "depth", "threads", "classes", "flat") if k in g},
               "ms": round(t, 5), "gflops": round(2.0 * nnz / t / 1e6, 1), "bytes_moved": moved, "frac_of_8TBps": round(moved / t / 1e6 / 8000.0, 4),
               "bit_identical_to_pair_product": bool(torch.equal(y, yb))}
        # y += 0.5 A x and the float matrix
        y2 = yb.clone(); A.apply(x, y2, 0.5, True)
        row["append_bit_identical"] = bool(torch.equal(y2, yb + 0.5 * yb)) if False else None
        rows["line %s" % ("auto" if ln is None else ln)] = row
        print(W, H, ln, row, flush=True)
        del A
        torch.cuda.empty_cache()
    os.environ.pop("VEXHIP_GRID_2D_LINE", None)
    # float
    valf = val.float(); xf = x.float(); yf = torch.empty_like(xf); ybf = torch.empty_like(xf)
    Af = ops.SpMat(ptr, col, valf); Af.apply(xf, yf)
    tf = min(bench.timed_events(torch, lambda: Af.apply(xf, yf), 20) for _ in range(3))
    os.environ["VEXHIP_GRID_2D_LINE"] = "0"
    Bf = ops.SpMat(ptr, col, valf); Bf.apply(xf, ybf)
    tbf = min(bench.timed_events(torch, lambda: Bf.apply(xf, ybf), 10) for _ in range(2))
    os.environ.pop("VEXHIP_GRID_2D_LINE", None)
    rows["float"] = {"product": Af.product, "ms": round(tf, 5), "frac_of_8TBps": round((Af.matrix_bytes() + 8 * n) / tf / 1e6 / 8000.0, 4), "round5_product": Bf.product, "round5_ms": round(tbf, 5),
                     "bit_identical": bool(torch.equal(yf, ybf))}
    print(W, H, "float", rows["float"], flush=True)
    del Af, Bf
    out["%d x %d" % (W, H)] = rows
    del ptr, col, val, x, y, yb
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r06_2d.json", "w"), indent=1)
