"""Round 6: SELL-512 products of matrices WITHOUT a traversal of their own (planes that are no multiple of 512 rows), slices dealt to the XCDs as
eight contiguous ranges (traversal.hpp): variable coefficients 500^3 and 384^3 (pair product, values streamed), Poisson 500^3 through the pair
product of value codes, 19-point 320^3 (any-width kernel)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, bench
from vexcl_amd import ops
dev = torch.device("cuda:0")
out = []
def row(name, p, c, v, **kw):
    n = p.numel() - 1
    x = ops.fill_hash(torch.empty(n, dtype=torch.float64, device=dev), 42); y = torch.empty_like(x)
    A = ops.SpMat(p, c, v, **kw)
    A.apply(x, y)
    ms = min(bench.timed_events(torch, lambda: A.apply(x, y), 20) for _ in range(3))
    r = {"row": name, "rows": n, "product": A.product, "ms": round(ms, 4), "sum_y": float(y.sum())}
    print(json.dumps(r), flush=True); out.append(r)
    del A, x, y; torch.cuda.empty_cache()
for g in (500, 384):
    p, c, v = ops.diffusion3d(g, dev); row("variable coefficients %d^3" % g, p, c, v); del p, c, v
p, c, v = ops.poisson3d(500, dev); row("Poisson 500^3, plane=False (pair product of value codes)", p, c, v, plane=False); del p, c, v
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r06_contig%s.json" % os.environ.get("TAG", ""), "w"), indent=1)
