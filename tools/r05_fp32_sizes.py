"""Round 5: where fp32 stands on grids whose lines are not 512 points (no fp32 grid product: the march / pair products of the SELL-512 storage)."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vexcl_amd import ops
import bench
dev = torch.device("cuda:0")
out = {}
for g in [int(a) for a in sys.argv[1:]] or [384, 500, 640]:
    N = g ** 3
    p, c, v = ops.poisson3d(g, dev)
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        vv = v.to(dt)
        x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 7).to(dt); y = torch.empty_like(x)
        A = ops.SpMat(p, c, vv)
        A.apply(x, y)
        t = min(bench.timed_events(torch, lambda: A.apply(x, y), 10) for _ in range(2))
        out["%d %s" % (g, name)] = {"ms": round(t, 4), "frac": round((A.matrix_bytes() + 2 * x.element_size() * N) / t / 1e6 / 8000, 4),
                                    "product": "grid" if A.grid and not A.plane else "plane" if A.plane else "march" if A.march else "pair", "dict": A.dictionary_blocks}
        print(g, name, out["%d %s" % (g, name)], flush=True)
        del A, x, y, vv
    del p, c, v; torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r05_fp32_sizes.json", "w"), indent=1)
