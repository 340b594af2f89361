"""Round 5: where fp32 stands -- the 512^3 Poisson matrix with float values through the default vexhip_spmat (the plane / grid
products are fp64 only: the value-coded SELL-512 storage with the march or pair product takes it), against the fp64 product."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops
import bench
dev = torch.device("cuda:0")
n = 512; N = n ** 3
ptr, col, val = ops.poisson3d(n, dev)
out = {}
for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
    v = val.to(dt)
    x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42).to(dt); y = torch.empty_like(x)
    A = ops.SpMat(ptr, col, v)
    A.apply(x, y)
    t = min(bench.timed_events(torch, lambda: A.apply(x, y), 20) for _ in range(3))
    moved = A.matrix_bytes() + 2 * x.element_size() * N
    out[name] = {"storage": A.storage, "plane": bool(A.plane), "march": A.march, "dictionary_blocks": A.dictionary_blocks, "ms": round(t, 5),
                 "gflops": round(2.0 * col.numel() / t / 1e6, 1), "bytes_moved": moved, "frac_of_8TBps": round(moved / t / 1e6 / 8000.0, 4)}
    print(name, out[name], flush=True)
    del A, x, y, v
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r05_fp32.json", "w"), indent=1)
