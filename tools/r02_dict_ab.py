"""Headline product through vex::SpMat's library object at 512^3: slice dictionary on / off (interleaved, bit-identity checked).
Output: gpurun_out/r02_dict_ab.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops
dev = torch.device("cuda:0"); n = 512; N = n ** 3
ptr, col, val = ops.poisson3d(n, device=dev)
nnz = col.numel()
A = ops.SpMat(ptr, col, val); B = ops.SpMat(ptr, col, val, dictionary=False)
del ptr, col, val
A.ptr = A.col = A.val = B.ptr = B.col = B.val = None
torch.cuda.empty_cache()
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42)
ya, yb = torch.empty_like(x), torch.empty_like(x)
res = {"dictionary": {"ms": [], "blocks": A.dictionary_blocks, "matrix_bytes": A.matrix_bytes()},
       "streamed codes": {"ms": [], "blocks": 0, "matrix_bytes": B.matrix_bytes()}}
for rnd in range(4):
    for label, M, y in (("dictionary", A, ya), ("streamed codes", B, yb)):
        for _ in range(60): M.apply(x, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): M.apply(x, y)
        e1.record(); torch.cuda.synchronize()
        res[label]["ms"].append(round(e0.elapsed_time(e1) / 100, 4))
same = bool(torch.equal(ya, yb))
for k, r in res.items():
    r["best_ms"] = min(r["ms"]); r["gflops"] = round(2.0 * nnz / r["best_ms"] / 1e6, 1)
    r["moved_tbps"] = round((r["matrix_bytes"] + 16 * N) / r["best_ms"] / 1e9, 3)
    print(k, r, flush=True)
print("identical:", same)
res["identical"] = same
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/r02_dict_ab.json", "w"), indent=1)
