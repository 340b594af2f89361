"""Interleaved A/B of sort variants on the SAME buffers (vexhip_sort_set_rank modes): 1e9 u32 keys.
Output: gpurun_out/r02_sort_ab.json"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops, lib, _capi
L = lib(); dev = torch.device("cuda:0")
modes = [int(a) for a in sys.argv[1:]] or [0, 1]      # 0 match words, 1 atomic ranks
n = 10**9
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
keys0 = ops.fill_hash(torch.empty(n, dtype=torch.int32, device=dev), 42)
keys, ktmp = torch.empty_like(keys0), torch.empty_like(keys0)
tmp = torch.empty(L.sort_tmp_bytes(_capi.U32, n), dtype=torch.uint8, device=dev)
ref = None; res = {}
for rnd in range(5):
    for m in modes:
        L.sort_set_rank(m)
        keys.copy_(keys0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.sort(0, s, _capi.U32, 0, p(keys), p(ktmp), 0, None, None, n, p(tmp))
        e1.record(); torch.cuda.synchronize()
        if ref is None: ref = keys.clone()
        r = res.setdefault("mode %d" % m, {"ms": [], "identical": True})
        r["ms"].append(round(e0.elapsed_time(e1), 3)); r["identical"] &= bool(torch.equal(keys, ref))
for k, r in res.items():
    r["best_ms"] = min(r["ms"]); r["gkeys_per_s"] = round(n / r["best_ms"] / 1e6, 1)
    print("%-8s %s best %.3f ms  %.1f Gkeys/s identical %s" % (k, r["ms"], r["best_ms"], r["gkeys_per_s"], r["identical"]), flush=True)
L.sort_set_rank(-1)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/r02_sort_ab.json", "w"), indent=1)
