"""Round 6: A/B of the sort's kernels on 1e9 hashed u32 keys (VEXHIP_SORT_STORE_AUX, VEXHIP_SORT_HIST_UNROLL are read once per process:
one process per variant, run by tools/r06_gpu5.sh), per-kernel times from events around each sort; result checked sorted."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops, lib
dev = torch.device("cuda:0"); L = lib()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 9
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
k = torch.empty(n, dtype=torch.int32, device=dev); ktmp = torch.empty_like(k)
tmp = torch.empty(L.sort_tmp_bytes(3, n), dtype=torch.uint8, device=dev)
best = None
for _ in range(4):
    ops.fill_hash(k, 42); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.sort(0, stream, 3, 0, ctypes.c_void_p(k.data_ptr()), ctypes.c_void_p(ktmp.data_ptr()), 0, None, None, n, ctypes.c_void_p(tmp.data_ptr()))
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1); best = t if best is None else min(best, t)
u = k.view(torch.int32)
# sortedness as unsigned: compare neighbours in chunks
ok = True
for a in range(0, n - 1, 1 << 28):
    b = min(n, a + (1 << 28) + 1)
    c = k[a:b].to(torch.int64) & 0xFFFFFFFF
    ok = ok and bool((c[1:] >= c[:-1]).all())
print(json.dumps({"env": {e: os.environ[e] for e in os.environ if e.startswith("VEXHIP_SORT")}, "ms": round(best, 3), "gkeys_per_s": round(n / best / 1e6, 1), "sorted": ok}))
