"""1e9 u32 keys: vexhip_sort per rank scheme.  -1 default (= 6: half-wave units, one returning 64-bit atomic per key that proves its own
order); 5 lean scatter, ranks taken from the order words (every key of every
tile checked); 4 lean, counter atomics checked by the order words; 3 lean, unchecked counter atomics (A/B only); 1 round 4's
kernel (counter atomics, one verified tile in 16); 0 match words.  Every scheme must give the bits of the first.  Also: keys whose
two upper digits are constant (tiles copied as blocks), and sort_by_key u32 -> u32 at 2.5e8.
torch.sort is a PAIRS sort (int32 keys + int64 indices): printed as such, not as a like-for-like yardstick."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops, lib
dev = torch.device("cuda:0"); L = lib()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 9
modes = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1, 6, 5, 4, 3, 1, 0]
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
k = torch.empty(n, dtype=torch.int32, device=dev); ktmp = torch.empty_like(k)
tmp = torch.empty(L.sort_tmp_bytes(3, n), dtype=torch.uint8, device=dev)
out = {}


def run(keys, vb=0, vals=None, vtmp=None, reps=3, fill=None):
    best = None
    for _ in range(reps):
        fill(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.sort(0, stream, 3, 0, ctypes.c_void_p(keys.data_ptr()), ctypes.c_void_p(ktmp.data_ptr()), vb,
               ctypes.c_void_p(vals.data_ptr()) if vals is not None else None, ctypes.c_void_p(vtmp.data_ptr()) if vtmp is not None else None,
               keys.numel(), ctypes.c_void_p(tmp.data_ptr()))
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1); best = t if best is None else min(best, t)
    return best


ref = None
for mode in modes:
    L.sort_set_rank(mode)
    try:
        best = run(k, fill=lambda: ops.fill_hash(k, 42))
    except Exception as e:                       # a trap surfaces here
        out["rank mode %d" % mode] = {"error": str(e)[:200]}
        break
    if ref is None:
        ref = k.clone()
        srt = bool((ref[1:].view(torch.int32).to(torch.int64) & 0xffffffff >= (ref[:-1].to(torch.int64) & 0xffffffff)).all())
    out["rank mode %d" % mode] = {"ms": round(best, 3), "gkeys_per_s": round(n / best / 1e6, 1), "same_as_first": bool(torch.equal(k, ref)), "sorted": srt}
    print(mode, out["rank mode %d" % mode], flush=True)
# small key range: digits 2 and 3 constant
for mode in [m for m in modes if m in (-1, 1)]:
    L.sort_set_rank(mode)
    def fill16():
        ops.fill_hash(k, 7); k.bitwise_and_(0xffff)
    best = run(k, fill=fill16)
    out["keys < 2^16, rank mode %d" % mode] = {"ms": round(best, 3), "gkeys_per_s": round(n / best / 1e6, 1)}
    print(out["keys < 2^16, rank mode %d" % mode], flush=True)
del ref
# pairs
m = min(n, 250_000_000)
kk = k[:m]; v = torch.empty(m, dtype=torch.int32, device=dev); vt = torch.empty_like(v)
pref = None
for mode in [x for x in modes if x in (-1, 1, 0)]:
    L.sort_set_rank(mode)
    def fillp():
        ops.fill_hash(kk, 11); kk.bitwise_and_(0xfffff); v.copy_(torch.arange(m, dtype=torch.int32, device=dev))
    best = run(kk, 4, v, vt, fill=fillp)
    if pref is None: pref = v.clone()
    out["sort_by_key u32->u32 n=%d, rank mode %d" % (m, mode)] = {"ms": round(best, 3), "gpairs_per_s": round(m / best / 1e6, 1), "same_as_first": bool(torch.equal(v, pref))}
    print(out["sort_by_key u32->u32 n=%d, rank mode %d" % (m, mode)], flush=True)
L.sort_set_rank(-1)
del ktmp, tmp, pref, v, vt
torch.cuda.empty_cache()
if os.environ.get("SORT_TORCH", "1") == "1":
    ops.fill_hash(k, 42); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); s = torch.sort(k); e1.record(); torch.cuda.synchronize()
    out["torch.sort = rocPRIM PAIRS sort (int32 keys + int64 indices; not like-for-like)"] = {"ms": round(e0.elapsed_time(e1), 3)}
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r05_sort_time.json", "w"), indent=1)
