"""Radix sort timing through the raw C-ABI call on pre-allocated buffers (no allocation in the timed
region); checks sortedness.  Key distributions: full-range u32, 7-bit u32, u32 keys + u32 values, i64."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops, lib, _capi
L = lib(); dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9

def run(keys0, vals0, code, reps=3):
    out = {}
    for variant in (0, 1):
        L.sort_set_rank(variant)
        keys, ktmp = keys0.clone(), torch.empty_like(keys0)
        vals = vals0.clone() if vals0 is not None else None
        vtmp = torch.empty_like(vals0) if vals0 is not None else None
        vb = vals0.element_size() if vals0 is not None else 0
        tmp = torch.empty(L.sort_tmp_bytes(code, keys.numel()), dtype=torch.uint8, device=dev)
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        ms = []
        for r in range(reps + 1):
            keys.copy_(keys0)
            if vals is not None: vals.copy_(vals0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.sort(0, s, code, 0, p(keys), p(ktmp), vb, p(vals), p(vtmp), keys.numel(), p(tmp))
            e1.record(); torch.cuda.synchronize()
            if r: ms.append(e0.elapsed_time(e1))
        out[variant] = (min(ms), keys, vals)
        del ktmp, vtmp, tmp
    return out

def report(name, out, nkeys, check_sorted):
    for variant, label in ((0, "match words"), (1, "atomic ranks")):
        ok = check_sorted(out[variant][1])
        print("%-34s %-13s %8.3f ms  %6.1f Gkeys/s  sorted: %s" % (name, label, out[variant][0], nkeys / out[variant][0] / 1e6, ok), flush=True)
    if out[0][2] is not None:
        print("%-34s payloads identical: %s" % (name, bool(torch.equal(out[0][2], out[1][2]))), flush=True)

def sorted_u32(k):
    a = k.view(torch.int32).to(torch.int64) & 0xffffffff if k.numel() <= 2**28 else None
    if a is not None: return bool((a[1:] >= a[:-1]).all())
    ok = True                      # chunked, to bound temporaries
    step = 2**27
    for i in range(0, k.numel() - 1, step):
        c = k[i:i + step + 1].to(torch.int64) & 0xffffffff
        ok = ok and bool((c[1:] >= c[:-1]).all())
    return ok

keys = ops.fill_hash(torch.empty(n, dtype=torch.int32, device=dev), 42)
report("u32 full range n=%.0e" % n, run(keys, None, _capi.U32), n, sorted_u32)
small = (keys & 0x7f).contiguous()
report("u32 values 0..127 n=%.0e" % n, run(small, None, _capi.U32), n, sorted_u32)
del small
m = n // 4
k2 = keys[:m].contiguous(); v2 = torch.arange(m, dtype=torch.int32, device=dev)
report("u32 keys + u32 values n=%.0e" % m, run(k2, v2, _capi.U32), m, sorted_u32)
k3 = ops.fill_hash(torch.empty(m, dtype=torch.int64, device=dev), 7)
report("i64 keys n=%.0e" % m, run(k3, None, _capi.I64), m, lambda k: bool((k[1:] >= k[:-1]).all()))
L.sort_set_rank(-1)
