"""Round-4 prototype driver: hand copy kernels in several geometries and the plane-march data flow (tools/r04_pm_proto.hip)
on the 512^3 Poisson problem, timed next to torch's copy and the library's march product; every plane-march variant is
compared bit for bit with the library product.  JSON on stdout (profiles/r04_pm_proto.json)."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vexcl_amd import lib, ops  # noqa: E402

L = lib()
dev = torch.device("cuda:0")
n = 512
N = n ** 3
SRC = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "r04_pm_proto.hip")).read()
mod = ctypes.c_void_p()
L.module_compile(0, SRC.encode(), b"", ctypes.byref(mod))
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42)
y = torch.empty(N, dtype=torch.float64, device=dev)
yref = torch.empty_like(y)
p, c, v = ops.poisson3d(n, dev)
A = ops.SpMat(p, c, v)
del p, c, v
A.ptr = A.col = A.val = None
torch.cuda.empty_cache()
A.apply(x, yref)


def timed(fn, reps=30, rounds=3):
    best = 1e30
    for _ in range(rounds):
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def kernel(name):
    fn = ctypes.c_void_p()
    L.module_get_function(0, mod, name.encode(), ctypes.byref(fn))
    return fn


def launcher(name, grid, args):
    fn = kernel(name)
    keep = list(args)
    arr = (ctypes.c_void_p * len(keep))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in keep])
    def run():
        L.launch(0, fn, grid, 1, 1, 256, 1, 1, 0, stream, arr)
    run.keep = (keep, arr, fn)
    return run


out = {"grid": n, "bytes_priced": 16 * N, "rows": []}
def row(name, ms, **kw):
    r = {"name": name, "ms": round(ms, 4), "TBps_priced": round(16 * N / ms / 1e9, 3)}
    r.update(kw)
    out["rows"].append(r)
    print(json.dumps(r), file=sys.stderr, flush=True)


px, py = ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr())
npairs = ctypes.c_longlong(N // 2)
row("torch_copy", timed(lambda: y.copy_(x)))
row("library_march", timed(lambda: A.apply(x, y)))
for nt in (0, 1):
    row("copy_1 nt=%d" % nt, timed(launcher("copy_1", N // 2 // 256, [px, py, npairs, ctypes.c_int(nt)])))
for name, u in (("copy_u2", 2), ("copy_u4", 4), ("copy_u4_plain", 4), ("copy_u4_ntl", 4)):
    row(name, timed(launcher(name, N // 2 // 256 // u, [px, py, npairs])))
for g in (1024, 2048, 4096, 8192):
    row("copy_gs grid=%d" % g, timed(launcher("copy_gs", g, [px, py, npairs, ctypes.c_int(1)])))
for run_, bar in ((32, 1), (32, 0), (64, 0)):
    row("copy_march run=%d barrier=%d" % (run_, bar),
        timed(launcher("copy_march", N // 512 // run_, [px, py, ctypes.c_longlong(N // 512), ctypes.c_int(run_), ctypes.c_int(bar)])))
for ty in (2, 4):
    for lz in (64, 128, 256):
        row("copy_pm%d LZ=%d" % (ty, lz), timed(launcher("copy_pm%d" % ty, n * n // ty // lz, [px, py, ctypes.c_int(n), ctypes.c_int(lz)])))
y.copy_(x)
assert torch.equal(y, x)

alpha = ctypes.c_double(1.0)
for ty in (1, 2, 4):
    for lz in (32, 64, 128, 256, 512):
        grid = n * n // ty // lz
        if grid < 256 or grid > 4096:
            continue
        for flags in (0, 1, 2, 3):
            name = "pm%d" % ty
            run = launcher(name, grid, [px, py, ctypes.c_int(n), ctypes.c_int(lz), ctypes.c_int(flags), alpha])
            y.zero_()
            run(); torch.cuda.synchronize()
            same = bool(torch.equal(y, yref))
            nbad = 0 if same else int((y != yref).sum().item())
            row("%s LZ=%d flags=%d" % (name, lz, flags), timed(run), grid=grid, bit_identical=same, mismatches=nbad)
for ty in (2, 4):
    for lz in (64, 128):
        run = launcher("pm%d_nocompute" % ty, n * n // ty // lz, [px, py, ctypes.c_int(n), ctypes.c_int(lz), ctypes.c_int(1), alpha])
        row("pm%d_nocompute LZ=%d flags=1" % (ty, lz), timed(run))
row("torch_copy (end)", timed(lambda: y.copy_(x)))
row("library_march (end)", timed(lambda: A.apply(x, y)))
print(json.dumps(out))
