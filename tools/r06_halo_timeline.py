"""Round 6 (pull form; round 5: tools/r05_halo_timeline.py) diagnostics: where the workgroups of the one-launch rank step spend their time (VEXHIP_HALO_DEBUG records: start, ghost flag
seen, first ghost line in registers, end -- 100 MHz ticks -- per workgroup), at the rank-3-of-8 geometry against the own window.
Prints, per role (push, main chunk, lower chunk, upper chunk), the distribution of start / flag / data / end relative to the earliest
start of the launch."""
import ctypes, json, os, sys
os.environ["VEXHIP_HALO_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vexcl_amd import ops, lib
L = lib(); dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
n, world, rank = 512, 8, 3
N = n ** 3; P = n * n
r0, r1 = rank * N // world, (rank + 1) * N // world
rows = r1 - r0
ptr, col, val = ops.poisson3d(n, dev, rows=(r0, r1))
last = ptr[-1:].to(torch.int32)
ptr_ext = torch.cat([torch.zeros(P, dtype=torch.int32, device=dev), ptr.to(torch.int32), last.expand(P)]).contiguous()
col_ext = (col.to(torch.int64) - (r0 - P)).to(torch.int32).contiguous()
ext = ops.SpMat(ptr_ext, col_ext, val, n_cols=rows + 2 * P)
p = lambda t: ctypes.c_void_p(t.data_ptr())
x = ops.fill_hash(torch.empty(rows, dtype=torch.float64, device=dev), 42); y = torch.empty_like(x)
s = torch.cuda.Stream(); sp = ctypes.c_void_p(s.cuda_stream)
PULL = os.environ.get("DIST_MODE", "pull") == "pull"
win = ctypes.c_void_p(); L.ipc_window_create(0, 0, 1, 0 if PULL else 2 * P * 8, ctypes.byref(win))
step = ctypes.c_void_p()
if PULL: L.dist_spmv_create_halo_pull(win, ext.handle, rows, P, 0, 0, 1, ctypes.byref(step))
else: L.dist_spmv_create_halo(win, ext.handle, rows, P, 0, 0, ctypes.byref(step))
xb, xa = ctypes.c_void_p(x.data_ptr() + (rows - P) * 8), ctypes.c_void_p(x.data_ptr())
apply = (lambda: L.dist_spmv_apply_pull(step, sp, 1.0, 0, p(x), p(y), xb, xa)) if PULL else (lambda: L.dist_spmv_apply(step, sp, 1.0, 0, p(x), p(y)))
for _ in range(50):
    apply()
s.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(s):
    e0.record()
    for _ in range(200):
        apply()
    e1.record()
s.synchronize()
print("device us per step (with the diagnostics on)", round(e0.elapsed_time(e1) * 1e3 / 200, 2))
buf = np.zeros(6 * 4096, dtype=np.uint64)
L.dist_spmv_debug(step, ctypes.c_void_p(buf.ctypes.data), buf.nbytes)
rec = buf.reshape(4096, 6)
rec = rec[rec[:, 0] != 0]
t0 = rec[:, 0].min()
us = lambda a: (a.astype(np.float64) - float(t0)) / 100.0
z0 = rec[:, 4].astype(np.int64); z1 = rec[:, 5].astype(np.int64)
roles = {"push": rec[:, 4] == np.uint64(0xFFFFFFFFFFFFFFFF)}
prod = ~roles["push"]
zlo = z0[prod].min(); zhi = z1[prod].max()
roles["lower chunk"] = prod & (z0 == zlo)
roles["upper chunk"] = prod & (z1 == zhi) & (z0 != zlo)
roles["main chunk"] = prod & ~roles["lower chunk"] & ~roles["upper chunk"]
out = {}
for name, m in roles.items():
    if not m.any():
        continue
    r = rec[m]
    d = {"workgroups": int(m.sum()), "planes": [int(z0[m].min()), int(z1[m].max())] if name != "push" else None}
    for k, label in ((0, "start"), (1, "ghost flag seen"), (2, "first ghost line in registers"), (3, "end")):
        v = r[:, k]; v = v[v != 0]
        if v.size:
            u = us(v)
            d[label] = {"min": round(float(u.min()), 1), "median": round(float(np.median(u)), 1), "max": round(float(u.max()), 1)}
    out[name] = d
    print(name, json.dumps(d))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r06_halo_timeline_%s.json" % os.environ.get("DIST_MODE", "pull") + "", "w"), indent=1)
