"""Round 5: the product step of ONE rank at the 8-GPU geometry (rank 3 of 8 of the 512^3 Poisson problem: 16 777 216 rows, one
ghost plane per z-neighbour) on the one GPU of the gpurun box, the rank exchanging with ITSELF through its own window
(tools/r04_dist_step.py explains the stand-in).  Three steps side by side, same x, same matrix entries:
  ipc   -- round 3/4: push kernel on a second stream | local part -> wait kernel -> remote part -> signal kernel (issued directly
           and as a hipGraph);
  halo  -- round 5 (csrc/halo.hpp): ONE launch -- the strip stored with its two ghost planes, the plane product reads them from the
           window, its first workgroups push the boundary planes -- plus the one-thread kernel that raises `consumed`;
  parts -- the local plane product alone, the stored strip's plane product alone (ghost planes never awaited: flags pre-raised).
The two steps must give the same bits (the ghost values are the same numbers: own last / first plane).
Output: JSON on stdout (profiles/r05_dist_step.json)."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops, lib, _capi

L = lib(); dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
n = int(os.environ.get("DIST_GRID", "512")); world, rank = 8, 3
N = n ** 3
P = n * n
r0, r1 = rank * N // world, (rank + 1) * N // world
rows = r1 - r0
ptr, col, val = ops.poisson3d(n, dev, rows=(r0, r1))
is_loc = (col >= r0) & (col < r1)
ghosts = torch.unique(col[~is_loc].to(torch.int64))
row_of = torch.repeat_interleave(torch.arange(rows, device=dev), (ptr[1:] - ptr[:-1]).to(torch.int64))


def sub(mask, cols):
    cnt = torch.bincount(row_of[mask], minlength=rows)
    p = torch.zeros(rows + 1, dtype=torch.int64, device=dev); p[1:] = torch.cumsum(cnt, 0)
    return p.to(torch.int32), cols.to(torch.int32).contiguous(), val[mask].contiguous()


lp, lc, lv = sub(is_loc, col[is_loc] - r0)
loc = ops.SpMat(lp, lc, lv, n_cols=rows)
rp, rc, rv = sub(~is_loc, torch.searchsorted(ghosts, col[~is_loc].to(torch.int64)))
cnt = rp[1:] - rp[:-1]
rows_with = torch.nonzero(cnt > 0).flatten().to(torch.int32)
cp = torch.zeros(rows_with.numel() + 1, dtype=torch.int32, device=dev); cp[1:] = torch.cumsum(cnt[rows_with.long()], 0).to(torch.int32)
ng = int(ghosts.numel())
send_idx = (ghosts % rows).to(torch.int32).contiguous()         # lower ghosts = own last plane, upper ghosts = own first plane
# the strip stored WITH its ghost planes: P empty rows, the rows, P empty rows; columns from the first element of the lower ghost plane
last = ptr[-1:].to(torch.int32)
ptr_ext = torch.cat([torch.zeros(P, dtype=torch.int32, device=dev), ptr.to(torch.int32), last.expand(P)]).contiguous()
col_ext = (col.to(torch.int64) - (r0 - P)).to(torch.int32).contiguous()
if os.environ.get("VEXHIP_PLANE_FORCE"):
    pass
ext = ops.SpMat(ptr_ext, col_ext, val, n_cols=rows + 2 * P)
ext_ptr, ext_col, ext_val = ptr_ext, col_ext, val
del row_of, is_loc, ptr, col, val, ptr_ext, col_ext
p = lambda t: ctypes.c_void_p(t.data_ptr())
x = ops.fill_hash(torch.empty(rows, dtype=torch.float64, device=dev), 42); y = torch.empty_like(x)
s = torch.cuda.Stream(); sp = ctypes.c_void_p(s.cuda_stream)
out = {"grid": n, "strip_rows": rows, "local_storage": loc.storage, "local_plane_plan": loc.plane, "ghosts": ng, "remote_rows": int(rows_with.numel()),
       "stored_strip": {"rows": rows + 2 * P, "storage": ext.storage, "plane_plan": ext.plane, "grid_plan": ext.grid}}
assert ext.plane, "the stored strip did not get a plane plan"
cnts = (ctypes.c_int64 * 1)(ng); zero = (ctypes.c_int64 * 1)(0)


def bench(fn, label, reps=400):
    for _ in range(10):
        fn()
    s.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        host = (time.perf_counter() - t0) / reps
        with torch.cuda.stream(s):
            e1.record()
        s.synchronize()
        r = {"host_us_per_step": round(host * 1e6, 2), "device_us_per_step": round(e0.elapsed_time(e1) * 1e3 / reps, 2)}
        if best is None or r["device_us_per_step"] < best["device_us_per_step"]:
            best = r
    out[label] = best
    print(label, best, file=sys.stderr, flush=True)


def status(step):
    to = ctypes.c_int(); tr = ctypes.c_int(); L.dist_spmv_status(step, ctypes.byref(to), ctypes.byref(tr), None)
    return to.value, tr.value


# ---- round 3/4: IPC step over the own window
win = ctypes.c_void_p(); L.ipc_window_create(0, 0, 1, ng * 8, ctypes.byref(win))
step = ctypes.c_void_p()
L.dist_spmv_create_ipc(win, _capi.F64, rows, loc.handle, rows_with.numel(), p(rows_with), p(cp), p(rc), p(rv), ng, p(send_idx), cnts, zero, ng, cnts, ctypes.byref(step))
bench(lambda: L.dist_spmv_apply(step, sp, 1.0, 0, p(x), p(y)), "ipc, issued directly")
y_ipc = y.clone()
L.dist_spmv_set_graph(step, 1)
bench(lambda: L.dist_spmv_apply(step, sp, 1.0, 0, p(x), p(y)), "ipc, hipGraph replay")
L.dist_spmv_set_graph(step, 0)
ms = (ctypes.c_float * 6)()
reps = []
for _ in range(7):
    L.dist_spmv_profile(step, sp, 1.0, 0, p(x), p(y), ms); reps.append(list(ms))
reps.sort(key=lambda r: r[0])
out["ipc step phases (ms, median of 7)"] = dict(zip(("total", "local part", "wait for ghosts", "remote part", "pack", "push"), [round(v, 5) for v in reps[3]]))
out["ipc timed_out"] = status(step)[0]
L.dist_spmv_destroy(step); L.ipc_window_destroy(win)

# ---- round 5: the step as one launch
win = ctypes.c_void_p(); L.ipc_window_create(0, 0, 1, 2 * P * 8, ctypes.byref(win))
step = ctypes.c_void_p()
L.dist_spmv_create_halo(win, ext.handle, rows, P, 0, 0, ctypes.byref(step))
y.fill_(float("nan"))
bench(lambda: L.dist_spmv_apply(step, sp, 1.0, 0, p(x), p(y)), "halo: one product launch + one signal launch per step")
out["halo timed_out, transport"] = list(status(step))
d = (y - y_ipc).abs()
out["halo_equals_ipc"] = bool(torch.equal(y, y_ipc))            # (not expected: the split step adds the remote entries LAST, the stored strip has them in column order)
out["halo_vs_ipc_max_abs_diff"] = float(d.max())
# the bits of the ONE-device product: the same stored strip through the CSR kernel on x with its ghost planes attached
x_ext = torch.cat([x[rows - P:], x, x[:P]]).contiguous()
y_ext = torch.empty(rows + 2 * P, dtype=torch.float64, device=dev)
ref = ops.SpMat(ext_ptr, ext_col, ext_val, n_cols=rows + 2 * P, fmt="csr")
with torch.cuda.stream(s):
    ref.apply(x_ext, y_ext)
s.synchronize()
y_one = y_ext[P:P + rows].clone()
out["halo_equals_one_device_csr_order"] = bool(torch.equal(y, y_one))
out["halo_vs_one_device_max_abs_diff"] = float((y - y_one).abs().max())
del ref, x_ext, y_ext
# '+=' and a second alpha through the same step
y2 = y_ipc.clone()
torch.cuda.synchronize()
with torch.cuda.stream(s):
    L.dist_spmv_apply(step, sp, 0.5, 1, p(x), p(y2))
s.synchronize()
out["halo_append_equals_reference"] = bool(torch.equal(y2, y_ipc + 0.5 * y_one))
L.dist_spmv_set_graph(step, 1)
bench(lambda: L.dist_spmv_apply(step, sp, 1.0, 0, p(x), p(y)), "halo, hipGraph replay")
out["halo_graph_equals_one_device"] = bool(torch.equal(y, y_one))
L.dist_spmv_set_graph(step, 0)
L.dist_spmv_destroy(step); L.ipc_window_destroy(win)

# ---- the parts alone
rem = ops.RowSubsetCSR(rows_with, cp, rc, rv)
ghost_buf = torch.zeros(ng, dtype=torch.float64, device=dev)
x_ext = torch.cat([x[rows - P:], x, x[:P]]).contiguous()
y_ext = torch.empty(rows + 2 * P, dtype=torch.float64, device=dev)
with torch.cuda.stream(s):
    for label, fn in (("local part alone", lambda: loc.apply(x, y)),
                      ("stored strip (66 planes) through the ordinary plane product, ghost planes part of x", lambda: ext.apply(x_ext, y_ext)),
                      ("local + remote part, no exchange", lambda: (loc.apply(x, y), rem.apply(ghost_buf, y, 1.0, True)))):
        bench(fn, label)
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r05_dist_step.json", "w"), indent=1)
