"""Host and device cost of ONE rank's product step over the IPC transport at the 8-GPU geometry, on the one GPU of the gpurun box.

The strip is rank 3 of 8 of the 512^3 Poisson problem (16 777 216 rows, one ghost plane from each z-neighbour: 2 x 262 144
columns).  With one GPU there is nobody to exchange with, so the rank exchanges WITH ITSELF through its own ghost window (the
whole protocol -- push kernel, arrive / consumed flags, wait and signal kernels -- with one participant; the bytes cross no
link).  The numbers in y are therefore not the 8-GPU product's: this tool measures what a step COSTS -- host time per
vexhip_dist_spmv_apply call and device time per step, issued directly and replayed from a hipGraph -- next to the same step
over RCCL (send/recv to self) and the parts alone.  Output: JSON on stdout (profiles/r04_dist_step.json)."""
import ctypes, json, os, sys, time
os.environ["VEXHIP_RCCL_SELF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops, lib, _capi

L = lib(); dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
n, world, rank = 512, 8, 3
N = n ** 3
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
# what the rank sends: the plane next to each of its boundaries (the two neighbours of a plane partition each get ONE run of
# consecutive elements, sent straight out of x); here both runs go to the one participant
send_idx = (ghosts % rows).to(torch.int32).contiguous()
del row_of, is_loc, ptr, col, val
p = lambda t: ctypes.c_void_p(t.data_ptr())
x = ops.fill_hash(torch.empty(rows, dtype=torch.float64, device=dev), 42); y = torch.empty_like(x)
s = torch.cuda.Stream(); sp = ctypes.c_void_p(s.cuda_stream)
out = {"strip_rows": rows, "local_storage": loc.storage, "local_plane_plan": loc.plane, "local_march_plan": loc.march, "ghosts": ng,
       "remote_rows": int(rows_with.numel()), "exchange_bytes_each_way": ng * 8}
cnts = (ctypes.c_int64 * 1)(ng); zero = (ctypes.c_int64 * 1)(0)


def bench(step, label, reps=400):
    for _ in range(10):
        L.dist_spmv_apply(step, sp, 1.0, 0, p(x), p(y))
    s.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record()
        t0 = time.perf_counter()
        for _ in range(reps):
            L.dist_spmv_apply(step, sp, 1.0, 0, p(x), p(y))
        host = (time.perf_counter() - t0) / reps
        with torch.cuda.stream(s):
            e1.record()
        s.synchronize()
        r = {"host_us_per_step": round(host * 1e6, 2), "device_us_per_step": round(e0.elapsed_time(e1) * 1e3 / reps, 2)}
        if best is None or r["device_us_per_step"] < best["device_us_per_step"]:
            best = r
    out[label] = best
    print(label, best, file=sys.stderr, flush=True)


# ---- IPC: own window
win = ctypes.c_void_p(); L.ipc_window_create(0, 0, 1, ng * 8, ctypes.byref(win))
step = ctypes.c_void_p()
L.dist_spmv_create_ipc(win, _capi.F64, rows, loc.handle, rows_with.numel(), p(rows_with), p(cp), p(rc), p(rv), ng, p(send_idx), cnts, zero, ng, cnts, ctypes.byref(step))
bench(step, "ipc, issued directly")
yd = y.clone()
L.dist_spmv_set_graph(step, 1)
bench(step, "ipc, hipGraph replay")
assert torch.equal(y, yd), "graph replay and direct issue differ"
L.dist_spmv_set_graph(step, 0)
ms = (ctypes.c_float * 6)()
reps = []
for _ in range(7):
    L.dist_spmv_profile(step, sp, 1.0, 0, p(x), p(y), ms); reps.append(list(ms))
reps.sort(key=lambda r: r[0])
out["ipc step phases (ms, median of 7)"] = dict(zip(("total", "local part", "wait for ghosts", "remote part", "pack", "push"), [round(v, 5) for v in reps[3]]))
to = ctypes.c_int(); L.dist_spmv_status(step, ctypes.byref(to), None, None)
out["ipc timed_out"] = to.value
L.dist_spmv_destroy(step); L.ipc_window_destroy(win)

# ---- RCCL: send/recv to self
send_buf = torch.empty(ng, dtype=torch.float64, device=dev); ghost_buf = torch.zeros(ng, dtype=torch.float64, device=dev)
raw = (ctypes.c_char * 128)(); L.comm_unique_id(ctypes.cast(raw, ctypes.c_void_p))
comm = ctypes.c_void_p(); L.comm_init_rank(0, 0, 1, ctypes.cast(raw, ctypes.c_void_p), ctypes.byref(comm))
step = ctypes.c_void_p()
L.dist_spmv_create(comm, _capi.F64, rows, loc.handle, rows_with.numel(), p(rows_with), p(cp), p(rc), p(rv),
                   ng, p(send_idx), p(send_buf), cnts, ng, p(ghost_buf), cnts, ctypes.byref(step))
bench(step, "rccl (send/recv to self), issued directly")
assert torch.equal(y, yd), "RCCL step and IPC step differ"
out["rccl_equals_ipc"] = True
L.dist_spmv_destroy(step); L.comm_destroy(comm)

# ---- the parts alone
rem = ops.RowSubsetCSR(rows_with, cp, rc, rv)
for label, fn in (("local part alone", lambda: loc.apply(x, y)),
                  ("remote part alone", lambda: rem.apply(ghost_buf, y, 1.0, True)),
                  ("local + remote, no exchange", lambda: (loc.apply(x, y), rem.apply(ghost_buf, y, 1.0, True)))):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(400):
        fn()
    e1.record(); torch.cuda.synchronize()
    out[label] = {"device_us_per_step": round(e0.elapsed_time(e1) * 1e3 / 400, 2)}
    print(label, out[label], file=sys.stderr, flush=True)
print(json.dumps(out))
