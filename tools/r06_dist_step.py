"""Round 6: the product step of ONE rank at the 8-GPU geometry (rank 3 of 8 of the 512^3 Poisson problem: 16 777 216 rows, a ghost
plane per z-neighbour) on the one GPU of the gpurun box, the rank exchanging with ITSELF (tools/r04_dist_step.py explains the stand-in).
  halo push -- round 5: one launch whose first workgroups copy the boundary planes into the neighbours' uncached windows;
  halo PULL -- round 6 (vexhip_dist_spmv_create_halo_pull, order flags): nothing is copied; the planes next to a ghost plane read the
               neighbour's boundary plane of x IN PLACE behind an "x is final" flag raised by the first workgroup of the owner's
               launch; the kernel behind the launch raises `consumed` and waits for the neighbours' -- what vex::SpMat runs on a
               multi-GPU vex::Context (vexcl/spmat.hpp);
  the same with the streams ordered by events instead of flags (no signal kernel), and the local part alone.
All must give the bits of the one-device product.  Sweeps: planes per edge chunk, the acquire behind the flag.
Output: JSON on stdout (profiles/r06_dist_step.json)."""
import ctypes, json, os, subprocess, sys, time
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
DT = torch.float32 if os.environ.get("DIST_DTYPE", "f64") == "f32" else torch.float64      # DIST_DTYPE=f32: the float step (plane32.hip; pull only)
val = val.to(DT)
if os.environ.get("DIST_VARIABLE"):       # a value per entry (the variable-coefficient problem): no geometry plan -- SELL-512 with diagonal codes, the pair product's HALO role
    val = val * (1.0 + 1e-3 * ops.fill_hash(torch.empty(val.numel(), dtype=torch.float64, device=dev), 7).to(DT))
p = lambda t: ctypes.c_void_p(t.data_ptr())
# the strip stored WITH its ghost planes, built by the library (what vexcl/spmat.hpp calls)
nnz = int(col.numel())
ptr_ext = torch.empty(rows + 2 * P + 1, dtype=torch.int32, device=dev); col_ext = torch.empty(nnz, dtype=torch.int32, device=dev)
bad = ctypes.c_int64(-1)
L.csr_extend_halo_i32(0, None, rows, nnz, p(ptr), p(col), r0, P, P, p(ptr_ext), p(col_ext), ctypes.byref(bad))
assert bad.value == 0, bad.value
torch.cuda.synchronize()
ext = ops.SpMat(ptr_ext, col_ext, val, n_cols=rows + 2 * P)
assert ext.plane or ext.grid or ext.storage in ("sell8", "sell8v"), "the stored strip did not get a plane or grid plan nor diagonal codes"
loc_only = None
x = ops.fill_hash(torch.empty(rows, dtype=torch.float64, device=dev), 42).to(DT); y = torch.empty_like(x)
s = torch.cuda.Stream(); sp = ctypes.c_void_p(s.cuda_stream)
out = {"grid": n, "strip_rows": rows, "stored_strip": {"rows": rows + 2 * P, "storage": ext.storage, "product": ext.product, "plane_plan": ext.plane},
       "env": {k: v for k, v in os.environ.items() if k.startswith("VEXHIP_HALO")}}
# the bits of the ONE-device product: the same stored strip through the CSR kernel on x with its ghost planes attached
x_ext = torch.cat([x[rows - P:], x, x[:P]]).contiguous()
y_ext = torch.empty(rows + 2 * P, dtype=DT, device=dev)
ref = ops.SpMat(ptr_ext, col_ext, val, n_cols=rows + 2 * P, fmt="csr")
ref.apply(x_ext, y_ext); torch.cuda.synchronize()
y_one = y_ext[P:P + rows].clone()
del ref


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
    return best


def status(step):
    to = ctypes.c_int(); tr = ctypes.c_int(); L.dist_spmv_status(step, ctypes.byref(to), ctypes.byref(tr), None)
    return to.value, tr.value


which = os.environ.get("DIST_ONLY", "push,pull,events,parts").split(",")
if "push" in which and ext.plane and DT == torch.float64:
    win = ctypes.c_void_p(); L.ipc_window_create(0, 0, 1, 2 * P * 8, ctypes.byref(win))
    step = ctypes.c_void_p()
    L.dist_spmv_create_halo(win, ext.handle, rows, P, 0, 0, ctypes.byref(step))
    y.fill_(float("nan"))
    bench(lambda: L.dist_spmv_apply(step, sp, 1.0, 0, p(x), p(y)), "halo push (round 5): product launch + signal launch")
    out["push timed_out, transport"] = list(status(step))
    out["push_equals_one_device_csr_order"] = bool(torch.equal(y, y_one))
    L.dist_spmv_destroy(step); L.ipc_window_destroy(win)

xb, xa = ctypes.c_void_p(x.data_ptr() + (rows - P) * x.element_size()), ctypes.c_void_p(x.data_ptr())      # own last plane below, own first plane above
if "pull" in which:
    win = ctypes.c_void_p(); L.ipc_window_create(0, 0, 1, 0, ctypes.byref(win))
    step = ctypes.c_void_p()
    L.dist_spmv_create_halo_pull(win, ext.handle, rows, P, 0, 0, 1, ctypes.byref(step))
    y.fill_(float("nan"))
    bench(lambda: L.dist_spmv_apply_pull(step, sp, 1.0, 0, p(x), p(y), xb, xa), "halo PULL, flags (round 6): product launch + signal launch")
    out["pull timed_out, transport"] = list(status(step))
    out["pull_equals_one_device_csr_order"] = bool(torch.equal(y, y_one))
    y2 = y_one.clone(); torch.cuda.synchronize()
    with torch.cuda.stream(s):
        L.dist_spmv_apply_pull(step, sp, 0.5, 1, p(x), p(y2), xb, xa)
    s.synchronize()
    out["pull_append_equals_reference"] = bool(torch.equal(y2, y_one + 0.5 * y_one))
    L.dist_spmv_set_graph(step, 1)
    bench(lambda: L.dist_spmv_apply_pull(step, sp, 1.0, 0, p(x), p(y), xb, xa), "halo PULL, flags, hipGraph replay")
    out["pull_graph_equals_one_device"] = bool(torch.equal(y, y_one))
    L.dist_spmv_set_graph(step, 0)
    L.dist_spmv_destroy(step); L.ipc_window_destroy(win)
if "events" in which:
    step = ctypes.c_void_p()
    L.dist_spmv_create_halo_pull(None, ext.handle, rows, P, 0, 0, 2, ctypes.byref(step))
    y.fill_(float("nan"))
    bench(lambda: L.dist_spmv_apply_pull(step, sp, 1.0, 0, p(x), p(y), xb, xa), "halo PULL, no flags (the host's events order the streams): one launch")
    out["pull_events_equals_one_device_csr_order"] = bool(torch.equal(y, y_one))
    # the same launch on the vectors the ORDINARY product of the stored strip uses below (x_ext with its ghost planes in place, y_ext):
    # same kernel body, same memory -- what the one-launch form of the kernel itself costs
    xm = x_ext[P:P + rows]; ym = y_ext[P:P + rows]
    bench(lambda: L.dist_spmv_apply_pull(step, sp, 1.0, 0, p(xm), p(ym), p(x_ext), ctypes.c_void_p(x_ext.data_ptr() + (P + rows) * x_ext.element_size())),
          "halo PULL, no flags, on the vectors of the ordinary product (x with its ghost planes in place)")
    out["pull_events_in_place_equals_one_device"] = bool(torch.equal(ym, y_one))
    L.dist_spmv_destroy(step)
if "parts" in which:
    lp, lc, lv = None, None, None
    is_loc = (col >= r0) & (col < r1)
    row_of = torch.repeat_interleave(torch.arange(rows, device=dev), (ptr[1:] - ptr[:-1]).to(torch.int64))
    cnt = torch.bincount(row_of[is_loc], minlength=rows)
    pp = torch.zeros(rows + 1, dtype=torch.int64, device=dev); pp[1:] = torch.cumsum(cnt, 0)
    loc = ops.SpMat(pp.to(torch.int32), (col[is_loc] - r0).to(torch.int32).contiguous(), val[is_loc].contiguous(), n_cols=rows)
    del is_loc, row_of, cnt, pp
    with torch.cuda.stream(s):
        bench(lambda: loc.apply(x, y), "local part alone (no ghost entries)")
        bench(lambda: ext.apply(x_ext, y_ext), "stored strip (66 planes) through the ordinary plane product, ghost planes part of x")
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(os.environ.get("DIST_OUT", "gpurun_out/r06_dist_step.json"), "w"), indent=1)
