"""Host and device cost of ONE rank's product step at the 8-GPU geometry, on the one GPU of the gpurun box.

The strip is rank 3 of 8 of the 512^3 Poisson problem (16 777 216 rows, one ghost plane from each z-neighbour: 2 x 262 144
columns).  With one GPU there is nobody to exchange with, so the rank exchanges WITH ITSELF over RCCL (ncclSend/ncclRecv to
self, VEXHIP_RCCL_SELF=1): the packed boundary values of its own x land in its ghost buffer.  The numbers in y are therefore
not the 8-GPU product's -- this tool measures what a step COSTS: host time per vexhip_dist_spmv_apply call (pack kernel,
grouped send/recv on the second stream, local part, event wait, remote part) with and without hipGraph replay, and the
device time per step.  Output: gpurun_out/r02_dist_step.json"""
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
# stand-in: this rank's own values travel to itself.  Default: the two planes next to its boundaries (two runs for the ONE peer
# it has here: packed by the gather kernel); "direct" on the command line: one run of consecutive elements -- what each of the two
# neighbours of a plane partition gets -- which vexhip_dist_spmv sends straight out of x, without a pack kernel
direct = "direct" in sys.argv[1:]
send_idx = (torch.arange(ghosts.numel(), device=dev) if direct else ghosts % rows).to(torch.int32).contiguous()
send_buf = torch.empty(ng, dtype=torch.float64, device=dev); ghost_buf = torch.zeros(ng, dtype=torch.float64, device=dev)
del row_of, is_loc, ptr, col, val

raw = (ctypes.c_char * 128)(); L.comm_unique_id(ctypes.cast(raw, ctypes.c_void_p))
comm = ctypes.c_void_p(); L.comm_init_rank(0, 0, 1, ctypes.cast(raw, ctypes.c_void_p), ctypes.byref(comm))
cnts = (ctypes.c_int64 * 1)(ng)
p = lambda t: ctypes.c_void_p(t.data_ptr())
step = ctypes.c_void_p()
L.dist_spmv_create(comm, _capi.F64, rows, loc.handle, rows_with.numel(), p(rows_with), p(cp), p(rc), p(rv),
                   ng, p(send_idx), p(send_buf), cnts, ng, p(ghost_buf), cnts, ctypes.byref(step))
x = ops.fill_hash(torch.empty(rows, dtype=torch.float64, device=dev), 42); y = torch.empty_like(x)
s = torch.cuda.Stream(); sp = ctypes.c_void_p(s.cuda_stream)
out = {"send": "one consecutive run, sent out of x" if direct else "two runs, packed", "strip_rows": rows, "local_storage": loc.storage, "ghosts": ng, "remote_rows": int(rows_with.numel()),
       "exchange_bytes_each_way": ng * 8}


def bench(label, reps=300):
    for _ in range(5):
        L.dist_spmv_apply(step, sp, 1.0, 0, p(x), p(y))
    s.synchronize()
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
    out[label] = {"host_us_per_step": round(host * 1e6, 2), "device_us_per_step": round(e0.elapsed_time(e1) * 1e3 / reps, 2)}
    print(label, out[label], flush=True)


bench("direct issue from C++")
out["hipGraph replay"] = "not used for steps with an exchange: capturing ncclSend/ncclRecv crashed in RCCL 2.26.6 (this tool, first version)"
# the local part alone, for comparison
for _ in range(5):
    loc.apply(x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(300):
    loc.apply(x, y)
e1.record(); torch.cuda.synchronize()
out["local part alone"] = {"device_us_per_step": round(e0.elapsed_time(e1) * 1e3 / 300, 2)}
print("local part alone", out["local part alone"], flush=True)
rem = ops.RowSubsetCSR(rows_with, cp, rc, rv)
for label, fn in (("remote part alone", lambda: rem.apply(ghost_buf, y, 1.0, True)),
                  ("pack alone", lambda: ops.gather(send_idx, x, send_buf)),
                  ("local + remote, no exchange", lambda: (loc.apply(x, y), rem.apply(ghost_buf, y, 1.0, True)))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(300):
        fn()
    e1.record(); torch.cuda.synchronize()
    out[label] = {"device_us_per_step": round(e0.elapsed_time(e1) * 1e3 / 300, 2)}
    print(label, out[label], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r02_dist_step%s.json" % ("_direct" if direct else ""), "w"), indent=1)
