"""Debug (round 6): where do four ranks sharing ONE GPU stall in DistSpMat's set-up?  Every step synchronised and timed, per rank."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import faulthandler; faulthandler.dump_traceback_later(60, exit=True)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from vexcl_amd import ops
from vexcl_amd.distributed import partition
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
n = int(os.environ.get("GRID", "512")); N = n ** 3
part = partition(N, world); r0, r1 = part[rank], part[rank + 1]
t0 = time.perf_counter()
def mark(what):
    torch.cuda.synchronize(); print("[rank %d %6.2f s] %s" % (rank, time.perf_counter() - t0, what), flush=True)
mark("start, free memory %.1f GB" % (torch.cuda.mem_get_info()[0] / 1e9))
ptr, col, val = ops.poisson3d(n, dev, rows=(r0, r1)); mark("strip generated: %d entries" % col.numel())
if os.environ.get("PRELUDE", "1") == "1":          # what bench.py allocates next
    x = ops.fill_hash(ops.device_vector(r1 - r0, torch.float64, dev), (42 + r0 * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF); mark("x placed by vexhip_malloc and filled")
    y = ops.device_vector(r1 - r0, torch.float64, dev, zero=True); mark("y placed and zeroed")
is_loc = (col >= r0) & (col < r1); mark("mask")
rem = col[~is_loc]; mark("remote columns: %d" % rem.numel())
r64 = rem.to(torch.int64); mark("to int64")
g = torch.unique(r64); mark("unique: %d" % g.numel())
dist.barrier(); mark("barrier")
del rem, r64, g, is_loc
import faulthandler as fh
fh.cancel_dump_traceback_later(); fh.dump_traceback_later(50, exit=True)
c0, c1 = r0, r1
nosync = os.environ.get("NOSYNC", "0") == "1"
def m2(what):
    if nosync: print("[rank %d %6.2f s] (no sync) %s" % (rank, time.perf_counter() - t0, what), flush=True)
    else: mark(what)
is_loc = (col >= c0) & (col < c1); m2("B mask")
rem_mask = ~is_loc; m2("B ~mask")
sel = col[rem_mask]; m2("B col[rem_mask]")
ghosts = torch.unique(sel.to(torch.int64)); m2("B unique")
row_of = torch.repeat_interleave(torch.arange(r1 - r0, device=dev), (ptr[1:] - ptr[:-1]).to(torch.int64)); m2("B repeat_interleave")
cnt = torch.bincount(row_of[is_loc], minlength=r1 - r0); m2("B bincount")
mark("B done")
