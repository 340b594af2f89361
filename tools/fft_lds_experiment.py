"""How sensitive the FFT row kernel is to workgroups per CU: the same rows with extra (unused) dynamic LDS per workgroup
(VEXHIP_FFT_EXTRA_LDS).  Measured: 1024-point fp64 rows (16 KiB of LDS per 128-lane workgroup) +1 % at 20 KiB, +8 % at 24,
+21 % at 32 KiB; 512-point rows show no difference between 8 and 16 KiB -- occupancy saturates near 16 KiB per workgroup, so
exchanging real and imaginary parts separately (half the LDS, twice the barriers) would buy little below 2048 points."""
import os, sys, json
sys.path.insert(0, '/root/repo')
import torch
from vexcl_amd import ops
def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
dev = torch.device("cuda:0")
n = 1 << 26
x = torch.randn(n, dtype=torch.float64, device=dev).to(torch.complex128); y = torch.empty_like(x)
for rowlen in (512, 1024):
    for extra in (0, 4096, 8192, 16384, 32768):
        os.environ["VEXHIP_FFT_EXTRA_LDS"] = str(extra)
        f = ops.FFT([n // rowlen, rowlen], [2, 0])
        print(rowlen, "extra", extra, round(timed(lambda: f(x, out=y, scaled=False), 10), 4), flush=True)
