"""Tile size of the strided FFT passes (VEXHIP_FFT_STRIDED_ELEMS): elements per workgroup tile = lines x length; wider tiles
mean longer contiguous runs, smaller ones more workgroups per CU.  Measured: 2048 elements is best for fp64 (the LDS limit)
and for fp32 (5-7 % better than the 4096 the LDS would allow); 1024 and below lose to shorter runs."""
import os, sys
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
for dtype, nm in ((torch.complex128, "fp64"), (torch.complex64, "fp32")):
    for label, sizes in (("2^24", [1 << 24]), ("4096^2", [4096, 4096]), ("256^3", [256, 256, 256]), ("2048^2", [2048, 2048])):
        total = 1
        for s in sizes: total *= s
        x = torch.randn(total, dtype=torch.float64 if nm == "fp64" else torch.float32, device=dev).to(dtype); y = torch.empty_like(x)
        for elems in (4096, 2048, 1024, 512):
            os.environ["VEXHIP_FFT_STRIDED_ELEMS"] = str(elems)
            f = ops.FFT(sizes, [0] * len(sizes), dtype=dtype)
            print(nm, label, "strided tile", elems, round(timed(lambda: f(x, out=y, scaled=False), 10), 4), flush=True)
