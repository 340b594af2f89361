#!/usr/bin/env python
"""Fixed workload for the FFT PMC passes (tools/fft_pmc.sh): a calibration stream of known size (2 GiB through the
reduction kernel, 16-byte loads) and then three launches each of: 65 536 fp64 rows of 1024 (one pass), 2^24 fp64 points
(three passes)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops

dev = torch.device("cuda:0")
cal = torch.empty(1 << 28, dtype=torch.float64, device=dev).normal_()
r = ops.Reductor("SUM")
for _ in range(3):
    r.device_result(cal)
n = 1 << 26
x = torch.randn(n, dtype=torch.float64, device=dev).to(torch.complex128)
y = torch.empty_like(x)
f = ops.FFT([65536, 1024], [ops.NONE, ops.FORWARD])
for _ in range(3):
    f(x, out=y, scaled=False)
torch.cuda.synchronize()
g = ops.FFT([1 << 24], [ops.FORWARD])
for _ in range(3):
    g(x[: 1 << 24], out=y[: 1 << 24], scaled=False)
torch.cuda.synchronize()
print("done")
