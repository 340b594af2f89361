#!/usr/bin/env python
"""vex::FFT throughput on one MI355X (tools, not part of bench.py's headline): complex-to-complex transforms
through the C ABI, HIP events around R repetitions, inputs resident.  Algorithmic bytes = one read + one write of
the data (2 * 16 B per fp64 element); the plan's pass count says how many times the data actually moves.
rocFFT through torch.fft is timed beside it as a cross-check of what the box can do (not a dependency)."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def tune():
    """Row-kernel knobs (VEXHIP_FFT_ROW_ELEMS: elements per tile of contiguous lines; VEXHIP_FFT_LANES_DIV: lanes per
    workgroup = butterflies of the widest stage / div), read by the plan at creation."""
    dev = torch.device("cuda:0")
    out = {}
    for dtype, name in ((torch.complex128, "fp64"), (torch.complex64, "fp32")):
        for n in (256, 1024, 2048, 4096 if name == "fp32" else 1000):
            sizes = [(1 << 26) // n, n]
            total = sizes[0] * sizes[1]
            x = torch.randn(total, dtype=torch.float64 if name == "fp64" else torch.float32, device=dev).to(dtype)
            y = torch.empty_like(x)
            for elems in (512, 1024, 2048, 4096):
                for div in (1, 2, 4):
                    os.environ["VEXHIP_FFT_ROW_ELEMS"] = str(elems)
                    os.environ["VEXHIP_FFT_LANES_DIV"] = str(div)
                    f = ops.FFT(sizes, [2, 0], dtype=dtype)
                    ms = timed(lambda: f(x, out=y, scaled=False), 10)
                    out["%s n=%d elems=%d div=%d" % (name, n, elems, div)] = round(ms, 4)
                    del f
            del x, y
            torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "tune":
        return tune()
    dev = torch.device("cuda:0")
    cases = [
        ("1-D n=1024, batch 65536", [65536, 1024], [2, 0]),
        ("1-D n=2048, batch 32768", [32768, 2048], [2, 0]),
        ("1-D n=4096, batch 16384", [16384, 4096], [2, 0]),
        ("1-D n=1000, batch 65536", [65536, 1000], [2, 0]),
        ("1-D n=2^24", [1 << 24], [0]),
        ("2-D 4096 x 4096", [4096, 4096], [0, 0]),
        ("2-D 2048 x 2048", [2048, 2048], [0, 0]),
        ("3-D 256^3", [256, 256, 256], [0, 0, 0]),
        ("1-D n=1009 (Bluestein), batch 16384", [16384, 1009], [2, 0]),
    ]
    out = {}
    for dtype, name, eb in ((torch.complex128, "fp64", 16), (torch.complex64, "fp32", 8)):
        for label, sizes, dirs in cases:
            total = 1
            for s in sizes:
                total *= s
            x = torch.randn(total, dtype=torch.float64 if eb == 16 else torch.float32, device=dev).to(dtype)
            y = torch.empty_like(x)
            f = ops.FFT(sizes, dirs, dtype=dtype)
            ms = timed(lambda: f(x, out=y, scaled=False), 10)
            xs = x.reshape(sizes)
            axes = tuple(i for i, d in enumerate(dirs) if d != 2)
            ms_ref = timed(lambda: torch.fft.fftn(xs, dim=axes), 10)
            out["%s %s" % (name, label)] = {"ms": round(ms, 4), "alg_gbps": round(2 * eb * total / ms / 1e6, 1),
                                            "steps(rows,transposes,other)": f.steps(), "rocfft_ms": round(ms_ref, 4)}
            del f, x, y, xs
            torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
