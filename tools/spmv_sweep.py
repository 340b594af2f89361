#!/usr/bin/env python
"""Times every SpMV kernel variant on the Poisson matrix (default 512^3) and
writes gpurun_out/spmv_sweep.json.  Variant word: bit0 = nontemporal matrix
streams, bit1 = XCD-contiguous block order, bits 2-3 = log2(rows per lane), HELL only."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--out", default="gpurun_out/spmv_sweep.json")
    args = ap.parse_args()
    import torch
    from vexcl_amd import lib, ops
    L = lib()
    dev = torch.device("cuda:0")
    n = args.grid
    N = n ** 3
    nnz = L.poisson3d_nnz(n)
    alg = nnz * 12 + (N + 1) * 4 + N * 16
    ptr, col, val = ops.poisson3d(n, dev)
    x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42)
    y = torch.zeros(N, dtype=torch.float64, device=dev)
    A_csr = ops.SpMat(ptr, col, val, fmt="csr")
    A_ell = ops.SpMat(ptr, col, val, fmt="hell")

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters

    res = []
    # calibration: plain copy of the same number of bytes / 2 (read + write)
    nb = alg // 2 // 8
    a = torch.empty(nb, dtype=torch.float64, device=dev).normal_()
    b = torch.empty_like(a)
    ms = timeit(lambda: b.copy_(a))
    res.append(dict(kernel="torch_copy", variant=-1, ms=ms, gbps=2 * nb * 8 / ms / 1e6))
    del a, b
    ref = None
    for v in range(4):
        L.spmv_csr_set_variant(v)
        ms = timeit(lambda: A_csr.apply(x, y))
        if ref is None:
            ref = y.clone()
        ok = bool(torch.equal(ref, y))
        res.append(dict(kernel="csr_stream", variant=v, ms=ms, gbps=alg / ms / 1e6, frac=alg / ms / 1e6 / 8000, ok=ok))
    L.spmv_csr_set_variant(-1)
    for v in range(12):
        L.spmv_hell_set_variant(v)
        ms = timeit(lambda: A_ell.apply(x, y))
        ok = bool(torch.equal(ref, y))
        res.append(dict(kernel="hell", variant=v, ms=ms, gbps=alg / ms / 1e6, frac=alg / ms / 1e6 / 8000, ok=ok))
    L.spmv_hell_set_variant(-1)
    for r in res:
        print(json.dumps(r))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(dict(grid=n, rows=N, nnz=nnz, algorithmic_bytes=alg, results=res), open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
