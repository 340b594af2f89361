#!/usr/bin/env python
"""Interleaved A/B timing of the SpMV kernel variants on the Poisson matrix
(default 512^3); writes gpurun_out/spmv_sweep.json.  Variant word: bit0 =
nontemporal matrix streams, bit1 = XCD-contiguous block order, bits 2-3 =
log2(rows per lane) (HELL only); "tiled" = L2-tiled traversal order
(vexhip_hell_order_i32).  Every variant is timed R times, round-robin, so that
clock / thermal drift hits all of them alike; median and min are reported."""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", default="csr0,hell5,sell32,sell_plain,sell")
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
    A_ell.hell = ops.HybridELL(ptr, col, val, order_mode=2)
    H_slab = None; H_rr = A_ell.hell
    print("tiled order grid:", A_ell.hell.order_grid, flush=True)

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters

    def runner(name):
        if name == "copy":
            nb = alg // 2 // 8
            a = torch.empty(nb, dtype=torch.float64, device=dev).normal_()
            b = torch.empty_like(a)
            return lambda: b.copy_(a)
        if name.startswith("csr"):
            v = int(name[3:])

            def f():
                L.spmv_csr_set_variant(v)
                A_csr.apply(x, y)
            return f
        if name == "tiled":
            return lambda: A_ell.hell.mul(x, y, tiled=True)
        if name == "tiled_slab":
            return lambda: H_slab.mul(x, y, tiled=True)
        if name == "tiled_rr":
            return lambda: H_rr.mul(x, y, tiled=True)
        if name == "sell":
            S = ops.SlicedELL(ptr, col, val)                      # default traversal order
            return lambda: S.mul(x, y)
        if name == "sell32":
            S = ops.SlicedELL(ptr, col, val, codes=False)
            return lambda: S.mul(x, y)
        if name == "sell_plain":
            S = ops.SlicedELL(ptr, col, val, tiled=False)
            return lambda: S.mul(x, y)
        if name.startswith("sell_"):            # sell_rr, sell_slab, sell_chunk<k>
            mode = {"rr": 2, "slab": 1}.get(name[5:]) or 100 + int(name[10:])
            S = ops.SlicedELL(ptr, col, val, order_mode=mode)
            return lambda: S.mul(x, y)
        if name.startswith("chunk"):
            H = ops.HybridELL(ptr, col, val, order_mode=100 + int(name[5:]))
            H.ell_col, H.ell_val = A_ell.hell.ell_col, A_ell.hell.ell_val      # share the matrix arrays
            return lambda: H.mul(x, y, tiled=True)
        v = int(name[4:])

        def g():
            L.spmv_hell_set_variant(v)
            A_ell.hell.mul(x, y, tiled=False)
        return g

    names = ["copy"] + args.variants.split(",")
    fns = {k: runner(k) for k in names}
    A_csr.apply(x, y)
    ref = y.clone()
    ok = {}
    for k in names[1:]:
        y.zero_()
        fns[k]()
        ok[k] = bool(torch.equal(ref, y))
    times = {k: [] for k in names}
    for _ in range(args.rounds):
        for k in names:
            times[k].append(timeit(fns[k]))
    L.spmv_csr_set_variant(-1)
    L.spmv_hell_set_variant(-1)
    res = []
    for k in names:
        med, mn = statistics.median(times[k]), min(times[k])
        b = alg if k != "copy" else (alg // 2 // 8) * 16
        r = dict(kernel=k, median_ms=round(med, 4), min_ms=round(mn, 4), gbps_median=round(b / med / 1e6, 1),
                 frac_of_8TBps=round(b / med / 1e6 / 8000, 4), bit_equal_to_csr=ok.get(k))
        res.append(r)
        print(json.dumps(r), flush=True)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(dict(grid=n, rows=N, nnz=nnz, algorithmic_bytes=alg, iters=args.iters, rounds=args.rounds, results=res),
              open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
