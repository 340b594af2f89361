#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric: fp64 CSR SpMV GFLOP/s + achieved HBM GB/s
on the 3-D Poisson matrix (examples/benchmark.cpp:353-477), 512^3 grid.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one product y = A*x through vex::SpMat's path (libvexhip.so).
The matrix is built directly in HBM; inputs are resident before
the timed region.  N > 1: the SAME 512^3 problem row-partitioned over the ranks
(strong scaling), ghost planes exchanged over RCCL (vexcl_amd/distributed.py).

Algorithmic work per product (independent of the internal
storage format):  bytes = nnz*12 + (N+1)*4 + N*8 + N*8,  flops = 2*nnz.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s


def algorithmic_bytes(n_rows, nnz):
    return nnz * 12 + (n_rows + 1) * 4 + n_rows * 8 + n_rows * 8


def cpu_baseline(grid, seconds):
    """The reference's CPU-device path restated (oracle/vex_oracle.c,
    vxo_spmv_csr_f64_i32_omp: 8 x threads contiguous row chunks), timed on the
    host cores of this box on a bounded sample of the same workload."""
    import numpy as np
    import oracle
    ptr, col, val = oracle.poisson3d(grid)
    N, nnz = grid ** 3, len(col)
    x = np.full(N, 1e-2)
    y = np.zeros(N)
    oracle.spmv_csr(ptr, col, val, x, y, omp=True)          # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        oracle.spmv_csr(ptr, col, val, x, y, omp=True)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or reps >= 1000:
            break
    per = dt / reps
    return {"value": round(2.0 * nnz / per / 1e9, 3), "unit": "GFLOP/s", "cores": oracle.num_threads(),
            "kind": "port", "hbm_equiv_gbps": round(algorithmic_bytes(N, nnz) / per / 1e9, 2),
            "sample": "%d products of the %d^3 Poisson matrix (N=%d, nnz=%d), OpenMP chunked csr_spmv restatement"
                      % (reps, grid, N, nnz)}


def read_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes
    (profiles/*pmc_summary.json written by tools/profile.sh + tools/pmc_summary.py:
    FETCH_SIZE calibrated x2 on a stream of known size, + WRITE_SIZE), or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json"))):
        try:
            d = json.load(open(f))
            if kernel in d and "hbm_bytes_per_launch" in d[kernel]:
                best = int(d[kernel]["hbm_bytes_per_launch"])
        except Exception:
            pass
    return best


ELEMENTWISE_SRC = r'''
// the reference's fused kernel for `a = b * c + sin(d)` (SURVEY appendix A.1 shape), two elements per trip
extern "C" __global__ void vexcl_vector_kernel(ulong n, double *prm_1, const double *prm_2, const double *prm_3, const double *prm_4) {
  const ulong grid_size = blockDim.x * (ulong)gridDim.x;
  for (ulong vex_i = blockDim.x * (ulong)blockIdx.x + threadIdx.x; vex_i < n; vex_i += 2 * grid_size) {
    const bool vex_two = vex_i + grid_size < n;
    const ulong i1 = vex_two ? vex_i + grid_size : vex_i;
    const double r0 = ( ( prm_2[vex_i] * prm_3[vex_i] ) + sin( prm_4[vex_i] ) );
    const double r1 = ( ( prm_2[i1] * prm_3[i1] ) + sin( prm_4[i1] ) );
    prm_1[vex_i] = r0;
    if (vex_two) prm_1[i1] = r1;
  }
}
'''


def secondary_rows(torch, L, ops, dev, local_rank):
    """BASELINE.json's secondary metrics: elementwise, reduce, scan, sort and the
    multi-right-hand-side product, each timed with HIP events after a warm-up, inputs resident.
    Reported next to the headline value; never part of it."""
    import ctypes
    rows = {}

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    n = 10 ** 8
    b, c, d = (ops.fill_hash(torch.empty(n, dtype=torch.float64, device=dev), s) for s in (1, 2, 3))
    a = torch.empty_like(b)
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    L.module_compile(local_rank, ELEMENTWISE_SRC.encode(), b"", ctypes.byref(mod))
    L.module_get_function(local_rank, mod, b"vexcl_vector_kernel", ctypes.byref(fn))
    args = [ctypes.c_uint64(n)] + [ctypes.c_void_p(t.data_ptr()) for t in (a, b, c, d)]
    arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(v), ctypes.c_void_p) for v in args])
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    blocks = (n + 511) // 512                      # one trip per lane (vexcl/backend.hpp config_streaming)
    ms = timed(lambda: L.launch(local_rank, fn, blocks, 1, 1, 256, 1, 1, 0, stream, arr), 20)
    rows["elementwise a=b*c+sin(d) f64 n=1e8"] = {"ms": round(ms, 4), "gbps": round(32.0 * n / ms / 1e6, 1)}
    red = ops.Reductor("SUM")
    ms = timed(lambda: red.dot(b, c), 20)
    rows["reduce sum(a*b) f64 n=1e8 (incl. host readback)"] = {"ms": round(ms, 4), "gbps": round(16.0 * n / ms / 1e6, 1)}
    L.module_unload(local_rank, mod)
    del a, b, c, d

    n = 10 ** 9
    k = ops.fill_hash(torch.empty(n, dtype=torch.int32, device=dev), 42)
    out = torch.empty_like(k)
    ms = timed(lambda: ops.inclusive_scan(k, out, unsigned=True), 10)
    rows["inclusive_scan u32 n=1e9"] = {"ms": round(ms, 4), "gbps": round(8.0 * n / ms / 1e6, 1)}
    del out
    # sort through the raw C-ABI call on pre-allocated buffers (in place: re-filled before every repetition)
    ktmp = torch.empty_like(k)
    tmp = torch.empty(L.sort_tmp_bytes(3, n), dtype=torch.uint8, device=dev)
    best = None
    for _ in range(3):
        ops.fill_hash(k, 42); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.sort(local_rank, stream, 3, 0, ctypes.c_void_p(k.data_ptr()), ctypes.c_void_p(ktmp.data_ptr()), 0, None, None, n,
               ctypes.c_void_p(tmp.data_ptr()))
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1)
        best = t if best is None else min(best, t)
    rows["sort u32 keys n=1e9 (stable LSD radix)"] = {"ms": round(best, 3), "gkeys_per_s": round(n / best / 1e6, 1)}
    del k, ktmp, tmp
    torch.cuda.empty_cache()

    # vex::FFT (outside BASELINE.json's configs): complex fp64, algorithmic bytes = one read + one write
    for label, sizes, dirs in (("fft c2c f64, 65536 rows x 1024", [65536, 1024], [ops.NONE, ops.FORWARD]),
                               ("fft c2c f64, 2^24 points", [1 << 24], [ops.FORWARD]),
                               ("fft c2c f64, 4096 x 4096", [4096, 4096], [ops.FORWARD, ops.FORWARD])):
        total = 1
        for s in sizes:
            total *= s
        z = torch.view_as_complex(ops.fill_hash(torch.empty(2 * total, dtype=torch.float64, device=dev), 7).view(total, 2))
        w = torch.empty_like(z)
        f = ops.FFT(sizes, dirs)
        ms = timed(lambda: f(z, out=w, scaled=False), 10)
        rows[label] = {"ms": round(ms, 4), "gbps": round(32.0 * total / ms / 1e6, 1), "passes": f.steps()[0]}
        del f, z, w
        torch.cuda.empty_cache()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--grid", type=int, default=512, help="Poisson grid edge (512 = BASELINE config)")
    ap.add_argument("--format", default="auto", choices=["auto", "sell", "sell8", "sell32", "hell", "csr"],
                    help="auto/sell: best storage the matrix allows (value codes, diagonal codes, 32-bit columns); "
                         "sell8: diagonal codes, values as they are; sell32: 32-bit columns; hell: reference layout; csr")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-grid", type=int, default=256)
    ap.add_argument("--dist", action="store_true", help="use the partitioned SpMat even on one GPU (debug)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (gloo: debug)")
    ap.add_argument("--one-device", action="store_true", help="all ranks on cuda:0 (debug: exercises the N>1 path on a 1-GPU box; needs --backend gloo)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary rows (elementwise, reduce, scan, sort, multi-rhs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from vexcl_amd import lib, ops
    from vexcl_amd.distributed import DistReductor, DistSpMat, partition

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    L = lib()
    n = args.grid
    N = n ** 3
    nnz_total = L.poisson3d_nnz(n)
    part = partition(N, world)
    r0, r1 = part[rank], part[rank + 1]

    # ---- inputs resident in HBM before the timed region
    ptr, col, val = ops.poisson3d(n, dev, rows=(r0, r1))
    # x is a function of the GLOBAL index, so the job computes the same product for every N
    x = ops.fill_hash(torch.empty(r1 - r0, dtype=torch.float64, device=dev),
                      (42 + r0 * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
    y = torch.zeros(r1 - r0, dtype=torch.float64, device=dev)
    if world == 1 and not args.dist:
        A = ops.SpMat(ptr, col, val, fmt=args.format)
        fmt = A.fmt
        if fmt in ("hell", "sell"):
            del ptr, col, val                    # the product only needs the ELL arrays
            A.ptr = A.col = A.val = None
        step = lambda: A.apply(x, y, 1.0, False)
    else:
        A = DistSpMat(ptr, col, val, N, N, local_fmt=args.format)
        fmt = A.loc.fmt
        step = lambda: A.apply(x, y, 1.0, False)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()

    # HIP events on the stream the kernels are launched on (torch's current stream)
    import ctypes
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    L.event_create(local_rank, 1, ctypes.byref(e0))
    L.event_create(local_rank, 1, ctypes.byref(e1))

    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    L.event_record(local_rank, e0, stream)
    for _ in range(args.steps):
        step()
    L.event_record(local_rank, e1, stream)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    ms = ctypes.c_float()
    L.event_elapsed_ms(local_rank, e0, e1, ctypes.byref(ms))

    checksum = DistReductor("SUM_Kahan")(y)          # sum(y): the same for every N up to rounding
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    per_step = elapsed / args.steps
    kern_s = ms.value / 1e3 / args.steps         # average launch duration of the product on this rank

    if rank == 0:
        nnz_rank = L.poisson3d_strip_nnz(n, r0, r1)
        alg_total = algorithmic_bytes(N, nnz_total)
        alg_rank = algorithmic_bytes(r1 - r0, nnz_rank)
        gflops = 2.0 * nnz_total / per_step / 1e9
        gbps = alg_total / per_step / 1e9
        hell = (A.hell if world == 1 and not args.dist else A.loc.hell)
        if fmt == "sell" and getattr(hell, "values", None) is not None:
            fmt = "sell8v"                       # ... and 1-byte value codes (<= 255 distinct values: constant-coefficient stencil)
        elif fmt == "sell" and getattr(hell, "deltas", None) is not None:
            fmt = "sell8"                        # SELL-512 with 1-byte diagonal codes (banded matrix detected)
        kname = {"hell": "hell_kernel", "sell": "sell_kernel", "sell8": "sell8_kernel", "sell8v": "sell8v_kernel"}.get(fmt, "csr_stream_kernel")
        traffic = read_traffic(kname) if (world == 1 and n == 512) else None
        out = {
            "metric": "fp64 CSR SpMV GFLOP/s, 3D Poisson %d^3 (y = A*x, vex::SpMat path)" % n,
            "value": round(gflops, 2),
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(per_step * 1e3, 5),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "checksum_sum_y": checksum,
            "hbm_gbps": round(gbps, 1),
            "hbm_frac_of_peak": round(gbps / (HBM_PEAK_GBPS * world), 4),
            "config": {"workload": "configs[%d]: 7-point 3D Poisson %d^3, N=%d rows, nnz=%d, fp64 values, int32 indices"
                                   % (2 if world == 1 else 3, n, N, nnz_total),
                       "format": fmt, "rows_per_gpu": r1 - r0,
                       "parallelism": "row-partitioned x%d" % world},
            "roofline": {"bound": "hbm",
                         # achieved / frac follow the contract: ALGORITHMIC (CSR-format) bytes over the launch time.  A
                         # format that stores fewer bytes than CSR moves less than that -- `traffic` (PMC) says how much
                         # -- so frac can exceed 1; traffic_gbps / traffic_frac_of_peak are what crosses the HBM pins.
                         "achieved": round(alg_rank / kern_s / 1e9, 1),
                         "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s",
                         "frac": round(alg_rank / kern_s / 1e9 / HBM_PEAK_GBPS, 4),
                         "traffic": traffic,
                         # the stored matrix is smaller than the CSR-based algorithmic figure (1-byte diagonal codes):
                         # what actually crosses the HBM pins, for comparison with the 6.3 TB/s a stream reaches
                         "traffic_gbps": (round(traffic / kern_s / 1e9, 1) if traffic else None),
                         "traffic_frac_of_peak": (round(traffic / kern_s / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None),
                         "kernel": kname,
                         "algorithmic_bytes_per_launch": alg_rank,
                         "avg_launch_ms": round(kern_s * 1e3, 5)},
        }
        if world > 1:
            out["config"]["exchange_bytes_per_rank"] = A.exchange_bytes()
        if world == 1 and not args.dist and not args.no_secondary:
            try:
                sec = {}
                if fmt in ("sell", "sell8", "sell8v"):   # Y = A * X, X a multivector<double, 4>
                    xs = [ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 100 + k) for k in range(4)]
                    ys = [torch.empty(N, dtype=torch.float64, device=dev) for _ in range(4)]
                    A.apply_multi(xs, ys); torch.cuda.synchronize()
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                    for _ in range(10):
                        A.apply_multi(xs, ys)
                    ev1.record(); torch.cuda.synchronize()
                    t4 = ev0.elapsed_time(ev1) / 10
                    sec["SpMV 4 right-hand sides (SpMat * multivector<double,4>), %d^3" % n] = {
                        "ms": round(t4, 4), "gflops": round(8.0 * nnz_total / t4 / 1e6, 1)}
                    del xs, ys
                hell = None
                del A
                torch.cuda.empty_cache()
                if fmt in ("sell8v", "sell8"):
                    # the same product with less compact storage of the same matrix, for comparison: values as they
                    # are (diagonal codes only), and plain 32-bit columns (what a matrix without structure gets)
                    p2, c2, v2 = ops.poisson3d(n, dev)
                    xx = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42)
                    yy = torch.empty_like(xx)
                    for f2, label in (("sell8", "diagonal codes, fp64 values stored"), ("sell32", "32-bit columns, fp64 values stored")):
                        if f2 == fmt:
                            continue
                        B = ops.SpMat(p2, c2, v2, fmt=f2)
                        B.apply(xx, yy); torch.cuda.synchronize()
                        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        ev0.record()
                        for _ in range(20):
                            B.apply(xx, yy)
                        ev1.record(); torch.cuda.synchronize()
                        tb = ev0.elapsed_time(ev1) / 20
                        sec["SpMV same matrix, %s" % label] = {"ms": round(tb, 4), "gflops": round(2.0 * nnz_total / tb / 1e6, 1)}
                        del B
                    del p2, c2, v2, xx, yy
                    torch.cuda.empty_cache()
                sec.update(secondary_rows(torch, L, ops, dev, local_rank))
                out["secondary"] = sec
            except Exception as e:                   # the headline must not depend on the secondary rows
                out["secondary"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_grid, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
