#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric: fp64 CSR SpMV GFLOP/s + achieved HBM GB/s
on the 3-D Poisson matrix (examples/benchmark.cpp:353-477), 512^3 grid.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its N ranks itself,
                                                            and refuses when the box has fewer GPUs than ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one product y = A*x through vex::SpMat's path: the C-ABI object `vexhip_spmat`
(include/vexhip.h) that both front ends -- the C++ headers (vexcl/spmat.hpp) and the Python
mirror used here (vexcl_amd/ops.py) -- create and apply.  The matrix is built directly in HBM
(SURVEY 8(d)); inputs are resident before the timed region.  N > 1: the SAME 512^3 problem
row-partitioned over the ranks (strong scaling); the ghost planes travel by the fastest transport that reproduces an
evaluation of the stencil that involves no matrix and no transport (RCCL send/recv issued from C++, peer-mapped ghost
windows, or torch.distributed requests; `distributed.transports_tried`).

What the line reports, and how to read it:
  value / csr_algorithmic_gbps   2*nnz / t and CSR-ALGORITHMIC bytes / t (nnz*12 + (N+1)*4 + 16*N):
                      the metric's own definition, independent of how the matrix is stored -- NOT an achieved HBM rate when the
                      storage is smaller than CSR (it exceeds the 8 TB/s peak then); the achieved rate is roofline.achieved.
  ms_per_step         the MEDIAN of five blocks of K steps (median-of-5 of total / M), each block bracketed by a
                      barrier + synchronize on both sides; block 1 is the driver's "W warm-ups, then exactly K steps"; all five,
                      their minimum and maximum are under `blocks`.
  roofline            the launched kernel against the HBM roofline by the bytes it REALLY moves: the stored matrix
                      (vexhip_spmat_get_info: matrix_bytes) + x once + y once.  frac <= 1 by construction.  `traffic`
                      = HBM bytes per launch measured in THIS run (rocprofv3 FETCH_SIZE / WRITE_SIZE passes over
                      tools/pmc_headline.py, FETCH calibrated on a stream of known size), or null.  `device_copy_hand`
                      (plane / march product): x copied to y by a hand kernel (vexhip_stream_copy_f64: one 16-byte pair per
                      lane, non-temporal stores) in the same process -- the same HBM traffic as the product (x once, y once):
                      what the memory system gives a kernel with nothing else to do; `device_copy`: torch's copy of the same;
                      `frac_of_measured_copy` = hand copy time / product time.
  roofline_csr        the same product by kernels that stream fp64 values + int32 columns (no compression), priced
                      with the CSR-algorithmic bytes: SELL-512 with 32-bit columns and the CSR arrays themselves.
  variable_coefficient  the same 7-point pattern with a coefficient per face (~4 N distinct values: no value coding
                      applies) through the default SpMat: the general-matrix figure.
  checksum            sum(y) asserted against an independent evaluation of the stencil (torch slicing, no matrix); for N > 1 every
                      rank checks its rows against x regenerated from the global index (no transport involved).
  setup               what building the storage from CSR arrays in HBM costs (ms, bytes, products to amortise it).
N > 1: rank 0 says on stderr where the run is ("[bench  12.3 s] ..."); the trial and settle runs of a transport are bounded
in time by a three-product probe whose result all ranks share.
"""
import argparse
import json
import os
import subprocess
import sys
import time

T_START = time.perf_counter()


def progress(msg):
    """rank 0, N > 1: where the run is (stderr) -- a multi-GPU run that stalls must say where"""
    if os.environ.get("RANK", "0") == "0" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        print("[bench %7.1f s] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s
KERNEL_OF = {"sell8v_grid": "sell8_grid_kernel", "sell8v": "sell8_pair_kernel<double, 7, true, false>", "sell8v_march": "sell8_march_kernel<double, 7>", "sell8v_plane": "sell8_plane_kernel", "sell8": "sell8_pair_kernel<double, 7, false, false>",
             "sell32": "sell_pair_kernel<double, 7>", "csr": "csr_stream2_kernel<double, int, false>", "hell": "hell_kernel"}


def algorithmic_bytes(n_rows, nnz):
    return nnz * 12 + (n_rows + 1) * 4 + n_rows * 8 + n_rows * 8


def cpu_baseline(grid, seconds):
    """The reference's CPU-device path restated (oracle/vex_oracle.c: 8 x threads contiguous row chunks, OpenMP --
    backend/opencl/source.hpp:255-268, kernel.hpp:166-171,193-194), every array first-touched by the thread that uses
    it, on the 512^3 matrix when host memory allows; plus the single-thread loop the reference harness itself prints as
    its "C++" line (examples/benchmark.cpp:447-453)."""
    import oracle
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 0
    g = grid
    while g > 64 and algorithmic_bytes(g ** 3, 7 * g ** 3) * 1.3 > avail:
        g //= 2
    r = oracle.cpu_baseline_poisson(g, seconds, 1)
    if r is None:
        return {"error": "host arrays for %d^3 could not be allocated" % g}
    N, nnz = g ** 3, oracle.poisson3d_nnz(g)
    per, per1 = r["seconds_per_product"], r["single_thread_seconds_per_product"]
    return {"value": round(2.0 * nnz / per / 1e9, 3), "unit": "GFLOP/s", "cores": r["threads"], "kind": "port",
            "hbm_equiv_gbps": round(algorithmic_bytes(N, nnz) / per / 1e9, 2),
            "single_thread": {"value": round(2.0 * nnz / per1 / 1e9, 3), "unit": "GFLOP/s",
                              "what": "the harness's own host loop, examples/benchmark.cpp:447-453"},
            "sample": "%d products of the %d^3 Poisson matrix (N=%d, nnz=%d) in %.1f s, OpenMP chunked csr_spmv restatement, "
                      "arrays first-touched in parallel by the chunks' owners" % (r["products"], g, N, nnz, per * r["products"])}


def independent_product(torch, x, n, variable_seed=None, lazy=False):
    """y = A*x for the Poisson matrix WITHOUT the matrix: boundary rows are identity, interior rows the 7-point stencil
    (examples/benchmark.cpp:364-415), evaluated with torch slicing.  Also returns sum |terms| for the tolerance."""
    h2i = float((n - 1) * (n - 1))
    X = x.view(n, n, n)
    y = x.clone()
    Y = y.view(n, n, n)
    c = X[1:-1, 1:-1, 1:-1]
    nb = (X[:-2, 1:-1, 1:-1], X[1:-1, :-2, 1:-1], X[1:-1, 1:-1, :-2], X[1:-1, 1:-1, 2:], X[1:-1, 2:, 1:-1], X[2:, 1:-1, 1:-1])
    acc = 6.0 * h2i * c
    mag = 6.0 * h2i * c.abs()
    for t in nb:
        acc = acc - h2i * t
        mag = mag + h2i * t.abs()
    Y[1:-1, 1:-1, 1:-1] = acc
    if lazy:                                   # device scalar, no host synchronisation (read back by the caller later)
        return y, mag.sum() + x.abs().sum()
    bound = float(mag.sum()) + float(x.abs().sum())
    return y, bound


GOLD = 0x9E3779B97F4A7C15


def independent_strip(torch, ops, n, r0, r1, dev):
    """Rows [r0, r1) of y = A*x for the Poisson matrix WITHOUT the matrix and WITHOUT any transport: x is a hash of the
    GLOBAL index, so the rank generates x on its planes plus one ghost plane on each side itself and evaluates the
    stencil there (boundary rows identity, interior rows the 7-point stencil, examples/benchmark.cpp:364-415).
    Returns (x[r0:r1], y[r0:r1], sum|terms| per row)."""
    nn = n * n
    p0, p1 = r0 // nn, (r1 + nn - 1) // nn                   # planes that hold rows of the strip
    q0, q1 = max(p0 - 1, 0), min(p1 + 1, n)                  # ... plus the neighbours their stencils reach
    xs = ops.fill_hash(torch.empty((q1 - q0) * nn, dtype=torch.float64, device=dev), (42 + q0 * nn * GOLD) & 0xFFFFFFFFFFFFFFFF)
    X = xs.view(q1 - q0, n, n)
    Y = X[p0 - q0:p1 - q0].clone()
    M = Y.abs()
    k0, k1 = max(p0, 1), min(p1, n - 1)                      # interior planes of the strip
    if k1 > k0 and n > 2:
        h2i = float((n - 1) * (n - 1))
        a, b = k0 - q0, k1 - q0
        c = X[a:b, 1:-1, 1:-1]
        nb = (X[a - 1:b - 1, 1:-1, 1:-1], X[a:b, :-2, 1:-1], X[a:b, 1:-1, :-2], X[a:b, 1:-1, 2:], X[a:b, 2:, 1:-1], X[a + 1:b + 1, 1:-1, 1:-1])
        acc = 6.0 * h2i * c
        mag = 6.0 * h2i * c.abs()
        for t in nb:
            acc = acc - h2i * t
            mag = mag + h2i * t.abs()
        Y[k0 - p0:k1 - p0, 1:-1, 1:-1] = acc
        M[k0 - p0:k1 - p0, 1:-1, 1:-1] = mag
    lo = r0 - p0 * nn
    sl = slice(lo, lo + (r1 - r0))
    return xs[(p0 - q0) * nn:][sl].clone(), Y.reshape(-1)[sl].clone(), M.reshape(-1)[sl].clone()


def self_launch(args, argv):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here (one process per GPU) -- and refuse
    when the box has fewer GPUs than ranks instead of quietly measuring one."""
    import socket
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < 1:
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if have < args.gpus and not args.one_device:
        raise SystemExit("bench.py --gpus %d: this box has %d GPU%s (hipGetDeviceCount); one process per GPU is the only mode "
                         "that is measured -- use --one-device --backend gloo to exercise the %d-rank path on one GPU (debug, not a result)"
                         % (args.gpus, have, "" if have == 1 else "s", args.gpus))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    extra = []
    if args.one_device and "--backend" not in " ".join(argv):
        extra = ["--backend", "gloo"]                       # RCCL refuses two ranks on one device
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv + extra
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(cmd, env=env))


def try_transports(names, enable, validate, trial, disable, error_of, settle=lambda: None):
    """`--transport auto`: every candidate is enabled, validated against the independent evaluation and timed; a transport that raises,
    times out or disagrees at ANY of these stages is recorded as invalid and skipped -- never fatal (a first run on real multi-GPU
    hardware meets transports this project could only exercise on one device).  -> {name: record}.  The callables agree across the
    ranks themselves (enable / validate end in an all-reduce); an exception on one rank must not leave the others waiting: the record
    is made invalid on every rank by the next agreeing call of the caller."""
    tried = {}
    for tr in names:
        rec = {"valid": False}
        try:
            settle()
            if enable(tr):
                rec["valid"], rec["max_abs_err"] = validate()
                if rec["valid"]:
                    rec["trial_ms_per_step"] = round(trial(), 5)
        except BaseException as e:          # noqa: BLE001 -- SystemExit / KeyboardInterrupt are re-raised below
            if isinstance(e, (SystemExit, KeyboardInterrupt)):
                raise
            rec = {"valid": False, "error": repr(e)[:300]}
        if not rec["valid"] and "error" not in rec:
            rec["error"] = error_of() or "results differ from the independent evaluation on some rank"
        tried[tr] = rec
        try:
            settle()
            disable()
        except Exception as e:              # noqa: BLE001
            rec.setdefault("disable_error", repr(e)[:200])
    return tried


def transports_by_time(tried):
    """valid transports, fastest first (timed ones before merely valid ones)"""
    timed = sorted((k for k in tried if tried[k].get("valid") and "trial_ms_per_step" in tried[k]), key=lambda k: tried[k]["trial_ms_per_step"])
    return timed + [k for k in tried if tried[k].get("valid") and k not in timed]


def timed_events(torch, fn, reps):
    """Average duration of fn over `reps` launches, after ~30 ms of the same launches: these rows run after host-side work
    (matrix set-up, subprocesses), and after >= 5 ms without work the first ~16 ms of launches are up to 12 % slow."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    for _ in range(min(200, int(30.0 / max(e0.elapsed_time(e1), 0.05)))):
        fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def measure_traffic(grid, timeout=240, only=None, names=None, extra_env=None):
    """HBM bytes per launch of the headline kernels measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE need
    separate passes: MI355X_MICROARCH.md, rocprofv3 PMC slots) over tools/pmc_headline.py.  FETCH_SIZE is calibrated on
    a 2 GiB 16-byte-per-lane stream captured in the same pass (gfx950 reports half the bytes of such a stream)."""
    import csv, glob, shutil, tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return None, "already running under a profiler"
    out = tempfile.mkdtemp(prefix="vexpmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", GRID=str(grid))
    if only:
        env["PMC_ONLY"] = only                 # tools/pmc_headline.py: just these storages (comma-separated)
    if extra_env:
        env.update(extra_env)
    wanted = tuple(names) if names else ("sell8_grid_kernel", "sell8_plane_kernel", "sell8_march_kernel", "sell8_pair_kernel", "sell_pair_kernel", "csr_stream2_kernel")
    res = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, ctr)
            p = subprocess.run([exe, "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "pmc", "--output-format", "csv", "--",
                                sys.executable, os.path.join(ROOT, "tools", "pmc_headline.py")],
                               cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
            if p.returncode != 0:
                return None, "rocprofv3 --pmc %s exited %d" % (ctr, p.returncode)
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] != ctr:
                        continue
                    name = r["Kernel_Name"]
                    key = None
                    for k in wanted + ("reduce_stage1",):
                        if k + "<" in name or k + "(" in name:
                            key = k
                            if k == "sell8_pair_kernel":
                                targs = [a.strip() for a in name.split("sell8_pair_kernel<")[1].split(">")[0].split(",")]   # <V, W, VCODED, DICT>
                                key += "_vcoded" if targs[2] in ("true", "1") else "_values"
                    if key:
                        res.setdefault(key, {}).setdefault(ctr, []).append(float(r["Counter_Value"]))
    except subprocess.TimeoutExpired:
        return None, "rocprofv3 pass timed out"
    finally:
        shutil.rmtree(out, ignore_errors=True)
    cal = res.get("reduce_stage1", {}).get("FETCH_SIZE")
    if not cal:
        return None, "no calibration kernel in the counter output"
    factor = float((1 << 28) * 8) / (sum(cal) / len(cal) * 1024.0)
    out = {"fetch_calibration_factor": round(factor, 4)}
    for k, d in res.items():
        if k == "reduce_stage1" or "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
            continue
        rd = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 1024.0 * factor
        wr = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024.0
        out[k] = {"read": int(rd), "written": int(wr), "total": int(rd + wr)}
    return out, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes in this run (tools/pmc_headline.py)"


def cpp_rows(args_list, timeout=300, env=None):
    """Rows printed by a C++ example (JSON objects, one per line): the vex:: header API end to end."""
    exe = os.path.join(ROOT, "examples", "build", args_list[0])
    if not os.path.exists(exe):
        return [{"error": "%s is not built (python -c 'import __graft_entry__ as g; g.build()')" % args_list[0]}]
    p = subprocess.run([exe] + [str(a) for a in args_list[1:]], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout,
                       env=dict(os.environ, TMPDIR="/tmp", **(env or {})))
    rows = []
    for line in p.stdout.decode(errors="replace").splitlines():
        line = line.strip()
        if line.startswith("{"):
            try:
                rows.append(json.loads(line))
            except Exception:
                pass
    if p.returncode != 0:
        rows.append({"error": "%s exited %d" % (args_list[0], p.returncode)})
    return rows


def unstructured_rows(torch, ops, dev, args):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import unstructured as U
    rows = {}
    m = int(float(os.environ.get("UNSTRUCTURED_ROWS", "2e7")))
    x = ops.fill_hash(torch.empty(m, dtype=torch.float64, device=dev), 42)
    y = torch.empty(m, dtype=torch.float64, device=dev)
    what = {"random16": "16 entries per row, columns uniform in [0, n), sorted (the shape of tests/random_matrix.hpp)",
            "powerlaw": "row lengths floor(6 / sqrt(u)) capped at 4096 (mean ~12), columns uniform, sorted",
            "banded16": "round 6, WITH locality: 16 entries per row, columns uniform within +-100 000 of the diagonal, sorted (a mesh-ordered operator as the memory system sees it)",
            "stencil27": "round 6, WITH locality: 27-point operator on 256^3, a different value in every entry (nothing to compress)"}
    # ---- round 5: a structured operator in TWO dimensions (5-point, 16384^2): walked along virtual 512-point lines by the plane product
    try:
        W = H = int(os.environ.get("BENCH_2D_SIDE", "16384"))
        ptr, col, val, h2i = U.stencil2d(W, H, dev)
        n2, nnz2 = W * H, int(col.numel())
        x2 = ops.fill_hash(torch.empty(n2, dtype=torch.float64, device=dev), 42)
        y2 = torch.empty_like(x2)
        A = ops.SpMat(ptr, col, val)
        del ptr, col, val
        A.ptr = A.col = A.val = None
        torch.cuda.empty_cache()
        A.apply(x2, y2)
        yr, mag = U.stencil2d_reference(x2, W, H, h2i)
        bad = int(((y2 - yr).abs() > 1e-10 * mag).sum())
        assert bad == 0, "2-D 5-point: %d rows outside 1e-10 * sum|terms|" % bad
        del yr, mag
        t = min(timed_events(torch, lambda: A.apply(x2, y2), 20) for _ in range(3))
        moved = A.matrix_bytes() + 16 * n2
        rows["SpMV 5-point 2-D %d^2 (y = A*x, default vexhip_spmat)" % W] = {
            "rows": n2, "nnz": nnz2, "storage": A.storage, "plane_plan": A.plane, "grid_plan": A.grid, "ms": round(t, 5), "gflops": round(2.0 * nnz2 / t / 1e6, 1),
            "rows_outside_tolerance": bad,
            "roofline": {"bound": "hbm", "bytes_per_launch": moved, "achieved": round(moved / t / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(moved / t / 1e6 / HBM_PEAK_GBPS, 4), "what": "stored matrix (class tables + 4 B per virtual line) + x once + y once"}}
        del A, x2, y2
        torch.cuda.empty_cache()
    except AssertionError:
        raise
    except Exception as e:  # noqa: BLE001 -- a secondary row
        rows["SpMV 5-point 2-D"] = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()
    # ---- round 6: structured operators OUTSIDE the grid storage's pattern -- 2-D rows that are not whole 512-point lines (march product of
    # the SELL-512 storage) and a constant-coefficient 27-point operator (value-coded slices of width 27, pooled in the slice dictionary)
    for name, make in (("SpMV 5-point 2-D 12000^2 (y = A*x, default vexhip_spmat)", lambda: U.stencil2d(12000, 12000, dev)[:3]),
                       ("SpMV 27-point constant coefficients 320^3 (y = A*x, default vexhip_spmat)", lambda: U.stencil27_const(320, dev))):
        try:
            ptr, col, val = make()
            nr, nz_ = ptr.numel() - 1, int(col.numel())
            xs = ops.fill_hash(torch.empty(nr, dtype=torch.float64, device=dev), 42)
            ys = torch.empty_like(xs)
            A = ops.SpMat(ptr, col, val)
            A.apply(xs, ys)
            yr, mag = U.reference_product(ptr, col, val, xs)
            bad = int(((ys - yr).abs() > 1e-10 * mag).sum())
            assert bad == 0, "%s: %d rows outside 1e-10 * sum|terms|" % (name, bad)
            del yr, mag, ptr, col, val
            A.ptr = A.col = A.val = None
            torch.cuda.empty_cache()
            t = min(timed_events(torch, lambda: A.apply(xs, ys), 20) for _ in range(3))
            moved = A.matrix_bytes() + 16 * nr
            alg = 12 * nz_ + 4 * (nr + 1) + 16 * nr
            rows[name] = {"rows": nr, "nnz": nz_, "storage": A.storage, "kernel": A.product, "selection": A.reason, "dictionary_blocks": A.dictionary_blocks,
                          "ms": round(t, 5), "gflops": round(2.0 * nz_ / t / 1e6, 1), "rows_outside_tolerance": bad,
                          "roofline": {"bound": "hbm", "bytes_per_launch": moved, "achieved": round(moved / t / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                       "frac": round(moved / t / 1e6 / HBM_PEAK_GBPS, 4), "algorithmic_bytes_per_launch": alg, "algorithmic_gbps": round(alg / t / 1e6, 1),
                                       "what": "stored matrix (pooled code blocks + 4 B per slice) + x once + y once; the kernel gathers x per entry through the L1 (5 / 27 "
                                               "16-byte requests per row pair): bound there, not by these bytes"}}
            del A, xs, ys
            torch.cuda.empty_cache()
        except AssertionError:
            raise
        except Exception as e:  # noqa: BLE001 -- a secondary row
            rows[name] = {"error": repr(e)[:300]}
            torch.cuda.empty_cache()
    # ---- round 5: the headline operator in fp32 (plane32.hip: the plane walk with four rows per lane; the march product took
    # float matrices until then) -- bit-compared with the library's CSR loop on the same arrays
    try:
        g = int(os.environ.get("BENCH_FP32_GRID", "512"))
        ptr, col, val = ops.poisson3d(g, dev)
        v32 = val.to(torch.float32)
        del val
        n3, nnz3 = g ** 3, int(col.numel())
        x3 = ops.fill_hash(torch.empty(n3, dtype=torch.float64, device=dev), 42).to(torch.float32)
        y3 = torch.empty_like(x3); yc = torch.empty_like(x3)
        A = ops.SpMat(ptr, col, v32)
        C = ops.SpMat(ptr, col, v32, fmt="csr")
        A.apply(x3, y3); C.apply(x3, yc)
        same = bool(torch.equal(y3, yc))
        assert same, "fp32 product differs from the CSR loop"
        del C, yc, ptr, col, v32
        A.ptr = A.col = A.val = None
        torch.cuda.empty_cache()
        t = min(timed_events(torch, lambda: A.apply(x3, y3), 20) for _ in range(3))
        moved = A.matrix_bytes() + 8 * n3
        rows["SpMV fp32 Poisson %d^3 (y = A*x, default vexhip_spmat)" % g] = {
            "rows": n3, "nnz": nnz3, "storage": A.storage, "plane_plan": A.plane, "ms": round(t, 5), "gflops": round(2.0 * nnz3 / t / 1e6, 1),
            "bit_identical_to_csr_loop": same,
            "roofline": {"bound": "hbm", "bytes_per_launch": moved, "achieved": round(moved / t / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(moved / t / 1e6 / HBM_PEAK_GBPS, 4), "traffic": None,
                         "what": "stored matrix (4 B per line + the class tables) + x once + y once, 4-byte elements"}}
        del A, x3, y3
        torch.cuda.empty_cache()
        if not args.no_pmc:
            tr, how = measure_traffic(g, only="fp32", names=("sell8_plane_f32_kernel", "sell8_march_kernel", "sell8_pair_kernel"))
            rf = rows["SpMV fp32 Poisson %d^3 (y = A*x, default vexhip_spmat)" % g]["roofline"]
            rf["traffic_source"] = how
            if tr:
                ks = {k: v for k, v in tr.items() if isinstance(v, dict)}
                total = sum(v["total"] for v in ks.values())
                rf["traffic"] = total
                rf["traffic_kernels"] = ks
                rf["traffic_over_bytes_per_launch"] = round(total / float(moved), 3)
    except AssertionError:
        raise
    except Exception as e:  # noqa: BLE001 -- a secondary row
        rows["SpMV fp32 Poisson"] = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()
    m0, x0, y0 = m, x, y
    for name in ("random16", "powerlaw", "banded16", "stencil27"):
        try:
            m, x, y = m0, x0, y0
            if U.ROWS_OF.get(name, m0) != m0:
                m = U.ROWS_OF[name]
                x = ops.fill_hash(torch.empty(m, dtype=torch.float64, device=dev), 42); y = torch.empty(m, dtype=torch.float64, device=dev)
            ptr, col, val = U.MAKERS[name](m, dev)
            nnz = int(col.numel())
            yr, mag = U.reference_product(ptr, col, val, x)
            A = ops.SpMat(ptr, col, val)
            A.apply(x, y)
            bad = int(((y - yr).abs() > 1e-10 * mag).sum())
            assert bad == 0, "%s: %d rows outside 1e-10 * sum|terms| of the row" % (name, bad)
            del yr, mag
            t = min(timed_events(torch, lambda: A.apply(x, y), 10) for _ in range(3))
            stored = A.matrix_bytes() + 16 * m            # what the product must move at least: the stored matrix + x once + y once
            alg = algorithmic_bytes(m, nnz)
            row = {"what": what[name], "rows": m, "nnz": nnz, "storage": A.storage, "ell_width": getattr(getattr(A, "hell", None), "width", None), "tail_nnz": getattr(getattr(A, "hell", None), "tail_nnz", None),
                   "ms": round(t, 5), "gflops": round(2.0 * nnz / t / 1e6, 1), "rows_outside_tolerance": bad,
                   "roofline": {"bound": "hbm", "bytes_per_launch": stored, "achieved": round(stored / t / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": round(stored / t / 1e6 / HBM_PEAK_GBPS, 4),
                                "algorithmic_bytes_per_launch": alg, "algorithmic_gbps": round(alg / t / 1e6, 1),
                                "frac_of_algorithmic_bytes": round(alg / t / 1e6 / HBM_PEAK_GBPS, 4), "traffic": None,
                                "what": ("the columns of a row lie close to its own: the lines of x a wave gathers from are shared with its neighbours and stay in the caches; "
                                         "traffic / bytes_per_launch says how much of x is fetched more than once") if name in ("banded16", "stencil27") else
                                        ("x is gathered 8 bytes at a time from lines nobody else in the wave uses: the memory system moves a whole "
                                         "sector per entry (traffic below), the bytes priced here are the floor (matrix + x once + y once)")}}
            if name in ("banded16", "random16"):
                # A/B: the same matrix with its slices dealt round-robin to the XCDs (every XCD's L2 then sees every eighth slice)
                B = ops.SpMat(ptr, col, val, plain_order=True)
                tb = min(timed_events(torch, lambda: B.apply(x, y), 10) for _ in range(2))
                row["slices_dealt_round_robin_ms"] = round(tb, 5)
                del B
            if name == "random16":
                # the bound this access pattern has: every entry pulls its own 128-byte line of x through the fabric (x, 160 MB, lies
                # under the 256 MiB Infinity Cache: FETCH_SIZE counts those lines) -- nnz lines at the rate the fetch path sustains
                # (the read kernel of tools/r06_mall_probe.hip on a working set of that size, non-temporal loads: 7.2 TB/s, profiles/r06_mall_probe.json)
                row["roofline"]["bound_ms"] = round((nnz * 128.0 + A.matrix_bytes()) / 7.2e12 * 1e3, 3)
                row["roofline"]["bound_what"] = ("nnz x 128-byte lines of x + the stored matrix at 7.2 TB/s, the rate at which a read-only kernel re-reads a 192 MiB "
                                                 "working set on this part (profiles/r06_mall_probe.json): the wall of this access pattern, not of the kernel")
            del A, ptr, col, val
            torch.cuda.empty_cache()
            if not args.no_pmc:
                tr, how = measure_traffic(8, only=name, names=U.PRODUCT_KERNELS, extra_env={"UNSTRUCTURED_ROWS": str(m), "GRID": "8"})
                row["roofline"]["traffic_source"] = how
                if tr:
                    ks = {k: v for k, v in tr.items() if isinstance(v, dict)}
                    total = sum(v["total"] for v in ks.values())
                    row["roofline"]["traffic"] = total
                    row["roofline"]["traffic_kernels"] = ks
                    row["roofline"]["traffic_over_bytes_per_launch"] = round(total / float(stored), 3)
                    row["roofline"]["traffic_over_algorithmic_bytes"] = round(total / float(alg), 3)
                    row["roofline"]["traffic_gbps"] = round(total / t / 1e6, 1)
            rows["SpMV unstructured: %s, %d rows (y = A*x, default vexhip_spmat)" % (name, m)] = row
        except AssertionError:
            raise
        except Exception as e:  # noqa: BLE001 -- a secondary row
            rows["SpMV unstructured: %s" % name] = {"error": repr(e)[:300]}
            torch.cuda.empty_cache()
    return rows


def secondary_rows(torch, L, ops, dev, local_rank):
    """BASELINE.json's secondary metrics.  Elementwise and reduce come from the C++ expression engine
    itself (examples/roofline.cpp: the kernels vexcl/operations.hpp and vexcl/reductor.hpp generate); scan and sort are
    timed through the C ABI on pre-allocated buffers.  Reported next to the headline value; never part of it."""
    import ctypes
    rows = {}
    for r in cpp_rows(["roofline", 1000000, "e"]):
        if "row" in r:
            rows["C++ vex:: " + r["row"]] = {k: r[k] for k in ("ms", "alg_gbps", "frac_of_8TBps") if k in r}
        elif "error" in r:
            rows["C++ roofline"] = r
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    n = 10 ** 9
    k = ops.fill_hash(torch.empty(n, dtype=torch.int32, device=dev), 42)
    out = torch.empty_like(k)
    ms = timed_events(torch, lambda: ops.inclusive_scan(k, out, unsigned=True), 10)
    rows["inclusive_scan u32 n=1e9"] = {"ms": round(ms, 4), "gbps": round(8.0 * n / ms / 1e6, 1), "frac_of_8TBps": round(8.0 * n / ms / 1e6 / HBM_PEAK_GBPS, 4)}
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
    rows["sort u32 keys n=1e9 (stable LSD radix)"] = {"ms": round(best, 3), "gkeys_per_s": round(n / best / 1e6, 1),
                                                      "lower_bound_frac_of_8TBps": round(8.0 * n / best / 1e6 / HBM_PEAK_GBPS, 4)}
    del ktmp, tmp
    torch.cuda.empty_cache()
    # yardstick, not the product: the library's sort (torch.sort = rocPRIM's radix sort) on the same keys, same box.  Its
    # allocations (values + indices) are part of the call, so the time is an upper bound of rocPRIM's own kernel time.
    try:
        best = None
        for _ in range(2):
            ops.fill_hash(k, 42); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sk = torch.sort(k.view(torch.int32))
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1)
            best = t if best is None else min(best, t)
            del sk
        rows["torch.sort = rocPRIM PAIRS sort on the same 1e9 keys (int32 keys + int64 indices: 12 B per element and pass, NOT a like-for-like keys-only sort)"] = {"ms": round(best, 3), "gpairs_per_s": round(n / best / 1e6, 1)}
    except Exception as e:  # noqa: BLE001 -- a comparison, not the measurement
        rows["sort yardstick: torch.sort"] = {"error": repr(e)[:200]}
    del k
    torch.cuda.empty_cache()
    return rows


def main():
    # a flag wait of the peer-window transports gives up after 20 s by default (ranks of a job may be seconds apart); inside this bench every
    # product stands between barriers, so a flag that does not come within 5 s never comes: a transport that does not work on this
    # machine then costs --transport auto seconds, not minutes (read once, when the library first waits)
    os.environ.setdefault("VEXHIP_IPC_TIMEOUT_MS", "5000")
    if os.environ.get("BENCH_DUMP_AFTER"):            # diagnostics: where is every rank after that many seconds?
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_DUMP_AFTER"]), exit=False)
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
    ap.add_argument("--cpu-grid", type=int, default=512)
    ap.add_argument("--dist", action="store_true", help="use the partitioned SpMat even on one GPU (debug)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (gloo: debug)")
    ap.add_argument("--one-device", action="store_true", help="all ranks on cuda:0 (debug: exercises the N>1 path on a 1-GPU box; needs --backend gloo)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary rows (CSR-bytes kernels, variable coefficients, C++ front end, elementwise, reduce, scan, sort)")
    ap.add_argument("--no-native", action="store_true", help="N > 1: keep the torch.distributed transport (do not try the C++ product step)")
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "ipc", "halo", "torch"],
                    help="N > 1 ghost exchange: rccl = grouped ncclSend/ncclRecv issued from C++; ipc = peer-mapped ghost windows "
                         "(hipIpcGetMemHandle); torch = torch.distributed requests; auto = every one that validates, fastest wins")
    ap.add_argument("--trial-steps", type=int, default=100, help="N > 1, --transport auto: products timed per candidate transport")
    ap.add_argument("--no-dictionary", action="store_true", help="value-coded storage with one code block per slice (no slice dictionary)")
    ap.add_argument("--no-march", action="store_true", help="keep the pair product where the march / plane products would apply")
    ap.add_argument("--no-direct", action="store_true", help="build the SELL-512 storage (slices, dictionary, plans) where the default set-up stores the matrix by grid line (round 4)")
    ap.add_argument("--no-plane", action="store_true", help="keep the march product (x window in an LDS ring, round 3) where the plane product (round 4) would apply")
    ap.add_argument("--blocks", type=int, default=5, help="blocks of K steps; ms_per_step is their median (block 1 = the W warm-ups + exactly K steps of the contract)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the ~3 s back-to-back run of the product after the timed region")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 counter passes (roofline.traffic = null)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args, sys.argv[1:])                      # does not return

    import torch
    import torch.distributed as dist
    from vexcl_amd import lib, ops
    from vexcl_amd.distributed import DistReductor, DistSpMat, partition

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.one_device:
        local_rank = 0
        if world > 1 and args.backend == "nccl":
            raise SystemExit("--one-device puts every rank on cuda:0, which RCCL refuses: add --backend gloo")
    elif torch.cuda.device_count() < world or local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py --gpus %d: this box has %d GPU(s); one process per GPU is the only mode that is measured"
                         % (world, torch.cuda.device_count()))
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
    single = world == 1 and not args.dist

    # ---- inputs resident in HBM before the timed region
    progress("process group up; generating rows [%d, %d) of the %d^3 matrix" % (r0, r1, n))
    ptr, col, val = ops.poisson3d(n, dev, rows=(r0, r1))
    # x is a function of the GLOBAL index, so the job computes the same product for every N
    # (x and y placed as the library places every vex::vector -- vexhip_malloc, round 6: at a multiple of 64 MiB plus a small stagger;
    #  where y lies relative to x is worth up to 11 % of this product: profiles/r06_xy_gap.json)
    x = ops.fill_hash(ops.device_vector(r1 - r0, torch.float64, dev),
                      (42 + r0 * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
    y = ops.device_vector(r1 - r0, torch.float64, dev, zero=True)
    setup = None
    march = None
    plane = None
    grid_plan = None
    direct = False
    if single:
        # set-up is timed (not part of `value`): CSR arrays in HBM -> the storage the product runs on.  A solver that rebuilds
        # its matrices (AMG set-up, a nonlinear iteration) pays this once per matrix.
        # The device is brought out of idle first (30 copies of x, ~12 ms): a set-up that follows a pause is clocked up to 5 ms slower
        # in its first kernel than one in the middle of a computation, which is where a solver rebuilds a matrix.
        for _ in range(30):
            y.copy_(x)
        y.zero_()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(dev)[0]
        # (Python's cyclic collector is run NOW: a full collection over what `import torch` leaves behind takes ~40 ms, and whether
        #  the allocation counter trips inside the creation, inside the synchronisation that follows it or nowhere near depended on
        #  how many objects the prelude had made -- the first `setup_ms` of round 4 read 5, 41 and 54 ms for the same 5 ms of work)
        import gc
        gc.collect()
        ts0 = time.perf_counter()
        A = ops.SpMat(ptr, col, val, fmt=args.format, dictionary=not args.no_dictionary, march=not args.no_march, plane=not args.no_plane, direct=not args.no_direct)
        create_ms = (time.perf_counter() - ts0) * 1e3
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - ts0) * 1e3
        if os.environ.get("BENCH_STOP_AFTER_SETUP"):      # diagnostics
            print("setup_ms %.3f returned_after_ms %.3f" % (setup_ms, create_ms), flush=True)
            if os.environ["BENCH_STOP_AFTER_SETUP"] == "2":
                os._exit(0)
            return
        storage = A.storage
        matrix_bytes = A.matrix_bytes()
        dict_blocks = A.dictionary_blocks
        march = A.march
        plane = A.plane if x.dtype == torch.float64 else None
        grid_plan = A.grid if x.dtype == torch.float64 else None
        direct = bool(A.direct)
        setup = {"setup_ms": round(setup_ms, 3), "returned_after_ms": round(create_ms, 3),
                 "what": ("vexhip_spmat_create on CSR arrays resident in HBM (host wall time, synchronised; the first creation in this process, after 30 copies of x "
                          "that bring the device out of idle): " + ("ELL width, diagonal / value tables, then ONE pass that stores the matrix by grid line (classes of lines)"
                          if direct else "hybrid-ELL analysis, diagonal / value tables, fill, slice dictionary, march / plane plans")),
                 "stored_by_grid_line": direct,
                 "csr_input_bytes": int(ptr.numel() * ptr.element_size() + col.numel() * col.element_size() + val.numel() * val.element_size()),
                 "stored_bytes": int(matrix_bytes),
                 "held_after_setup_bytes": int(free0 - torch.cuda.mem_get_info(dev)[0])}
        if storage not in ("csr",):
            # a second creation on the same arrays: the set-up in the middle of a computation (every one-time cost of the process paid)
            torch.cuda.synchronize(); ts1 = time.perf_counter()
            A2 = ops.SpMat(ptr, col, val, fmt=args.format, dictionary=not args.no_dictionary, march=not args.no_march, plane=not args.no_plane, direct=not args.no_direct)
            torch.cuda.synchronize()
            setup["repeat_ms"] = round((time.perf_counter() - ts1) * 1e3, 3)
            del A2
            del ptr, col, val                    # the product only needs the converted storage
            A.ptr = A.col = A.val = None
        step = lambda: A.apply(x, y, 1.0, False)
    else:
        progress("strip generated; splitting into local / remote parts and planning the exchange")
        A = DistSpMat(ptr, col, val, N, N, local_fmt=args.format, keep_strip=True)      # (transport "halo" stores the strip once more with its ghost planes; dropped below)
        progress("exchange planned; independent evaluation of this rank's rows")
        storage = A.loc.storage
        matrix_bytes = A.loc.matrix_bytes()
        dict_blocks = getattr(A.loc, "dictionary_blocks", 0)
        march = getattr(A.loc, "march", None)
        plane = getattr(A.loc, "plane", None)
        grid_plan = getattr(A.loc, "grid", None)
        direct = bool(getattr(A.loc, "direct", False))
        step = lambda: A.apply(x, y, 1.0, False)

        # ---- every transport is validated against an evaluation that trusts NO transport: x is a hash of the global index,
        #      so the rank regenerates x on its planes + one ghost plane per side and evaluates the stencil matrix-free
        x_ind, y_ind, mag_ind = independent_strip(torch, ops, n, r0, r1, dev)
        assert torch.equal(x_ind, x), "x of this rank is not the hash of the global index"
        del x_ind
        tol_ind = 1e-10 * mag_ind                                # per row: 1e-10 * sum |terms|

        def validate(products=3):
            """`products` consecutive products (the exchange buffers / step counters are reused) into a poisoned y, each
            compared row by row with the independent evaluation; the verdict is the same on every rank."""
            ok, worst = True, 0.0
            try:
                for k in range(products):
                    y.fill_(7.0 + k)
                    A.apply(x, y, 1.0, False)
                    torch.cuda.synchronize()
                    d = (y - y_ind).abs()
                    bad = d > tol_ind
                    if bool(bad.any()):
                        ok = False
                        rows_bad = torch.nonzero(bad).flatten()
                        print("[bench rank %d] product %d: %d rows outside the tolerance, first local row %d (global %d), last %d; error there %.3e"
                              % (rank, k, rows_bad.numel(), int(rows_bad[0]), r0 + int(rows_bad[0]), int(rows_bad[-1]), float(d[rows_bad[0]])),
                              file=sys.stderr, flush=True)
                    worst = max(worst, float(d.max()))
                st = A.native_status()
                if st and st["timed_out"]:
                    ok = False
                    A.native_error = "a flag wait of the IPC transport ran into its bound (VEXHIP_IPC_TIMEOUT_MS)"
            except Exception as e:
                A.native_error = repr(e)
                ok = False
            return A._agree(ok), worst

        def timed_products(k):
            torch.cuda.synchronize(); barrier()
            t0 = time.perf_counter()
            for _ in range(k):
                step()
            torch.cuda.synchronize(); barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=A._coll_device())
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.cpu()[0]) / k

        def trial(steps):
            """ms per product over `steps` products between barriers (max over the ranks), after a settle run.  Both runs are
            bounded in time by a three-product probe whose result every rank shares (a transport that takes 0.2 s per product
            -- gloo with all ranks on one device -- must not cost minutes): settle <= 2 s, trial <= 5 s."""
            probe = max(timed_products(3), 1e-6)
            settle = int(min(50, 2.0 / probe))
            steps = int(max(3, min(steps, 5.0 / probe)))
            for _ in range(settle):
                step()
            return timed_products(steps) * 1e3

        def barrier():
            if world > 1:
                dist.barrier()

        tried = {}
        progress("matrix partitioned and converted on %d ranks; validating torch.distributed (%s)" % (world, args.backend))
        ok, worst = validate()
        tried["torch"] = {"valid": ok, "max_abs_err": worst, "what": "torch.distributed batch_isend_irecv (%s), pack and products launched from Python" % args.backend}
        candidates = []
        if not args.no_native and args.transport != "torch":
            # (three or more ranks SHARING one device can starve each other in the one-launch step: the workgroups that wait for a
            #  ghost plane hold their CU slots, csrc/halo.hpp -- with a GPU per rank a launch only waits for other devices)
            if args.transport == "pull" or (args.transport == "auto" and not (args.one_device and world > 2)):
                candidates.append("pull")           # round 6: one launch, the neighbours' boundary planes of x read in place (their allocations mapped through IPC handles)
            if args.transport == "halo" or (args.transport == "auto" and not (args.one_device and world > 2)):
                candidates.append("halo")           # round 5: the whole step in one launch (ghost planes read by the plane product itself)
            if args.transport in ("auto", "ipc"):
                candidates.append("ipc")
            if args.transport in ("auto", "rccl") and (args.backend == "nccl" or world == 1):
                candidates.append("rccl")
        if args.transport == "auto" or not candidates:
            if tried["torch"]["valid"]:
                tried["torch"]["trial_ms_per_step"] = round(trial(args.trial_steps), 5)
        def settle():
            torch.cuda.synchronize(); barrier()

        def enable(tr):
            progress("transports so far: %r; trying %s" % ({k: v.get("trial_ms_per_step", v.get("valid")) for k, v in tried.items()}, tr))
            return A.enable_native(transport=tr)

        native = try_transports(candidates, enable, validate, lambda: trial(args.trial_steps), A.disable_native, lambda: A.native_error, settle)
        for tr, rec in native.items():
            if rec.get("valid"):
                rec["status"] = None
            tried[tr] = rec
        order = transports_by_time(tried)
        if not order:
            raise SystemExit("no ghost-exchange transport reproduces the independent evaluation of the product: %r" % tried)
        progress("transports, fastest first: %r of %r" % (order, {k: v.get("trial_ms_per_step", v.get("valid")) for k, v in tried.items()}))
        # the fastest one that comes up AGAIN and still validates is used (one that fails now is skipped like one that failed above)
        chosen = None
        for k in order:
            if k == "torch":
                chosen = k
                break
            try:
                if A.enable_native(transport=k):
                    ok, _ = validate(1)
                    if ok:
                        tried[k]["status"] = A.native_status()
                        if k == "rccl":
                            tried[k]["rccl"] = A.rccl_info()
                        chosen = k
                        break
                tried[k]["second_enable_error"] = A.native_error or "did not validate after it was re-enabled"
            except Exception as e:          # noqa: BLE001
                tried[k]["second_enable_error"] = repr(e)[:300]
            A.disable_native()
        if chosen is None:
            raise SystemExit("no ghost-exchange transport could be enabled a second time: %r" % tried)
        A.drop_strip()                 # (kept by DistSpMat for transport "halo", which stores the strip once more with its ghost planes)
        transport = {"torch": "torch.distributed batch_isend_irecv (%s)" % args.backend,
                     "rccl": "vexhip_dist_spmv_apply: pack + grouped ncclSend/ncclRecv + local + remote part issued from C++ (own RCCL communicator)",
                     "ipc": "vexhip_dist_spmv_apply over peer-mapped ghost windows (hipIpcGetMemHandle): owners write their neighbours' shares "
                            "into the consumers' windows, step-numbered flags, no communicator",
                     "halo": "vexhip_dist_spmv_apply, ONE product launch per step (vexhip_dist_spmv_create_halo): the strip stored with its two ghost "
                             "planes, the plane product reads them from the peer-mapped window behind the owners' flags and its first workgroups "
                             "push the rank's boundary planes; no second stream, no remote part",
                     "pull": "vexhip_dist_spmv_apply_pull, ONE product launch per step (vexhip_dist_spmv_create_halo_pull): the strip stored with its two "
                             "ghost planes, read IN PLACE from the neighbours' x (their allocations mapped through IPC handles) behind 'x is final' flags; "
                             "nothing is pushed"}.get(chosen, chosen)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- the result is checked inside the run: an evaluation of the stencil that never touches the matrix.  Its kernels are
    #      QUEUED here, ahead of the warm-up, and their three scalars are read back after the timed region.  Where it stands
    #      matters for the clock: after >= 5 ms without work the next ~20 launches of this product run up to 12 % slow
    #      (tools/r02_ramp.py), and with the driver's W = 5, K = 20 the whole timed region would sit in that transient; this
    #      way the device goes from the set-up kernels through the check straight into the warm-up without draining.
    def queue_check():
        yref, bound_dev = independent_product(torch, x, n, lazy=True)
        return ((y - yref).abs().max(), yref.sum(dtype=torch.float64), bound_dev)

    check_dev = None
    if single:
        # The first evaluation only loads torch's kernels and fills its allocator's cache (both stall the host for
        # milliseconds at a time: the device drains); the second one, which is the check, then runs without a gap.
        step()
        queue_check()
        step()
        check_dev = queue_check()

    # N > 1: the set-up above (partition exchange plan, validation of the C++ step) ends in host-side waits; the same 200
    # products on every rank bring the devices out of the post-idle transient before the W warm-ups (reported as settle_steps)
    settle_steps = 0
    if not single:
        per_step_ms = tried[chosen].get("trial_ms_per_step", 1.0)          # the same on every rank (all-reduced)
        settle_steps = int(max(3, min(200, 2000.0 / max(per_step_ms, 1e-3))))
        for _ in range(settle_steps):
            step()

    # HIP events on the stream the kernels are launched on (torch's current stream)
    import ctypes
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    L.event_create(local_rank, 1, ctypes.byref(e0))
    L.event_create(local_rank, 1, ctypes.byref(e1))

    import gc
    gc.disable()                       # a full collection takes ~40 ms in this process (see the set-up above): none inside the blocks of K x 0.37 ms
    for _ in range(args.warmup):       # (and none right in front of them either: the device would idle and clock the first block down)
        step()
    # Five blocks of exactly K steps, each bracketed by a barrier + torch.cuda.synchronize() on both sides and by HIP events on
    # the launch stream; the wall time of a block is the MAX over the ranks.  Block 1 is the driver's contract (W warm-ups, then
    # exactly K steps); the headline is the median block (median-of-5 of total / M).
    block_wall, block_kern = [], []
    for _blk in range(max(1, args.blocks)):
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
        block_wall.append(t1 - t0); block_kern.append(ms.value)
    gc.enable()
    if world > 1:
        t = torch.tensor(block_wall, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        block_wall = [float(v) for v in t]
    order = sorted(range(len(block_wall)), key=lambda i: block_wall[i])
    mid = order[len(order) // 2]

    checksum = DistReductor("SUM_Kahan")(y)          # sum(y): the same for every N up to rounding
    elapsed = block_wall[mid]
    per_step = elapsed / args.steps
    kern_s = block_kern[mid] / 1e3 / args.steps   # average launch duration of the product on this rank in the median block (HIP events)
    ms = ctypes.c_float(block_kern[mid])

    check = None
    dist_extra = None
    if not single:
        # the y of the timed steps against the independent evaluation, then the slowest rank's phases
        d = (y - y_ind).abs()
        row_ok = bool((d <= tol_ind).all())
        st = A.native_status()
        mine = {"rank": rank, "rows": r1 - r0, "avg_launch_ms": round(ms.value / args.steps, 5), "max_abs_err": float(d.max()),
                "rows_within_tolerance": row_ok, "timed_out": (st or {}).get("timed_out", 0)}
        prof = None
        if chosen != "torch":
            reps = [A.profile_step(x, y) for _ in range(7)]
            prof = {k: round(sorted(r[k] for r in reps)[3], 5) for k in reps[0]}       # median of 7
        mine["step_ms"] = prof
        everyone = [None] * world
        if world > 1:
            dist.all_gather_object(everyone, mine)
        else:
            everyone = [mine]
        sum_ind = DistReductor("SUM_Kahan")(y_ind)
        bound_all = DistReductor("SUM")(mag_ind)
        slowest = max(everyone, key=lambda r: r["avg_launch_ms"])
        check = {"sum_y": checksum, "sum_y_independent": sum_ind, "max_abs_err": max(r["max_abs_err"] for r in everyone),
                 "tolerance": "per row 1e-10 * sum|terms| of that row (SURVEY 8c); sum(y) within 1e-10 * %.3e" % bound_all,
                 "what": "every rank regenerates x (a hash of the global index) on its planes + one ghost plane per side and evaluates "
                         "the stencil matrix-free: no matrix, no transport involved; compared with the y of the timed steps"}
        assert all(r["rows_within_tolerance"] and not r["timed_out"] for r in everyone), "a rank's product does not match the independent evaluation: %r" % everyone
        assert abs(checksum - sum_ind) <= 1e-10 * bound_all, "sum(y) does not match the independent evaluation: %r" % check
        dist_extra = {"per_rank": everyone, "slowest_rank": slowest["rank"], "slowest_rank_step_ms": slowest["step_ms"],
                      "step_ms_what": "median of 7 products with HIP events on both streams (vexhip_dist_spmv_profile): total, local part, wait "
                                      "for the ghosts after the local part, remote part; pack and exchange run beside the local part",
                      "transports_tried": tried, "transport_chosen": chosen, "rccl": A.rccl_info()}
    if single:
        err, sum_ref, bound = (float(v) for v in check_dev)
        tol = 1e-10 * bound
        check = {"sum_y": checksum, "sum_y_independent": sum_ref, "max_abs_err": err,
                 "tolerance": "1e-10 * sum|terms| = %.3e (SURVEY 8c)" % tol,
                 "what": "torch slicing of x on the n^3 grid, no matrix involved; evaluated before the warm-up, read back after the timed region"}
        assert abs(checksum - sum_ref) <= tol and err <= tol, "product does not match the independent stencil evaluation: %r" % check

    # ---- the same product launched back to back for ~3 s: a cross-check of the K-step figure against a long run (and GPU
    #      activity an outside sampler of the device can see; the timed region above lasts K x 0.8 ms)
    sustained = None
    if single and not args.no_sustained:
        nsoak = min(6000, max(args.steps, int(3.0 / per_step)))
        L.event_record(local_rank, e0, stream)
        for _ in range(nsoak):
            step()
        L.event_record(local_rank, e1, stream)
        torch.cuda.synchronize()
        ms2 = ctypes.c_float()
        L.event_elapsed_ms(local_rank, e0, e1, ctypes.byref(ms2))
        sustained = {"steps": nsoak, "ms_per_step": round(ms2.value / nsoak, 5),
                     "gflops": round(2.0 * nnz_total / (ms2.value / nsoak) / 1e6, 1),
                     "what": "the same product launched back to back after the timed region (HIP events on the launch stream)"}

    if setup is not None:
        setup["products_to_amortise"] = round(setup["setup_ms"] / (per_step * 1e3), 1)
    if rank == 0:
        nnz_rank = L.poisson3d_strip_nnz(n, r0, r1)
        rows_rank = r1 - r0
        alg_total = algorithmic_bytes(N, nnz_total)
        alg_rank = algorithmic_bytes(rows_rank, nnz_rank)
        moved_rank = matrix_bytes + 16 * rows_rank                 # what the launched kernel streams: stored matrix + x + y
        gflops = 2.0 * nnz_total / per_step / 1e9
        gbps = alg_total / per_step / 1e9
        out = {
            "metric": "fp64 CSR SpMV GFLOP/s, 3D Poisson %d^3 (y = A*x through vex::SpMat's library object vexhip_spmat)" % n,
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
            "front_end": "vexcl_amd/ops.py SpMat -> vexhip_spmat_create / vexhip_spmat_apply_f64 (the C++ vex::SpMat calls the same two; "
                         "its own timing of this product is under secondary['C++ front end'])",
            "blocks": {"what": "blocks of K steps, each between barrier + synchronize; ms_per_step is the median block; block 1 = W warm-ups then exactly K steps",
                       "ms_per_step": [round(w / args.steps * 1e3, 5) for w in block_wall],
                       "min": round(min(block_wall) / args.steps * 1e3, 5), "max": round(max(block_wall) / args.steps * 1e3, 5),
                       "event_ms_per_step": [round(k / args.steps, 5) for k in block_kern]},
            "checksum": check if check else {"sum_y": checksum},
            "csr_algorithmic_gbps": round(gbps, 1),
            "csr_algorithmic_gbps_note": "CSR-algorithmic bytes / time (the metric's definition, SURVEY 8(d)); NOT an HBM rate when the storage is "
                                         "smaller than CSR -- the bytes the kernel moves and its achieved rate are under roofline",
            "config": {"workload": "configs[%d]: 7-point 3D Poisson %d^3, N=%d rows, nnz=%d, fp64 values, int32 indices"
                                   % (2 if world == 1 else 3, n, N, nnz_total),
                       "format": storage, "rows_per_gpu": rows_rank,
                       "parallelism": "row-partitioned x%d" % world,
                       "vector_placement": {"y_minus_x_mod_64MiB_in_MiB": round(((y.data_ptr() - x.data_ptr()) % (64 << 20)) / float(1 << 20), 3),
                                            "rule": "vexhip_malloc: multiples of 64 MiB + 0 / 2 / 4 / 6 / 8 MiB (ops.device_vector)"}},
            "roofline": {"bound": "hbm",
                         "kernel": (("sell8_plane_kernel<%d, false, %d, false>" % (plane["tile"], {0: 2, 1: 18, 2: 17, 3: 0}[plane["store_policy"]])) if (storage == "sell8v" and plane) else
                                    KERNEL_OF["sell8v_grid"] if (storage == "sell8v" and grid_plan) else
                                    KERNEL_OF["sell8v_march"] if (storage == "sell8v" and march) else
                                    "sell8_pair_kernel<double, 7, true, true>" if (storage == "sell8v" and dict_blocks) else KERNEL_OF.get(storage, storage)),
                         "plane": plane,
                         "grid": grid_plan,
                         "stored_by_grid_line": direct,
                         "march": march,
                         "achieved": round(moved_rank / kern_s / 1e9, 1),
                         "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s",
                         "frac": round(moved_rank / kern_s / 1e9 / HBM_PEAK_GBPS, 4),
                         "dictionary_blocks": dict_blocks,
                         "bytes_per_launch": moved_rank,
                         "bytes_per_launch_what": "stored matrix (%d B: %s%s) + x once + y once" % (
                             matrix_bytes, storage, (", %d distinct 512-row code blocks + 4 B per slice" % dict_blocks) if dict_blocks else
                             (", by grid line: %d classes x 7 positions x %d value codes + 4 B per line" % (grid_plan["classes"], grid_plan["nx"])) if (direct and grid_plan) else ""),
                         "algorithmic_bytes_per_launch": alg_rank,
                         "algorithmic_gbps": round(alg_rank / kern_s / 1e9, 1),
                         "traffic": None,
                         "avg_launch_ms": round(kern_s * 1e3, 5)},
        }
        if setup is not None:
            out["setup"] = setup
        if world > 1:
            # north_star's second number: the fraction of HBM peak per GPU and over the N GPUs, by the bytes the ranks' kernels move
            # (stored matrix + x once + y once per rank; the ghost planes add 2 x 2 MB per rank and product) in the wall time of a step
            moved_all = world * matrix_bytes + 16 * N
            out["roofline"]["per_gpu"] = {"bytes_per_step": moved_rank, "achieved": round(moved_rank / per_step / 1e9, 1), "peak": HBM_PEAK_GBPS,
                                          "frac": round(moved_rank / per_step / 1e9 / HBM_PEAK_GBPS, 4), "what": "rank 0's bytes / wall time of a step (max over ranks)"}
            out["roofline"]["aggregate"] = {"bytes_per_step": moved_all, "achieved": round(moved_all / per_step / 1e9, 1), "peak": HBM_PEAK_GBPS * world,
                                            "frac": round(moved_all / per_step / 1e9 / (HBM_PEAK_GBPS * world), 4),
                                            "what": "all ranks' bytes / wall time of a step / (N x 8 TB/s): north_star's >= 0.50 at N = 8"}
        if storage == "sell8v" and (march or plane or grid_plan):
            if plane or grid_plan:
                # the plane product: per plane step a lane requests tile + 2 pairs of x for tile lines (centre lines + the two halo
                # lines; +-512 / +-P neighbours are its own earlier loads), 8 B per wave end and line for the +-1 neighbours, and
                # stores tile pairs.  HBM sees x and y once -- the traffic of a copy of x to y.
                tl = plane["tile"] if plane else 2
                l1_bytes = int((8 * (tl + 2) / tl + 8) * rows_rank)
                out["roofline"]["on_chip"] = {
                    "what": "bytes through L1 per launch: x %.0f B/row (tile %d: centre lines + 2 halo lines per %d) + y 8 B/row; HBM sees x and y once; "
                            "no LDS in the fast loop (profiles/r04_sq_summary_plane.txt, DESIGN.md 3.0c)" % (8 * (tl + 2) / tl, tl, tl),
                    "bytes_per_launch": l1_bytes, "achieved": round(l1_bytes / kern_s / 1e9, 1), "peak_l2": 34500.0, "unit": "GB/s",
                    "frac_of_l2": round(l1_bytes / kern_s / 1e9 / 34500.0, 4)}
            else:
                # the march product: the near diagonals' window of x comes once per slice (8 B/row + the overlap of a run's first
                # window), the two far diagonals come as two more coalesced streams (16 B/row), y is stored (8 B/row); codes are
                # decoded once per run.  HBM sees x and y once -- the traffic of a copy of x to y, timed here for comparison.
                l1_bytes = (8 + 16 + 8) * rows_rank
                out["roofline"]["on_chip"] = {
                    "what": "bytes through L1 per launch: window 8 B/row + far diagonals 16 B/row + y 8 B/row; HBM sees x and y once; "
                            "with four workgroups per CU the memory system is saturated (profiles/r03_sq_summary_march_v8.txt, DESIGN.md 3.0b)",
                    "bytes_per_launch": l1_bytes, "achieved": round(l1_bytes / kern_s / 1e9, 1), "peak_l2": 34500.0, "unit": "GB/s",
                    "frac_of_l2": round(l1_bytes / kern_s / 1e9 / 34500.0, 4)}
            try:
                yc = torch.empty_like(x)
                copies = {"device_copy": ("torch's copy of x to y on this device, same process: the same HBM traffic as the product (x once, y once)",
                                          lambda: yc.copy_(x)),
                          "device_copy_hand": ("vexhip_stream_copy_f64: x copied to y with one 16-byte pair per lane and non-temporal stores, same process -- "
                                               "the ceiling the memory system gives a kernel that moves these bytes and does nothing else",
                                               lambda: L.stream_copy_f64(local_rank, stream, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(yc.data_ptr()), x.numel()))}
                for name, (what, fn) in copies.items():
                    copy_ms = min(timed_events(torch, fn, 20) for _ in range(3))
                    out["roofline"][name] = {"what": what, "ms": round(copy_ms, 5), "gbps": round(2 * x.numel() * x.element_size() / copy_ms / 1e6, 1),
                                             "frac_of_8TBps": round(2 * x.numel() * x.element_size() / copy_ms / 1e6 / HBM_PEAK_GBPS, 4),
                                             "copy_over_product": round(copy_ms / 1e3 / kern_s, 4)}
                assert torch.equal(yc, x), "vexhip_stream_copy_f64 does not copy"
                out["roofline"]["frac_of_measured_copy"] = out["roofline"]["device_copy_hand"]["copy_over_product"]
                del yc
            except AssertionError:
                raise
            except Exception as e:  # noqa: BLE001 -- a comparison, not the measurement
                out["roofline"]["device_copy"] = {"error": str(e)[:200]}
        elif storage == "sell8v" and dict_blocks:
            # not HBM-bound any more: what the kernel pulls through L1 per product (ELL width 7: seven 16-byte x gathers per
            # lane and row pair = 56 B/row, 16 B/row of codes from the pooled blocks, 8 B/row stored), against the L2 rate
            l1_bytes = (56 + 16 + 8) * rows_rank
            out["roofline"]["on_chip"] = {
                "what": "bytes through L1 per launch: x gathers 56 B/row + pooled codes 16 B/row + y 8 B/row; HBM sees x and y once",
                "bytes_per_launch": l1_bytes, "achieved": round(l1_bytes / kern_s / 1e9, 1), "peak_l2": 34500.0, "unit": "GB/s",
                "frac_of_l2": round(l1_bytes / kern_s / 1e9 / 34500.0, 4),
                "peak_source": "MI355X_MICROARCH.md: L2 ~34.5 TB/s aggregate"}
        if sustained:
            out["sustained"] = sustained
        if not single:
            out["config"]["settle_steps"] = settle_steps
            out["config"]["exchange_bytes_per_rank"] = A.exchange_bytes()
            out["config"]["exchange_transport"] = transport
            out["config"]["backend"] = args.backend
            out["config"]["one_device_debug"] = bool(args.one_device)
            out["distributed"] = dist_extra
        if single and not args.no_secondary:
            sec = {}
            try:
                # ---- round 6: the product with a vector added in the same pass (vexhip_spmat_apply_axpby_f64): y = x + 2 A x takes x from the
                # registers that hold it (the bytes of y = A x), a residual r = b - A x reads b as well; both checked against the product
                if hasattr(A, "apply_axpby") and getattr(A, "handle", None) and A.dtype == torch.float64:
                    bvec = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 43)
                    r1 = torch.empty_like(y)
                    A.apply(x, y)
                    A.apply_axpby(x, r1, -1.0, bvec, 1.0)
                    same = bool(torch.equal(r1, bvec - y))
                    t_res = min(timed_events(torch, lambda: A.apply_axpby(x, r1, -1.0, bvec, 1.0), 40) for _ in range(2))
                    t_self = min(timed_events(torch, lambda: A.apply_axpby(x, r1, 2.0, x, 1.0), 40) for _ in range(2))
                    mb = int(A.info.matrix_bytes)
                    fused = bool(lib().spmat_axpby_fused(A.handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(bvec.data_ptr()), ctypes.c_void_p(r1.data_ptr())))
                    sec["SpMV + vector in one pass: r = b - A*x, Poisson %d^3 (vexhip_spmat_apply_axpby_f64)" % n] = {
                        "ms": round(t_res, 5), "one_pass": fused, "bits_equal_b_minus_product": same, "bytes_per_launch": mb + 24 * N,
                        "frac": round((mb + 24 * N) / t_res / 1e6 / HBM_PEAK_GBPS, 4),
                        "what": "x once + b once + r once + the stored matrix; the reference's two passes (r = b, then r -= A*x: vector.hpp:698-801) move 40 B per row"}
                    sec["SpMV + vector in one pass: y = x + 2 A*x, Poisson %d^3 (vexhip_spmat_apply_axpby_f64, the addend is x itself)" % n] = {
                        "ms": round(t_self, 5), "one_pass": fused, "bytes_per_launch": mb + 16 * N, "frac": round((mb + 16 * N) / t_self / 1e6 / HBM_PEAK_GBPS, 4),
                        "what": "x once + y once + the stored matrix: the addend comes from the registers that hold the centre lines"}
                    del bvec, r1
                # ---- kernels that stream what the metric counts: fp64 values + 32-bit columns, no compression
                del A
                torch.cuda.empty_cache()
                p2, c2, v2 = ops.poisson3d(n, dev)
                rcsr = []
                if storage == "sell8v" and plane:
                    # the SELL-512 storage with its slice dictionary (the set-up of rounds 2-3) and the march product of round 3 on it
                    torch.cuda.synchronize(); tc0 = time.perf_counter()
                    B = ops.SpMat(p2, c2, v2, fmt=args.format, plane=False)
                    torch.cuda.synchronize()
                    out["setup"]["sell512_setup_ms"] = round((time.perf_counter() - tc0) * 1e3, 3)
                    out["setup"]["sell512_setup_what"] = "the set-up of rounds 2-3 on the same arrays (VEXHIP_SPMAT_NO_PLANE: analysis, fill of 2.1 GB of per-slice codes, slice dictionary, march plan), second creation in this process"
                if storage == "sell8v" and plane and B.march:
                    tb = timed_events(torch, lambda: B.apply(x, y), 40)
                    assert abs(DistReductor("SUM_Kahan")(y) - checksum) <= 1e-10 * abs(checksum) + 1e-300
                    out["march_product"] = {"kernel": KERNEL_OF["sell8v_march"], "avg_launch_ms": round(tb, 5),
                                            "gflops": round(2.0 * nnz_total / tb / 1e6, 1),
                                            "what": "the same storage through the round-3 kernel (VEXHIP_SPMAT_NO_PLANE): x window in an LDS ring along runs of slices"}
                    del B
                if storage == "sell8v" and (march or plane):
                    # the pair product of round 2 on the same storage (the march / plane products replace it where they apply)
                    B = ops.SpMat(p2, c2, v2, fmt=args.format, march=False)
                    tb = timed_events(torch, lambda: B.apply(x, y), 40)
                    assert abs(DistReductor("SUM_Kahan")(y) - checksum) <= 1e-10 * abs(checksum) + 1e-300
                    out["pair_product"] = {"kernel": "sell8_pair_kernel<double, 7, true, true>", "avg_launch_ms": round(tb, 5),
                                           "gflops": round(2.0 * nnz_total / tb / 1e6, 1),
                                           "what": "the same storage through the round-2 kernel (VEXHIP_SPMAT_NO_MARCH): seven 16-byte gathers per lane and slice"}
                    del B
                if storage == "sell8v" and (dict_blocks or direct):
                    # the value-coded storage WITHOUT the slice dictionary: one code block per slice, streamed from HBM
                    B = ops.SpMat(p2, c2, v2, fmt=args.format, dictionary=False)
                    tb = timed_events(torch, lambda: B.apply(x, y), 40)
                    assert abs(DistReductor("SUM_Kahan")(y) - checksum) <= 1e-10 * abs(checksum) + 1e-300
                    mb = B.matrix_bytes() + 16 * N
                    out["value_codes_streamed"] = {
                        "kernel": "sell8_pair_kernel<double, 7, true, false>",
                        "what": "the same matrix with one code block per slice (VEXHIP_SPMAT_NO_DICTIONARY): 2 B per entry streamed from HBM",
                        "avg_launch_ms": round(tb, 5), "gflops": round(2.0 * nnz_total / tb / 1e6, 1),
                        "roofline": {"bound": "hbm", "achieved": round(mb / tb / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                     "frac": round(mb / tb / 1e6 / HBM_PEAK_GBPS, 4), "bytes_per_launch": mb}}
                    del B
                for f2, label in (("sell32", "SELL-512, 32-bit columns + fp64 values (vexhip_spmat format SELL)"),
                                  ("csr", "the CSR arrays themselves (row pointers + columns + values)")):
                    B = ops.SpMat(p2, c2, v2, fmt=f2)
                    tb = timed_events(torch, lambda: B.apply(x, y), 20)
                    assert abs(DistReductor("SUM_Kahan")(y) - checksum) <= 1e-10 * abs(checksum) + 1e-300
                    moved = B.matrix_bytes() + 16 * N          # what THIS kernel streams (sell_pair_kernel reads no row pointers: 13.45 GB, not the 13.85 GB of the CSR arrays)
                    rcsr.append({"kernel": KERNEL_OF[B.storage], "what": label, "avg_launch_ms": round(tb, 5),
                                 "gflops": round(2.0 * nnz_total / tb / 1e6, 1),
                                 "achieved": round(moved / tb / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": round(moved / tb / 1e6 / HBM_PEAK_GBPS, 4),
                                 "frac_of_algorithmic_bytes": round(alg_total / tb / 1e6 / HBM_PEAK_GBPS, 4),
                                 "bytes_per_launch": moved, "algorithmic_bytes_per_launch": alg_total})
                    del B
                out["roofline_csr"] = rcsr
                del p2, c2, v2
                torch.cuda.empty_cache()
                # ---- the general matrix: same pattern, a coefficient per face (~4 N distinct values), default SpMat
                p3, c3, v3 = ops.diffusion3d(n, dev)
                V = ops.SpMat(p3, c3, v3)
                del p3, c3, v3
                V.ptr = V.col = V.val = None
                tv = timed_events(torch, lambda: V.apply(x, y), 20)
                mv = V.matrix_bytes() + 16 * N
                out["variable_coefficient"] = {
                    "workload": "7-point -div(k grad u) on %d^3, k different on every face: ~4 distinct values per row of the %d entries (vexhip_diffusion3d_strip_f64_i32)" % (n, nnz_total),
                    "format": V.storage, "dictionary_blocks": V.dictionary_blocks,
                    "kernel": ("sell8_pair_kernel<double, 7, false, true>" if (V.storage == "sell8" and V.dictionary_blocks) else KERNEL_OF.get(V.storage, V.storage)),
                    "avg_launch_ms": round(tv, 5),
                    "gflops": round(2.0 * nnz_total / tv / 1e6, 1),
                    "roofline": {"bound": "hbm", "achieved": round(mv / tv / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": round(mv / tv / 1e6 / HBM_PEAK_GBPS, 4), "bytes_per_launch": mv,
                                 "algorithmic_bytes_per_launch": alg_total, "algorithmic_gbps": round(alg_total / tv / 1e6, 1)},
                    "sum_y": DistReductor("SUM_Kahan")(y)}
                # the general-matrix figure next to the headline (inside `roofline`, which the driver's summary keeps): the headline
                # product runs on a lossless re-coding of a constant-coefficient stencil; a matrix with a coefficient per face
                # streams diagonal codes + fp64 values, and the CSR-sized storages stream what the metric's name says
                out["value_general"] = out["variable_coefficient"]["gflops"]
                out["roofline"]["general_matrix"] = {
                    "what": "same 7-point pattern with a coefficient per face (%d^3): 1-byte diagonal codes + fp64 values streamed, default vex::SpMat" % n,
                    "gflops": out["variable_coefficient"]["gflops"], "avg_launch_ms": round(tv, 5),
                    "bytes_per_launch": mv, "achieved": round(mv / tv / 1e6, 1), "frac": round(mv / tv / 1e6 / HBM_PEAK_GBPS, 4)}
                out["roofline"]["frac_csr_bytes_kernel"] = max(r["frac"] for r in rcsr) if rcsr else None
                out["roofline"]["frac_csr_bytes_kernel_what"] = ("the kernels that stream CSR-sized bytes on the headline matrix (32-bit columns + fp64 values), each priced "
                                                                 "by the bytes IT moves: "
                                                                 + "; ".join("%s %.5f ms x %.3f GB = %.4f" % (r["kernel"].split("<")[0], r["avg_launch_ms"], r["bytes_per_launch"] / 1e9, r["frac"]) for r in rcsr))
                # the metric under its own name: GFLOP/s of the product that streams the CSR arrays themselves (row pointers +
                # 32-bit columns + fp64 values) on the headline matrix, beside `value` (the re-coded stencil) and `value_general`
                csr_row = [r for r in rcsr if r["kernel"].startswith("csr_stream2_kernel")]
                if csr_row:
                    out["value_csr_stream"] = csr_row[0]["gflops"]
                    out["roofline"]["csr_stream"] = {"kernel": csr_row[0]["kernel"], "avg_launch_ms": csr_row[0]["avg_launch_ms"], "gflops": csr_row[0]["gflops"],
                                                     "bytes_per_launch": csr_row[0]["bytes_per_launch"], "achieved": csr_row[0]["achieved"], "frac": csr_row[0]["frac"]}
                # multi-right-hand-side product on the general matrix
                xs = [ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 100 + k) for k in range(4)]
                ys = [torch.empty(N, dtype=torch.float64, device=dev) for _ in range(4)]
                t4 = timed_events(torch, lambda: V.apply_multi(xs, ys), 10)
                sec["SpMV 4 right-hand sides (SpMat * multivector<double,4>), variable coefficients %d^3" % n] = {
                    "ms": round(t4, 4), "gflops": round(8.0 * nnz_total / t4 / 1e6, 1)}
                del V, xs, ys
                torch.cuda.empty_cache()
                # ---- grids whose lines are not 512 points long (round 4, grid.hip): the same operator on 384^3 and 500^3
                for g in (384, 500):
                    Ng = g ** 3
                    pg, cg, vg = ops.poisson3d(g, dev)
                    G = ops.SpMat(pg, cg, vg)
                    xg = ops.fill_hash(torch.empty(Ng, dtype=torch.float64, device=dev), 7)
                    yg = torch.empty(Ng, dtype=torch.float64, device=dev)
                    tg = min(timed_events(torch, lambda: G.apply(xg, yg), 20) for _ in range(3))
                    row = {"storage": G.storage, "grid_plan": G.grid, "plane_plan": G.plane, "march_plan": G.march, "dictionary_blocks": G.dictionary_blocks,
                           "kernel": "sell8_grid_kernel" if G.grid else ("sell8_plane_kernel" if G.plane else "sell8_march_kernel" if G.march else "sell8_pair_kernel"),
                           "ms": round(tg, 5), "gflops": round(2.0 * cg.numel() / tg / 1e6, 1)}
                    mg = G.matrix_bytes() + 16 * Ng
                    row["roofline"] = {"bound": "hbm", "achieved": round(mg / tg / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                       "frac": round(mg / tg / 1e6 / HBM_PEAK_GBPS, 4), "bytes_per_launch": mg}
                    Bg = ops.SpMat(pg, cg, vg, march=False)
                    yb = torch.empty_like(yg)
                    tb = min(timed_events(torch, lambda: Bg.apply(xg, yb), 20) for _ in range(2))
                    row["pair_product_ms"] = round(tb, 5)
                    row["bit_identical_to_pair_product"] = bool(torch.equal(yg, yb))
                    assert row["bit_identical_to_pair_product"], "grid product differs from the pair product at %d^3" % g
                    # the same matrix in float (round 5, grid32.hip): bit-compared with the library's CSR loop
                    try:
                        v32 = vg.to(torch.float32); x32 = xg.to(torch.float32)
                        y32 = torch.empty_like(x32); yc32 = torch.empty_like(x32)
                        F = ops.SpMat(pg, cg, v32); C = ops.SpMat(pg, cg, v32, fmt="csr")
                        F.apply(x32, y32); C.apply(x32, yc32)
                        same32 = bool(torch.equal(y32, yc32))
                        assert same32, "fp32 grid product differs from the CSR loop at %d^3" % g
                        del C, yc32
                        tf = min(timed_events(torch, lambda: F.apply(x32, y32), 20) for _ in range(3))
                        mf = F.matrix_bytes() + 8 * Ng
                        row["fp32"] = {"ms": round(tf, 5), "gflops": round(2.0 * cg.numel() / tf / 1e6, 1), "grid_plan": F.grid is not None, "bit_identical_to_csr_loop": same32,
                                       "roofline": {"bound": "hbm", "achieved": round(mf / tf / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                                    "frac": round(mf / tf / 1e6 / HBM_PEAK_GBPS, 4), "bytes_per_launch": mf}}
                        del F, v32, x32, y32
                    except AssertionError:
                        raise
                    except Exception as e:  # noqa: BLE001 -- a secondary figure
                        row["fp32"] = {"error": repr(e)[:200]}
                    sec["SpMV Poisson 7-point %d^3 (y = A*x, vexhip_spmat)" % g] = row
                    del G, Bg, pg, cg, vg, xg, yg, yb
                    torch.cuda.empty_cache()
                # ---- matrices WITHOUT structure at size (round 5; the reference's tests are built on tests/random_matrix.hpp, its
                #      real-world caller feeds irregular matrices): default SpMat, checked against a gather + segmented sum in torch
                sec.update(unstructured_rows(torch, ops, dev, args))
                # ---- the C++ front end on the same two matrices (examples/spmv_headline.cpp)
                sec["C++ front end"] = cpp_rows(["spmv_headline", n, 50])
                sec.update(secondary_rows(torch, L, ops, dev, local_rank))
            except AssertionError:
                raise
            except Exception as e:                   # the headline must not depend on the secondary rows
                sec["error"] = repr(e)
            out["secondary"] = sec
        A = None
        if single and not args.no_pmc:
            torch.cuda.empty_cache()
            try:
                tr, how = measure_traffic(n)
            except Exception as e:
                tr, how = None, repr(e)
            out["roofline"]["traffic_source"] = how
            if tr:
                key = {"sell8v": "sell8_pair_kernel_vcoded", "sell8": "sell8_pair_kernel_values", "sell32": "sell_pair_kernel", "csr": "csr_stream2_kernel"}.get(storage)
                if storage == "sell8v" and march:
                    key = "sell8_march_kernel"
                if storage == "sell8v" and grid_plan:
                    key = "sell8_grid_kernel"
                if storage == "sell8v" and plane:
                    key = "sell8_plane_kernel"
                if key in tr:
                    out["roofline"]["traffic"] = tr[key]["total"]
                    out["roofline"]["traffic_read_written"] = [tr[key]["read"], tr[key]["written"]]
                    out["roofline"]["traffic_over_bytes_per_launch"] = round(tr[key]["total"] / float(moved_rank), 4)
                out["roofline"]["traffic_all_kernels"] = tr
        if single and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_grid, args.cpu_seconds)
    # N > 1: north_star's own multi-GPU shape -- ONE vex::Context driving all N GPUs through the C++ headers (vexcl/spmat.hpp
    # built from per-device strips, vexcl/exchange.hpp) -- timed by rank 0 in a child process while the ranks of this job wait
    # at a barrier with their devices idle; reported next to the one-process-per-GPU figure, never part of it.
    if world > 1 and not args.no_secondary:
        torch.cuda.synchronize()
        barrier()
        if rank == 0:
            progress("one vex::Context x%d through the C++ headers (examples/spmv_headline --devices %d)" % (world, world))
            extra_env = {"VEXCL_LOGICAL_DEVICES": str(world)} if args.one_device else {}
            try:
                rows = cpp_rows(["spmv_headline", n, 50, "--devices", 1 if args.one_device else world], env=extra_env)
            except Exception as e:  # noqa: BLE001 -- a comparison, not the measurement
                rows = [{"error": repr(e)[:300]}]
            out.setdefault("secondary", {})["C++ vex::Context x%d" % world] = rows
        barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
