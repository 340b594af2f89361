"""world_size-2/3 CPU tests of the row-partitioned product (vexcl_amd/distributed.py)
over the gloo backend: partitioning, local/remote split, ghost renumbering and the
point-to-point exchange plan are exercised for real; only the three device
kernels (local product, remote product, gather) are replaced by the CPU oracle,
injected from here -- the product itself has no CPU path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class OracleKernels:
    """Test double for vexcl_amd.distributed.DeviceKernels."""

    class Mat:
        def __init__(self, ptr, col, val, n_cols):
            self.ptr, self.col, self.val = (t.numpy() for t in (ptr, col, val))
            self.fmt = "csr"

        def apply(self, x, y, alpha, append):
            import oracle
            yn = y.numpy()
            oracle.spmv_csr(self.ptr, self.col, self.val, np.ascontiguousarray(x.numpy()), yn, alpha, append)
            return y

    def make_matrix(self, ptr, col, val, n_cols, fmt):
        return self.Mat(ptr, col, val, n_cols)

    def gather(self, idx, src, dst):
        dst.copy_(src[idx.long()])
        return dst


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from vexcl_amd.distributed import DistSpMat, partition
        if case == "poisson":
            n = 12
            ptr, col, val = oracle.poisson3d(n)
            N = M = n ** 3
        elif case == "upwind":
            # a ONE-SIDED coupling: the lower triangle of the Poisson matrix (entries at -n^2, -n, -1, 0 only).  Rank r has ghosts
            # from r - 1 and nobody has ghosts from r - 1's upper neighbour: the one-launch step's push / `sent` protocol would wait
            # for a flag nobody raises (advisor, round 5) -- _halo_plan must decline it on EVERY rank, by name
            n = 12
            ptr, col, val = oracle.poisson3d(n)
            N = M = n ** 3
            rows = np.repeat(np.arange(N), np.diff(ptr))
            keep = col <= rows
            cnt = np.bincount(rows[keep], minlength=N)
            ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(ptr.dtype)
            col, val = col[keep].copy(), val[keep].copy()
        elif case == "random_square":
            N = M = 1024
            ptr, col, val = oracle.random_matrix(7, N, M, 16)
        else:                                           # nonsquare with trailing empty rows
            N, M = 1000, 2048
            ptr, col, val = oracle.random_matrix(8, N, M, 16, empty_tail=300)
        x = oracle.random_f64(9, M)
        y0 = oracle.random_f64(10, N)
        want = y0.copy()
        oracle.spmv_csr(ptr, col, val, x, want, 1.5, True)
        bound = 1.5 * oracle.spmv_abs_bound(ptr, col, val, x) + np.abs(y0)

        part, cpart = partition(N, world), partition(M, world)
        assert part == oracle.partition(N, world) and cpart == oracle.partition(M, world)
        r0, r1 = part[rank], part[rank + 1]
        j0, j1 = int(ptr[r0]), int(ptr[r1])
        A = DistSpMat(torch.from_numpy((ptr[r0:r1 + 1] - ptr[r0]).astype(np.int32)),
                      torch.from_numpy(col[j0:j1].copy()), torch.from_numpy(val[j0:j1].copy()),
                      N, M, kernels=OracleKernels(), keep_strip=(case == "upwind"))
        if case == "upwind":
            try:
                A._halo_plan()
                declined = False
            except RuntimeError as e:
                declined = "not symmetric" in str(e)
            assert declined, "a one-sided coupling must be declined by the one-launch step's plan"
            assert A.enable_native(transport="halo") is False and A.native_error is not None
        # the split must agree with the oracle's restatement of spmat.hpp:291-378
        S = oracle.split_rows(ptr, col, val, M, world)["devs"][rank]
        assert np.array_equal(A.ghosts.numpy(), S["ghosts"])
        if A.rem is not None:
            assert np.array_equal(A.rem.col, S["rem"][1]) and np.array_equal(A.rem.ptr, S["rem"][0])
        if A.loc is not None:
            assert np.array_equal(A.loc.col, S["loc"][1]) and np.array_equal(A.loc.val, S["loc"][2])
        assert sum(A.recv_counts) == len(S["ghosts"])

        xs = torch.from_numpy(x[cpart[rank]:cpart[rank + 1]].copy())
        ys = torch.from_numpy(y0[r0:r1].copy())
        for _ in range(2):                              # buffers are reused across products
            ys.copy_(torch.from_numpy(y0[r0:r1]))
            A.apply(xs, ys, 1.5, True)
        err = np.abs(ys.numpy() - want[r0:r1])
        ok = bool(np.all(err <= 1e-10 * np.maximum(bound[r0:r1], 1e-300)))
        # Reductor final combine across ranks (all-reduce of one scalar per rank)
        from vexcl_amd.distributed import DistReductor

        class HostReductor:                             # test double for the device stage
            def __init__(self, f): self.f = f
            def device_result(self, t): return self.f(t).reshape(1)
        tot = DistReductor("SUM", local=HostReductor(torch.sum))(ys)
        ok_red = abs(tot - float(want.sum())) <= 1e-9 * float(np.abs(want).sum())
        mx = DistReductor("MAX", local=HostReductor(torch.max))(ys)
        ok_red = ok_red and abs(mx - float(want.max())) <= 1e-9 * float(np.abs(want).max())
        ys2 = torch.full_like(ys, 123.0)                # SET semantics overwrite
        A.apply(xs, ys2, 1.0, False)
        ok = ok and bool(np.allclose(ys2.numpy(), oracle.spmv_csr(ptr, col, val, x)[r0:r1], rtol=1e-12, atol=1e-12))
        # the C++ steps need the device kernels: on CPU every transport must decline ON EVERY RANK, together (each stage of
        # enable_native ends in an all-reduce), and leave the torch.distributed step working
        for tr in ("halo", "ipc"):
            ok = ok and A.enable_native(transport=tr) is False and A.native_error is not None and A.native_status() is None
        ys3 = torch.full_like(ys, -5.0)
        A.apply(xs, ys3, 1.0, False)
        ok = ok and bool(torch.equal(ys3, ys2))
        if case == "poisson":
            # the strip stored WITH its ghost planes (transport "halo", distributed.halo_extended_csr): its product with
            # [lower ghost plane | x | upper ghost plane] gives this rank's rows of the global product, for every rank of every world
            from vexcl_amd.distributed import halo_extended_csr
            P = n * n
            has_lo, has_hi = rank > 0, rank < world - 1
            lo, hi = (P if has_lo else 0), (P if has_hi else 0)
            if (r1 - r0) % P == 0:
                pe, ce = halo_extended_csr(torch.from_numpy((ptr[r0:r1 + 1] - ptr[r0]).astype(np.int32)), torch.from_numpy(col[j0:j1].copy()), r0, r1 - r0, lo, hi)
                xe = x[r0 - lo:r1 + hi]
                ye = oracle.spmv_csr(pe.numpy(), ce.numpy(), val[j0:j1].copy(), np.ascontiguousarray(xe))
                ok = ok and len(ye) == lo + (r1 - r0) + hi and np.array_equal(ye[lo:lo + r1 - r0], oracle.spmv_csr(ptr, col, val, x)[r0:r1])
                ok = ok and not ye[:lo].any() and not ye[lo + r1 - r0:].any()
        out[rank] = 1 if (ok and ok_red) else 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "poisson"), (2, "random_square"), (3, "nonsquare"), (3, "poisson"), (4, "random_square"),
                                        (8, "poisson"), (3, "upwind")])
def test_distributed_spmv_gloo(world, case, oracle):
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0] * world)
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert list(out) == [1] * world


def _scan_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from vexcl_amd.distributed import DistScan, partition

        class HostScan:                                 # test double for the device scan of one segment
            def inclusive_scan(self, inp, out):
                out.copy_(torch.cumsum(inp, 0).to(inp.dtype)); return out
            def exclusive_scan(self, inp, out, init):
                c = torch.cumsum(inp, 0).to(inp.dtype)
                res = torch.empty_like(inp)
                if inp.numel():
                    res[0] = init
                    res[1:] = c[:-1] + init
                out.copy_(res); return out

        ok = True
        for n in (0, 5, 1000, 4099):
            rng = np.random.default_rng(50 + n)
            x = rng.integers(-2**31, 2**31 - 1, size=n, dtype=np.int64).astype(np.int32)     # wraps mod 2^32
            part = partition(n, world)
            seg = torch.from_numpy(x[part[rank]:part[rank + 1]].copy())
            with np.errstate(over="ignore"):
                incl = np.cumsum(x.astype(np.int64)).astype(np.int32)
                excl = (np.concatenate([[0], np.cumsum(x.astype(np.int64))[:-1]]) + 7).astype(np.int32) if n else incl
            scan = DistScan(local=HostScan())
            got = scan(seg.clone())
            ok = ok and np.array_equal(got.numpy(), incl[part[rank]:part[rank + 1]])
            buf = seg.clone()
            got = scan(buf, buf, exclusive=True, init=7)                # in place
            ok = ok and np.array_equal(got.numpy(), excl[part[rank]:part[rank + 1]])
            xf = oracle.random_f64(60 + n, n)
            segf = torch.from_numpy(xf[part[rank]:part[rank + 1]].copy())
            gotf = scan(segf, exclusive=False)
            ok = ok and np.allclose(gotf.numpy(), np.cumsum(xf)[part[rank]:part[rank + 1]], rtol=1e-12, atol=1e-12)
        out[rank] = 1 if ok else 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_scan_gloo(world, oracle):
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0] * world)
    port = _free_port()
    procs = [ctx.Process(target=_scan_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert list(out) == [1] * world


def test_device_kernels_refuse_cpu_tensors(built_lib):
    from vexcl_amd import ops, Error
    t = torch.zeros(4, dtype=torch.float64)
    with pytest.raises(Error):
        ops.Reductor("SUM")(t)
    with pytest.raises(Error):
        ops.spmv_csr(torch.zeros(5, dtype=torch.int32), torch.zeros(0, dtype=torch.int32), t[:0], t, t)
