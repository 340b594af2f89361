"""Two ranks on ONE GPU (the gpurun box has one): the real device kernels behind
vexcl_amd.distributed -- strip generation, local / remote split, SELL8 local part,
pack kernel, remote CSR part, DistReductor, DistScan -- with the exchange carried by
gloo (RCCL refuses two ranks on one device).  Only the transport differs from the
one-process-per-GPU job bench.py launches."""
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
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, out, ipc=True, by_line=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if by_line:            # the ranks' local parts stored by grid line (the default above 2^23 rows per rank; forced at this size)
        os.environ["VEXHIP_PLANE_FORCE"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vexcl_amd import ops, lib
        from vexcl_amd.distributed import DistReductor, DistScan, DistSpMat, partition
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        L = lib()
        N = n ** 3
        part = partition(N, world)
        r0, r1 = part[rank], part[rank + 1]
        ptr, col, val = ops.poisson3d(n, dev, rows=(r0, r1))
        x = ops.fill_hash(torch.empty(r1 - r0, dtype=torch.float64, device=dev), (42 + r0 * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        y = torch.full((r1 - r0,), 7.0, dtype=torch.float64, device=dev)
        A = DistSpMat(ptr, col, val, N, N)
        for _ in range(2):                                  # exchange buffers are reused
            y.fill_(7.0)
            A.apply(x, y, 1.5, True)
        # the same product on the whole matrix, one "device"
        fp, fc, fv = ops.poisson3d(n, dev)
        fx = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42)
        fy = torch.full((N,), 7.0, dtype=torch.float64, device=dev)
        ops.SpMat(fp, fc, fv).apply(fx, fy, 1.5, True)
        ok = bool(torch.equal(fx[r0:r1], x))                 # x is a function of the global index
        scale = float(fv.abs().max()) * 8
        ok = ok and bool(((y - fy[r0:r1]).abs() <= 1e-12 * scale).all())
        ok = ok and A.loc is not None and (world == 1 or A.rem is not None)
        ok = ok and A.loc.fmt == "sell" and A.loc.hell.deltas is not None      # banded local part: 1-byte diagonal codes
        ok = ok and bool(A.loc.direct) == bool(by_line) and (not by_line or A.loc.grid is not None)
        tot = DistReductor("SUM")(y)
        ok = ok and abs(tot - float(fy.sum())) <= 1e-9 * float(fy.abs().sum())
        mx = DistReductor("MAX")(y)
        ok = ok and mx == float(fy.max())
        k = ops.fill_hash(torch.empty(r1 - r0, dtype=torch.int32, device=dev), (5 + r0 * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        fk = ops.fill_hash(torch.empty(N, dtype=torch.int32, device=dev), 5)
        got = DistScan()(k.clone())
        want = ops.inclusive_scan(fk.clone())
        ok = ok and bool(torch.equal(got, want[r0:r1]))
        got = DistScan()(k.clone(), exclusive=True, init=9)
        want = ops.exclusive_scan(fk.clone(), init=9)
        ok = ok and bool(torch.equal(got, want[r0:r1]))
        # the product step issued from C++ over peer-mapped ghost windows (IPC transport): every rank maps its neighbours'
        # windows (hipIpcGetMemHandle / hipIpcOpenMemHandle between the processes that share this GPU) -- 60 products with
        # a changing x, each against the one-device product, then back to the torch.distributed transport
        if ipc:
            assert A.enable_native(transport="ipc"), A.native_error
            assert A.native_status()["transport"] == "ipc"
            for k in range(60):
                xk = x * (1.0 + k)
                y.fill_(-3.0)
                A.apply(xk, y, 1.0, False)
                if k % 20 == 0:
                    fy.zero_()
                    ops.SpMat(fp, fc, fv).apply(fx * (1.0 + k), fy, 1.0, False)
                    torch.cuda.synchronize()
                    ok = ok and bool(((y - fy[r0:r1]).abs() <= 1e-12 * scale * (1.0 + k)).all())
            torch.cuda.synchronize()
            st = A.native_status()
            ok = ok and st["timed_out"] == 0
            prof = A.profile_step(x, y)
            ok = ok and prof["total"] > 0
            # the step over the IPC windows and the step over the torch.distributed transport (the one RCCL carries on a node
            # with one GPU per rank) must give the SAME bits: same local and remote kernels, only the ghosts travel differently
            y_ipc = torch.full((r1 - r0,), 7.0, dtype=torch.float64, device=dev)
            A.apply(x, y_ipc, 1.5, True)
            torch.cuda.synchronize()
            dist.barrier()
            A.disable_native()
            y.fill_(7.0)
            A.apply(x, y, 1.5, True)
            ok = ok and bool(torch.equal(y, y_ipc))
            ref = torch.full((N,), 7.0, dtype=torch.float64, device=dev)
            ops.SpMat(fp, fc, fv).apply(fx, ref, 1.5, True)
            ok = ok and bool(((y - ref[r0:r1]).abs() <= 1e-12 * scale).all())
        out[rank] = 1 if ok else 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,by_line", [(2, 64, False), (3, 48, False), (2, 64, True)])
def test_two_ranks_share_one_gpu(world, n, by_line, built_lib):
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0] * world)
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, out, True, by_line)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert list(out) == [1] * world


def _stencil_strip(torch, nx, ny, nz, r0, r1, dev):
    """Rows [r0, r1) of the 7-point operator on an nx x ny x nz grid (identity rows on the boundary, as the generator of
    examples/benchmark.cpp:364-415 writes them), GLOBAL columns, int32 CSR on the device."""
    r = torch.arange(r0, r1, device=dev, dtype=torch.int64)
    P = nx * ny
    ix, iy, iz = r % nx, (r // nx) % ny, r // P
    inner = (ix > 0) & (ix < nx - 1) & (iy > 0) & (iy < ny - 1) & (iz > 0) & (iz < nz - 1)
    cnt = torch.where(inner, 7, 1)
    ptr = torch.zeros(r1 - r0 + 1, dtype=torch.int64, device=dev)
    torch.cumsum(cnt, 0, out=ptr[1:])
    nnz = int(ptr[-1])
    col = torch.empty(nnz, dtype=torch.int64, device=dev)
    val = torch.empty(nnz, dtype=torch.float64, device=dev)
    h2i = float((nx - 1) ** 2)
    b = ptr[:-1]
    bi, ri = b[inner], r[inner]
    for k, (d, v) in enumerate(((-P, -h2i), (-nx, -h2i), (-1, -h2i), (0, 6 * h2i), (1, -h2i), (nx, -h2i), (P, -h2i))):
        col[bi + k] = ri + d
        val[bi + k] = v
    bo, ro = b[~inner], r[~inner]
    col[bo] = ro
    val[bo] = 1.0
    return ptr.to(torch.int32), col.to(torch.int32), val


def _stencil_strip_2d(torch, W, Hrows, r0, r1, dev):
    """Rows [r0, r1) of the 5-point operator on a W x Hrows grid (identity rows on the boundary), GLOBAL columns, int32 CSR on the device."""
    r = torch.arange(r0, r1, device=dev, dtype=torch.int64)
    ix, iz = r % W, r // W
    inner = (ix > 0) & (ix < W - 1) & (iz > 0) & (iz < Hrows - 1)
    cnt = torch.where(inner, 5, 1)
    ptr = torch.zeros(r1 - r0 + 1, dtype=torch.int64, device=dev)
    torch.cumsum(cnt, 0, out=ptr[1:])
    nnz = int(ptr[-1])
    col = torch.empty(nnz, dtype=torch.int64, device=dev)
    val = torch.empty(nnz, dtype=torch.float64, device=dev)
    h2i = float((W - 1) ** 2)
    b = ptr[:-1]
    bi, ri = b[inner], r[inner]
    for k, (d, v) in enumerate(((-W, -h2i), (-1, -h2i), (0, 4 * h2i), (1, -h2i), (W, -h2i))):
        col[bi + k] = ri + d
        val[bi + k] = v
    bo, ro = b[~inner], r[~inner]
    col[bo] = ro
    val[bo] = 1.0
    return ptr.to(torch.int32), col.to(torch.int32), val


def _halo_worker(rank, world, port, planes_per_rank, ny, out, transport="halo", variable=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VEXHIP_PLANE_FORCE="1", VEXHIP_IPC_TIMEOUT_MS="20000")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vexcl_amd import ops
        from vexcl_amd.distributed import DistSpMat, partition
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        nx = 512
        nz = planes_per_rank * world
        two_d = ny < 0                 # ny = -W: a 5-point operator on a 2-D grid with rows of W points, `planes_per_rank` rows each
        if two_d:
            nx, ny = -ny, 1
        N = nx * ny * nz
        part = partition(N, world)
        r0, r1 = part[rank], part[rank + 1]
        assert (r1 - r0) == planes_per_rank * nx * ny
        if two_d:
            ptr, col, val = _stencil_strip_2d(torch, nx, nz, r0, r1, dev)
            fp, fc, fv = _stencil_strip_2d(torch, nx, nz, 0, N, dev)
        else:
            ptr, col, val = _stencil_strip(torch, nx, ny, nz, r0, r1, dev)
            fp, fc, fv = _stencil_strip(torch, nx, ny, nz, 0, N, dev)
        if variable:           # a different value in every entry: nothing to code -- SELL-512 with diagonal codes and stored values (the pair product's role)
            fv = fv * (1.0 + 1e-3 * ops.fill_hash(torch.empty_like(fv), 5))
            val = fv[int(fp[r0]):int(fp[r1])].clone()
        A = DistSpMat(ptr, col, val, N, N, keep_strip=True)
        # the whole matrix on one "device": the bits the N-rank product must reproduce (the stored strip keeps a row's entries in
        # column order, ghost columns included -- unlike the split step, which adds the remote entries last)
        F = ops.SpMat(fp, fc, fv)
        fx = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 42)
        x = fx[r0:r1].clone()
        ok = A.enable_native(transport=transport)
        assert ok, A.native_error
        st = A.native_status()
        ok = ok and st["transport"] == transport
        if variable:
            ok = ok and A._ext.storage == "sell8" and A._ext.plane is None and A._ext.grid is None
        if two_d:                      # the rows cut into virtual lines, a flat plan (grid.hip): the ghost range is one row of the grid
            ok = ok and A._ext.grid is not None and A._ext.grid["flat"] == 1 and A._ext.grid["nx"] * A._ext.grid["lines_per_plane"] == nx
        y = torch.empty(r1 - r0, dtype=torch.float64, device=dev)
        fy = torch.empty(N, dtype=torch.float64, device=dev)
        xk = torch.empty_like(x)                             # ONE vector rewritten between products (pull: mapped by the neighbours once)
        for k in range(40):                                  # back to back, x changing: flags, ghost planes and step numbers are reused
            torch.mul(x, 1.0 + k, out=xk)
            y.fill_(-3.0)
            A.apply(xk, y, 1.0, False)
            if k % 13 == 0:
                F.apply(fx * (1.0 + k), fy, 1.0, False)
                torch.cuda.synchronize()
                ok = ok and bool(torch.equal(y, fy[r0:r1]))
        # y += alpha A x
        y.fill_(7.0); fy.fill_(7.0)
        A.apply(x, y, 1.5, True)
        F.apply(fx, fy, 1.5, True)
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(y, fy[r0:r1]))
        ok = ok and A.native_status()["timed_out"] == 0
        # against the split step over the torch.distributed transport: the same numbers up to the order of a boundary row's sum
        dist.barrier()
        A.disable_native()
        y2 = torch.full((r1 - r0,), 7.0, dtype=torch.float64, device=dev)
        A.apply(x, y2, 1.5, True)
        scale = float(fv.abs().max()) * 8
        ok = ok and bool(((y - y2).abs() <= 1e-12 * scale).all())
        out[rank] = 1 if ok else 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,planes,ny", [(2, 8, 512), (3, 8, 128), (2, 3, 128), (3, 5, 64), (2, 21, 64)])
def test_one_launch_step_reads_ghost_planes_from_the_window(world, planes, ny, built_lib):
    """Transport "halo" (round 5, csrc/halo.hpp; reference: the five phases of vexcl/spmat.hpp:120-185): every rank stores its strip
    with the two ghost planes, ONE plane-product launch pushes the boundary planes and reads the neighbours' from the peer-mapped
    window.  Ranks share the GPU; 512 x ny x (8 per rank) grid; 40 products back to back.  The result must have the BITS of the
    one-device product of the whole matrix.
    (Three ranks on ONE GPU run 128 lines per plane: the workgroups next to a ghost plane wait for the neighbour's launch while they
    hold their CU slots -- 256 of them per ghost plane at 512 lines -- and the launches of three processes share the 768 slots of
    the one device: a middle rank's 512 waiting workgroups and 256 of a neighbour that is one product ahead can leave no slot for
    the third rank's push, and all of them sit there until the time-out.  With a GPU per rank a launch only ever waits for
    launches on OTHER devices; the stand-in is kept below the device's capacity.)
    Strips of 3 and 5 planes are shorter than the chunks next to the ghost planes (a middle rank reads BOTH ghost planes from
    one chunk); 21 planes leave a main chunk of five between them."""
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0] * world)
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, planes, ny, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert list(out) == [1] * world


@pytest.mark.parametrize("world,planes,ny,variable", [(2, 8, 128, False), (2, 8, 128, True), (2, 3, 64, True), (2, 40, -2000, False), (2, 8, -3000, False)])
def test_one_launch_step_reads_the_neighbours_x_in_place(world, planes, ny, variable, built_lib):
    """Transport "pull" (round 6) BETWEEN PROCESSES: nothing is pushed -- every rank maps the allocation that holds its neighbours' x
    (vexhip_ipc_export / _open) and the product launch reads their boundary planes where they lie, behind "x is final" flags.  For the
    plane product, -- variable -- for a strip stored with diagonal codes and a value per entry (the pair product's role,
    csrc/sell8.hip): any banded operator, and -- ny = -W -- for the 5-point operator on a 2-D grid with rows of W points (virtual lines,
    a flat grid plan: the ghost range is ONE row).  40 products back to back with x rewritten in between; the BITS of the one-device product."""
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0] * world)
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, planes, ny, out, "pull", variable)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert list(out) == [1] * world
