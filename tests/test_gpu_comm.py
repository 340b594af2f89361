"""The RCCL / peer transport behind the C ABI (include/vexhip.h vexhip_comm_*, vexhip_halo_exchange,
vexhip_allreduce_scalar, vexhip_allgather, vexhip_dist_spmv_*) on the one GPU of the gpurun box:

* RCCL itself with a ONE-rank communicator: init from a unique id, all-reduce, all-gather, and ncclSend/ncclRecv of the
  rank to itself (VEXHIP_RCCL_SELF forces the send/recv pair instead of the device copy) -- proves loading, linking and
  the call sequence; more than one rank per GPU is refused by RCCL, so
* the bookkeeping of the exchange (offsets, counts, event ordering) is exercised with 2 and 3 LOGICAL devices on the one
  GPU through the PEER transport -- the same vexhip_halo_exchange entry point the C++ vex::SpMat calls;
* one rank's product step issued from C++ (with and without hipGraph replay) against the one-device product.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

_SELF = r'''
import ctypes, sys, torch
sys.path.insert(0, %r)
from vexcl_amd import lib, _capi
L = lib(); dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
raw = (ctypes.c_char * 128)()
L.comm_unique_id(ctypes.cast(raw, ctypes.c_void_p))
comm = ctypes.c_void_p()
L.comm_init_rank(0, 0, 1, ctypes.cast(raw, ctypes.c_void_p), ctypes.byref(comm))
w, nl, tr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
L.comm_size(comm, ctypes.byref(w), ctypes.byref(nl), ctypes.byref(tr))
assert (w.value, nl.value, tr.value) == (1, 1, 1), (w.value, nl.value, tr.value)          # transport 1 = RCCL
s = torch.cuda.Stream(); st = (ctypes.c_void_p * 1)(s.cuda_stream)
# ncclSend / ncclRecv of the rank to itself
a = torch.arange(1000, dtype=torch.float64, device=dev) * 0.5; b = torch.zeros_like(a)
sb = (ctypes.c_void_p * 1)(a.data_ptr()); rb = (ctypes.c_void_p * 1)(b.data_ptr())
cnt = (ctypes.c_int64 * 1)(1000)
torch.cuda.synchronize()
L.halo_exchange(comm, _capi.F64, sb, cnt, rb, cnt, st)
s.synchronize()
assert torch.equal(a, b)
# all-reduce and all-gather on one rank
v = torch.tensor([3.25, -1.0], dtype=torch.float64, device=dev)
L.allreduce_scalar(comm, _capi.SUM, _capi.F64, (ctypes.c_void_p * 1)(v.data_ptr()), 2, st); s.synchronize()
assert v.tolist() == [3.25, -1.0]
g = torch.zeros(2, dtype=torch.float64, device=dev)
L.allgather(comm, _capi.F64, (ctypes.c_void_p * 1)(v.data_ptr()), (ctypes.c_void_p * 1)(g.data_ptr()), 2, st); s.synchronize()
assert g.tolist() == [3.25, -1.0]
# one rank's product step with a REAL exchange (the rank's ghosts are elements of its own x, sent to itself): the share as one run
# of consecutive elements (sent straight out of x) and as a permutation (packed by the gather kernel first)
from vexcl_amd import ops
rows, ng = 6000, 1500
ptr = torch.arange(rows + 1, dtype=torch.int32, device=dev); col = torch.arange(rows, dtype=torch.int32, device=dev)
loc = ops.SpMat(ptr, col, torch.full((rows,), 2.0, dtype=torch.float64, device=dev))                      # local part: 2 I
rw = torch.arange(ng, dtype=torch.int32, device=dev) * 3                                                   # rows 0, 3, 6, ... reach one ghost each
cp = torch.arange(ng + 1, dtype=torch.int32, device=dev); rc = torch.arange(ng, dtype=torch.int32, device=dev)
rv = torch.full((ng,), 3.0, dtype=torch.float64, device=dev)
x = torch.rand(rows, dtype=torch.float64, device=dev)
cnts = (ctypes.c_int64 * 1)(ng)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for label, sidx in (("run", torch.arange(ng, device=dev) + 777), ("perm", torch.randperm(rows, device=dev)[:ng])):
    sidx = sidx.to(torch.int32).contiguous()
    sbuf = torch.zeros(ng, dtype=torch.float64, device=dev); gbuf = torch.zeros(ng, dtype=torch.float64, device=dev)
    step = ctypes.c_void_p()
    L.dist_spmv_create(comm, _capi.F64, rows, loc.handle, ng, p(rw), p(cp), p(rc), p(rv), ng, p(sidx), p(sbuf), cnts, ng, p(gbuf), cnts, ctypes.byref(step))
    want = 2.0 * x
    want[rw.long()] += 3.0 * x[sidx.long()]
    y = torch.empty(rows, dtype=torch.float64, device=dev)
    for _ in range(3):
        L.dist_spmv_apply(step, ctypes.c_void_p(s.cuda_stream), 1.0, 0, p(x), p(y))
    s.synchronize()
    assert torch.equal(y, want), label
    assert (float(sbuf.abs().sum()) == 0.0) == (label == "run"), label      # the run never touches the pack buffer
    L.dist_spmv_destroy(step)
L.comm_destroy(comm)
# the same step over a peer-mapped ghost WINDOW (IPC transport): the rank writes its own share into its own window, waits on its
# own flags and releases them -- the whole protocol with one participant; 40 products: the step counters gate every reuse
win = ctypes.c_void_p(); L.ipc_window_create(0, 0, 1, ng * 8, ctypes.byref(win))
hnd = (ctypes.c_char * 64)(); L.ipc_window_export(win, ctypes.cast(hnd, ctypes.c_void_p))
assert any(b != 0 for b in hnd.raw)
zero = (ctypes.c_int64 * 1)(0)
for label, sidx in (("run", torch.arange(ng, device=dev) + 777), ("perm", torch.randperm(rows, device=dev)[:ng])):
    sidx = sidx.to(torch.int32).contiguous()
    step = ctypes.c_void_p()
    L.dist_spmv_create_ipc(win, _capi.F64, rows, loc.handle, ng, p(rw), p(cp), p(rc), p(rv), ng, p(sidx), cnts, zero, ng, cnts, ctypes.byref(step))
    y = torch.empty(rows, dtype=torch.float64, device=dev)
    for k in range(40):
        xk = x * (k + 1)
        L.dist_spmv_apply(step, ctypes.c_void_p(s.cuda_stream), 1.0, 0, p(xk), p(y))
        s.synchronize()
        want = 2.0 * xk
        want[rw.long()] += 3.0 * xk[sidx.long()]
        assert torch.equal(y, want), (label, k)
    to, tr, di = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.dist_spmv_status(step, ctypes.byref(to), ctypes.byref(tr), ctypes.byref(di))
    assert (to.value, tr.value, di.value) == (0, 3, 1 if label == "run" else 0), (to.value, tr.value, di.value)
    ms = (ctypes.c_float * 6)()
    L.dist_spmv_profile(step, ctypes.c_void_p(s.cuda_stream), 1.0, 0, p(x), p(y), ms)
    assert ms[0] > 0 and all(v >= 0 for v in ms), list(ms)
    L.dist_spmv_destroy(step)
# two LIVE plans on one window, used alternately (AMG levels, A and A^T): each plan counts its own launches, the flags carry
# the window's step numbers (round 4; the round-3 form never raised `arrive` here and ran into its timeout)
plans = []
for sidx in (torch.arange(ng, device=dev) + 777, torch.randperm(rows, device=dev)[:ng]):
    sidx = sidx.to(torch.int32).contiguous()
    step = ctypes.c_void_p()
    L.dist_spmv_create_ipc(win, _capi.F64, rows, loc.handle, ng, p(rw), p(cp), p(rc), p(rv), ng, p(sidx), cnts, zero, ng, cnts, ctypes.byref(step))
    plans.append((step, sidx))
y = torch.empty(rows, dtype=torch.float64, device=dev)
for k in range(12):
    step, sidx = plans[k %% 2] if k %% 5 else plans[0]
    xk = x * (k + 1)
    L.dist_spmv_apply(step, ctypes.c_void_p(s.cuda_stream), 1.0, 0, p(xk), p(y))
    s.synchronize()
    want = 2.0 * xk
    want[rw.long()] += 3.0 * xk[sidx.long()]
    assert torch.equal(y, want), ("two plans", k)
# the IPC step replayed from a hipGraph (its step numbers live in device memory): x changes IN PLACE between the replays
step, sidx = plans[1]
L.dist_spmv_set_graph(step, 1)
xg = x.clone()
for k in range(25):
    with torch.cuda.stream(s):
        xg.mul_(1.0 + 1.0 / (k + 1))
    L.dist_spmv_apply(step, ctypes.c_void_p(s.cuda_stream), 1.0, 0, p(xg), p(y))
    s.synchronize()
    want = 2.0 * xg
    want[rw.long()] += 3.0 * xg[sidx.long()]
    assert torch.equal(y, want), ("graph", k)
L.dist_spmv_set_graph(step, 0)
L.dist_spmv_apply(plans[0][0], ctypes.c_void_p(s.cuda_stream), 1.0, 0, p(x), p(y))      # and the other plan still works
s.synchronize()
want = 2.0 * x
want[rw.long()] += 3.0 * x[plans[0][1].long()]
assert torch.equal(y, want)
for step, _ in plans:
    to = ctypes.c_int()
    L.dist_spmv_status(step, ctypes.byref(to), None, None)
    assert to.value == 0
    L.dist_spmv_destroy(step)
L.ipc_window_destroy(win)
print("rccl self ok")
'''


def test_rccl_one_rank_self_exchange():
    env = dict(os.environ, VEXHIP_RCCL_SELF="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", _SELF % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "rccl self ok" in out, out[-3000:]


@pytest.mark.parametrize("nd", [2, 3])
def test_peer_transport_logical_devices(nd):
    """vexhip_halo_exchange / allreduce / allgather over nd logical devices of the one GPU (transport AUTO -> PEER)."""
    from vexcl_amd import lib, _capi
    L = lib(); dev = torch.device("cuda:0")
    devs = (ctypes.c_int * nd)(*([0] * nd))
    comm = ctypes.c_void_p()
    L.comm_init(nd, devs, _capi_auto(), ctypes.byref(comm))
    tr = ctypes.c_int()
    L.comm_size(comm, None, None, ctypes.byref(tr))
    assert tr.value == 2                                    # PEER: the devices are not distinct GPUs
    rng = np.random.default_rng(nd)
    counts = rng.integers(0, 50, size=(nd, nd)); counts[0, nd - 1] = 0          # counts[o][d]: o sends to d
    send = [torch.tensor(rng.random(int(counts[o].sum())), dtype=torch.float64, device=dev) for o in range(nd)]
    recv = [torch.full((int(counts[:, d].sum()) + 1,), -1.0, dtype=torch.float64, device=dev) for d in range(nd)]
    streams = [torch.cuda.Stream() for _ in range(nd)]
    sc = (ctypes.c_int64 * (nd * nd))(*[int(counts[o, d]) for o in range(nd) for d in range(nd)])
    rc = (ctypes.c_int64 * (nd * nd))(*[int(counts[o, d]) for d in range(nd) for o in range(nd)])
    sb = (ctypes.c_void_p * nd)(*[t.data_ptr() if t.numel() else None for t in send])
    rb = (ctypes.c_void_p * nd)(*[t.data_ptr() for t in recv])
    st = (ctypes.c_void_p * nd)(*[s.cuda_stream for s in streams])
    torch.cuda.synchronize()
    for _ in range(3):                                      # buffers are reused: ordering against the previous round
        L.halo_exchange(comm, _capi.F64, sb, sc, rb, rc, st)
    torch.cuda.synchronize()
    for d in range(nd):
        want = torch.cat([send[o][int(counts[o, :d].sum()):int(counts[o, :d + 1].sum())] for o in range(nd)] + [torch.tensor([-1.0], dtype=torch.float64, device=dev)])
        assert torch.equal(recv[d], want), d
    # reductions and gathers
    for op, fn in ((_capi.SUM, sum), (_capi.MIN, min), (_capi.MAX, max)):
        vals = [torch.tensor([1.5 * (d + 1), -2.0 * d], dtype=torch.float64, device=dev) for d in range(nd)]
        want = [fn(float(v[k]) for v in vals) for k in range(2)]
        torch.cuda.synchronize()
        L.allreduce_scalar(comm, op, _capi.F64, (ctypes.c_void_p * nd)(*[v.data_ptr() for v in vals]), 2, st)
        torch.cuda.synchronize()
        for v in vals:
            assert v.tolist() == want
    parts = [torch.tensor([d, 10 * d], dtype=torch.int32, device=dev) for d in range(nd)]
    outs = [torch.zeros(2 * nd, dtype=torch.int32, device=dev) for _ in range(nd)]
    torch.cuda.synchronize()
    L.allgather(comm, _capi.I32, (ctypes.c_void_p * nd)(*[p.data_ptr() for p in parts]), (ctypes.c_void_p * nd)(*[o.data_ptr() for o in outs]), 2, st)
    torch.cuda.synchronize()
    for o in outs:
        assert o.tolist() == [v for d in range(nd) for v in (d, 10 * d)]
    L.comm_destroy(comm)


def _capi_auto():
    return 0            # VEXHIP_COMM_AUTO


@pytest.mark.parametrize("graph", [False, True])
def test_native_step_one_rank(graph):
    """DistSpMat with the product step issued from C++ (vexhip_dist_spmv_apply), one rank: == the one-device product."""
    from vexcl_amd import ops
    from vexcl_amd.distributed import DistSpMat
    dev = torch.device("cuda:0")
    n = 48
    N = n ** 3
    ptr, col, val = ops.diffusion3d(n, dev)
    x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 3)
    want = torch.full((N,), 2.0, dtype=torch.float64, device=dev)
    ops.SpMat(ptr, col, val).apply(x, want, -0.5, True)
    A = DistSpMat(ptr, col, val, N, N)
    assert A.enable_native(graph=graph), A.native_error
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            y = torch.full((N,), 2.0, dtype=torch.float64, device=dev)
            ycopy = y
            A.apply(x, ycopy, -0.5, True)
    s.synchronize()
    assert torch.equal(ycopy, want)
