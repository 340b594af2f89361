"""Row-partitioned SpMat across GPUs, one process per GPU (torch.distributed;
backend "nccl" is RCCL over xGMI).

The reference partitions rows/columns contiguously across the devices of ONE
process (vexcl/vector.hpp:131-167) and stages the ghost exchange through host
memory in five phases (vexcl/spmat.hpp:120-185, setup :291-378).  Here each
rank owns one strip, and the exchange is one grouped point-to-point step
(`ncclSend/ncclRecv` per neighbour that needs at least one value) overlapped
with the local-part product:

    gather (pack owned boundary values)        compute stream
    grouped isend/irecv over RCCL              comm stream      | overlapped
    y  = alpha * A_loc * x_loc                 compute stream   |
    y += alpha * A_rem * ghosts                compute stream, after the recv

Semantics kept from the reference: column partition = partition(m, world);
local columns renumbered c - col_begin (spmat/csr.inl:92-131); ghost columns
renumbered to their rank in the sorted ghost set (hybrid_ell.inl:132-136);
`y = alpha*A*x` or `y += alpha*A*x` (spmat.hpp:120-121).  Because ghosts are
sorted by global column and owners hold contiguous column ranges, the ghost
buffer is the concatenation, in rank order, of what each owner sends.
"""
import torch
import torch.distributed as dist


def partition(n, nparts):
    """vex::partition with equal weights (vector.hpp:157-162): boundaries at
    alignup(n*d/nparts, 16), clamped to n."""
    part = [0]
    for d in range(1, nparts):
        b = (n * d // nparts + 15) // 16 * 16
        part.append(min(n, b))
    part.append(n)
    return part


class DeviceKernels:
    """The HIP kernels behind the distributed product (no CPU stand-in here)."""

    def make_matrix(self, ptr, col, val, n_cols, fmt):
        from . import ops
        return ops.SpMat(ptr, col, val, n_cols=n_cols, fmt=fmt)

    def gather(self, idx, src, dst):
        from . import ops
        return ops.gather(idx, src, dst)

    def make_remote(self, rows, ptr, col, val):
        """Remote part: entries only in the rows next to a partition boundary."""
        from . import ops
        return ops.RowSubsetCSR(rows, ptr, col, val)


class DistSpMat:
    """One rank's strip of a row-partitioned sparse matrix.

    ptr/col/val: CSR of rows [part[rank], part[rank+1]) with GLOBAL column ids
    (int32), on this rank's device.  x and y passed to ``apply`` are this
    rank's segments of the partitioned vectors (vex::vector's x(d), y(d))."""

    def __init__(self, ptr, col, val, n_rows, n_cols, group=None, local_fmt="auto", kernels=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.k = kernels or DeviceKernels()
        self.n_rows, self.n_cols = n_rows, n_cols
        self.part = partition(n_rows, self.world)
        self.col_part = partition(n_cols, self.world)
        r0, r1 = self.part[self.rank], self.part[self.rank + 1]
        c0, c1 = self.col_part[self.rank], self.col_part[self.rank + 1]
        self.rows = r1 - r0
        self.local_cols = c1 - c0
        if ptr.numel() != self.rows + 1:
            raise ValueError("strip has %d rows, partition expects %d" % (ptr.numel() - 1, self.rows))
        dev = val.device
        self.dev = dev

        # ---- split into local / remote parts (setup; not on the timed path)
        is_loc = (col >= c0) & (col < c1)
        rem_mask = ~is_loc
        ghosts = torch.unique(col[rem_mask].to(torch.int64))            # sorted global ids
        self.ghosts = ghosts
        row_of = torch.repeat_interleave(torch.arange(self.rows, device=dev),
                                         (ptr[1:] - ptr[:-1]).to(torch.int64))

        def sub(mask, cols):
            cnt = torch.bincount(row_of[mask], minlength=self.rows)
            p = torch.zeros(self.rows + 1, dtype=torch.int64, device=dev)
            p[1:] = torch.cumsum(cnt, 0)
            return p.to(torch.int32), cols.to(torch.int32).contiguous(), val[mask].contiguous()

        lp, lc, lv = sub(is_loc, col[is_loc] - c0)
        self.loc = self.k.make_matrix(lp, lc, lv, self.local_cols, local_fmt) if lc.numel() else None
        if ghosts.numel():
            rp, rc, rv = sub(rem_mask, torch.searchsorted(ghosts, col[rem_mask].to(torch.int64)))
            if hasattr(self.k, "make_remote"):
                # only the rows that reach a ghost column (2 planes of 64 for 512^3 over 8 GPUs)
                cnt = (rp[1:] - rp[:-1])
                rows_with = torch.nonzero(cnt > 0).flatten().to(torch.int32)
                cp = torch.zeros(rows_with.numel() + 1, dtype=torch.int32, device=dev)
                cp[1:] = torch.cumsum(cnt[rows_with.long()], 0).to(torch.int32)
                self.rem = self.k.make_remote(rows_with, cp, rc, rv)
            else:
                self.rem = self.k.make_matrix(rp, rc, rv, int(ghosts.numel()), "csr")
        else:
            self.rem = None
        del row_of, is_loc, rem_mask

        # ---- exchange plan: who needs which of my columns
        self.recv_counts = [0] * self.world         # from each owner, in ghost order
        self.send_counts = [0] * self.world
        self.send_idx = torch.empty(0, dtype=torch.int32, device=dev)
        if self.world > 1:
            bounds = torch.tensor(self.col_part, dtype=torch.int64, device=dev)
            seg = torch.searchsorted(ghosts, bounds)                    # ghosts owned by rank o: [seg[o], seg[o+1])
            self.recv_counts = (seg[1:] - seg[:-1]).tolist()
            # every rank tells every owner how many values it wants, then which
            want = torch.tensor(self.recv_counts, dtype=torch.int64, device=dev)
            give = torch.empty_like(want)
            self._all_to_all_single(give, want)
            self.send_counts = give.tolist()
            reqs_out = ghosts.contiguous()                              # already grouped by owner
            reqs_in = torch.empty(sum(self.send_counts), dtype=torch.int64, device=dev)
            self._all_to_all_v(reqs_in, self.send_counts, reqs_out, self.recv_counts)
            self.send_idx = (reqs_in - c0).to(torch.int32).contiguous()   # local ids (spmat.hpp:360-365)
            if self.send_idx.numel():
                assert int(self.send_idx.min()) >= 0 and int(self.send_idx.max()) < self.local_cols
        self.send_buf = torch.empty(self.send_idx.numel(), dtype=val.dtype, device=dev)
        self.ghost_buf = torch.empty(int(ghosts.numel()), dtype=val.dtype, device=dev)
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._p2p = self._plan_p2p()
        self._ops = None
        self._native, self._comm, self.native_error = None, None, None

    # ---- the product step issued from C++ (include/vexhip.h vexhip_dist_spmv_*) --------------------------------------
    def enable_native(self, graph=False):
        """Replace the per-product Python step (gather launch, torch.distributed request objects, two ctypes launches)
        by ONE call into libvexhip: pack -> grouped ncclSend/ncclRecv on a second stream -> local part -> remote part,
        with its own RCCL communicator (the unique id travels over this process group).  Returns True when active;
        on any failure the torch.distributed transport stays in place and `native_error` says why."""
        import ctypes
        from . import _capi
        try:
            if self.dev.type != "cuda" or not hasattr(self.k, "make_remote"):
                raise RuntimeError("native step needs the device kernels")
            if self.loc is not None and not getattr(self.loc, "handle", None):
                raise RuntimeError("local part is not a vexhip_spmat")
            L = _capi.lib()
            idbuf = torch.zeros(128, dtype=torch.uint8)
            if self.rank == 0:
                raw = (ctypes.c_char * 128)()
                L.comm_unique_id(ctypes.cast(raw, ctypes.c_void_p))
                idbuf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
            if self.world > 1:
                t = idbuf.to(self.dev) if dist.get_backend(self.group) == "nccl" else idbuf
                dist.broadcast(t, src=self._global_rank(0), group=self.group)
                idbuf = t.cpu()
            idbytes = (ctypes.c_char * 128).from_buffer_copy(bytes(idbuf.numpy().tobytes()))
            comm = ctypes.c_void_p()
            L.comm_init_rank(self.dev.index or 0, self.rank, self.world, ctypes.cast(idbytes, ctypes.c_void_p), ctypes.byref(comm))
            self._comm = comm
            p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else None
            sc = (ctypes.c_int64 * self.world)(*[int(c) for c in self.send_counts])
            rc = (ctypes.c_int64 * self.world)(*[int(c) for c in self.recv_counts])
            rem = self.rem
            step = ctypes.c_void_p()
            L.dist_spmv_create(comm, _capi.F64 if self.send_buf.dtype == torch.float64 else _capi.F32, self.rows,
                               self.loc.handle if self.loc is not None else None,
                               rem.rows.numel() if rem is not None else 0,
                               p(rem.rows) if rem is not None else None, p(rem.ptr) if rem is not None else None,
                               p(rem.col) if rem is not None else None, p(rem.val) if rem is not None else None,
                               self.send_idx.numel(), p(self.send_idx), p(self.send_buf), sc,
                               self.ghost_buf.numel(), p(self.ghost_buf), rc, ctypes.byref(step))
            if graph:
                L.dist_spmv_set_graph(step, 1)
            self._native = step
            return True
        except Exception as e:          # keep the torch.distributed transport
            self.native_error = repr(e)
            self._native = None
            return False

    def disable_native(self):
        from . import _capi
        if self._native:
            _capi.lib().dist_spmv_destroy(self._native)
        self._native = None

    def __del__(self):
        try:
            from . import _capi
            if getattr(self, "_native", None):
                _capi.lib().dist_spmv_destroy(self._native)
            if getattr(self, "_comm", None):
                _capi.lib().comm_destroy(self._comm)
        except Exception:
            pass

    # ---- collectives used only at setup (portable across nccl / gloo) -------
    def _all_to_all_single(self, out, inp):
        gathered = [torch.empty_like(inp) for _ in range(self.world)]
        dist.all_gather(gathered, inp, group=self.group)
        for o in range(self.world):
            out[o] = gathered[o][self.rank]

    def _all_to_all_v(self, out, out_counts, inp, in_counts):
        ops, off_in, off_out = [], 0, 0
        for peer in range(self.world):
            ni, no = in_counts[peer], out_counts[peer]
            if peer != self.rank:
                if ni:
                    ops.append(dist.P2POp(dist.isend, inp[off_in:off_in + ni], self._global_rank(peer), self.group))
                if no:
                    ops.append(dist.P2POp(dist.irecv, out[off_out:off_out + no], self._global_rank(peer), self.group))
            off_in += ni
            off_out += no
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()

    def _global_rank(self, peer):
        return peer if self.group is None else dist.get_global_rank(self.group, peer)

    def _plan_p2p(self):
        plan, so, ro = [], 0, 0
        for peer in range(self.world):
            ns, nr = self.send_counts[peer], self.recv_counts[peer]
            if peer != self.rank:
                if ns:
                    plan.append(("send", peer, so, ns))
                if nr:
                    plan.append(("recv", peer, ro, nr))
            so += ns
            ro += nr
        return plan

    # ---- the product ---------------------------------------------------------
    def exchange_bytes(self):
        """payload this rank sends + receives per product (xGMI traffic)."""
        return (self.send_buf.numel() + self.ghost_buf.numel()) * self.send_buf.element_size()

    def _p2p_ops(self):
        """The grouped send/recv list is fixed for the life of the matrix: built once."""
        if self._ops is None:
            self._ops = []
            for kind, peer, off, cnt in self._p2p:
                buf = (self.send_buf if kind == "send" else self.ghost_buf)[off:off + cnt]
                self._ops.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, buf,
                                            self._global_rank(peer), self.group))
        return self._ops

    def apply(self, x, y, alpha=1.0, append=False):
        """y (=|+=) alpha * A * x on this rank's strip (spmat.hpp:120-185)."""
        if x.numel() != self.local_cols or y.numel() != self.rows:
            raise ValueError("segment sizes do not match the partition")
        if self._native:
            import ctypes
            from . import _capi
            _capi.lib().dist_spmv_apply(self._native, ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream),
                                        float(alpha), int(bool(append)), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()))
            return y
        reqs = ()
        if self._p2p:
            self.k.gather(self.send_idx, x, self.send_buf)                  # phase 1: pack
            ops = self._p2p_ops()
            if self.comm_stream is not None:
                self.comm_stream.wait_stream(torch.cuda.current_stream(self.dev))
                with torch.cuda.stream(self.comm_stream):
                    reqs = dist.batch_isend_irecv(ops)                      # phases 3-4 in one xGMI hop
            else:
                reqs = dist.batch_isend_irecv(ops)
        if self.loc is not None:                                            # phase 2, overlapped
            self.loc.apply(x, y, alpha, append)
        elif not append:
            y.zero_()                                                       # csr.inl:196-199
        for r in reqs:
            r.wait()                                                        # compute stream waits for the recv
        if self.comm_stream is not None and reqs:
            torch.cuda.current_stream(self.dev).wait_stream(self.comm_stream)
        if self.rem is not None:                                            # phase 5
            self.rem.apply(self.ghost_buf, y, alpha, True)
        return y


class DistReductor:
    """vex::Reductor across the GPUs of a job: each rank reduces its segment to ONE
    scalar on the device (both stages, vexcl_amd.ops.Reductor) and the final
    combine is an all-reduce of that scalar over RCCL (the reference folds the
    per-device partials on the host, reductor.hpp:412-436)."""
    _OPS = {"SUM": dist.ReduceOp.SUM, "SUM_Kahan": dist.ReduceOp.SUM, "MIN": dist.ReduceOp.MIN, "MAX": dist.ReduceOp.MAX}

    def __init__(self, op="SUM", group=None, local=None):
        if op not in self._OPS:
            raise ValueError("unsupported distributed reduction %r" % op)
        self.op, self.group = op, group
        if local is None:
            from . import ops
            local = ops.Reductor(op)
        self.local = local

    def __call__(self, x):
        r = self.local.device_result(x).clone()
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(r, op=self._OPS[self.op], group=self.group)
        return r.cpu()[0].item()


class DistScan:
    """vex::inclusive_scan / vex::exclusive_scan of a vector partitioned across the ranks of a job.

    The reference scans every device's partition and then adds, on the host, the sum of
    the preceding partitions to each of them (scan.hpp:445-457, :489-506).  Here each rank
    scans its segment on its GPU (vexcl_amd.ops: single-pass look-back scan for integers),
    the ranks all-gather ONE element -- their segment's total -- over RCCL, and each rank
    adds the sum of the totals before it with one elementwise pass.  Integer scans wrap
    mod 2^k exactly as the single-device scan does (the carry is added in the same type).
    """

    def __init__(self, group=None, local=None):
        self.group = group
        self.local = local          # test double: object with inclusive_scan(inp, out) / exclusive_scan(inp, out, init)

    def _scan(self, inp, out, exclusive, init):
        if self.local is not None:
            return self.local.exclusive_scan(inp, out, init) if exclusive else self.local.inclusive_scan(inp, out)
        from . import ops
        return ops.exclusive_scan(inp, out, init) if exclusive else ops.inclusive_scan(inp, out)

    def __call__(self, inp, out=None, exclusive=False, init=0):
        if out is None:
            out = torch.empty_like(inp)
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        n = inp.numel()
        # the segment total must be taken before an in-place scan overwrites the input
        last_in = inp[-1:].clone() if (n and exclusive) else None
        self._scan(inp, out, exclusive, init if rank == 0 else 0)
        if world == 1:
            return out
        if n:
            total = out[-1:].clone()
            if exclusive:
                total = total + last_in
                if rank == 0:
                    total = total - init       # init belongs to the carry of every later rank once, added below
        else:
            total = torch.zeros(1, dtype=inp.dtype, device=inp.device)
        totals = [torch.empty_like(total) for _ in range(world)]
        dist.all_gather(totals, total, group=self.group)
        if rank > 0:
            carry = torch.stack(totals[:rank]).sum(0).to(inp.dtype)
            if exclusive:
                carry = carry + init
            if n:
                out.add_(carry)
        return out
