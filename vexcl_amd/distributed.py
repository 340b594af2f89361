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
import os

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


def halo_extended_csr(ptr, col, col_begin, rows, lo, hi):
    """A rank's strip (CSR, GLOBAL columns, first owned column `col_begin`) as ONE square matrix that includes its ghost planes
    (transport "halo"): `lo` empty rows in front of the rank's rows and `hi` empty rows behind them, columns counted from the first
    element of the lower ghost plane -- so that x of that matrix is [lower ghost plane | the rank's segment | upper ghost plane] and
    its rows lo .. lo + rows are the rank's rows of the global product, entries in their global column order.
    -> (ptr_ext, col_ext) as int32 tensors; raises if a column lies outside the two ghost planes."""
    dev = col.device
    last = ptr[-1:].to(torch.int32)
    ptr_ext = torch.cat([torch.zeros(lo, dtype=torch.int32, device=dev), ptr.to(torch.int32), last.expand(hi)]).contiguous()
    col_ext = col.to(torch.int64) - (col_begin - lo)
    if col_ext.numel() and (int(col_ext.min()) < 0 or int(col_ext.max()) >= lo + rows + hi):
        raise RuntimeError("transport halo: a column outside the two ghost planes")
    return ptr_ext, col_ext.to(torch.int32).contiguous()


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

    def __init__(self, ptr, col, val, n_rows, n_cols, group=None, local_fmt="auto", kernels=None, keep_strip=False):
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

        # the strip as it was handed over (GLOBAL columns): transport "halo" stores it once more, ghost planes included
        # (enable_native).  Kept only on request (keep_strip=True, or enable_native(..., strip=(ptr, col, val))): the references
        # would otherwise hold several GB per rank beside the local / remote split built below; drop_strip() releases them
        self._strip = (ptr, col, val) if keep_strip else None
        self._ext = None
        # ---- split into local / remote parts (setup; not on the timed path)
        # On the GPU (round 6): the library's own split (csrc/split.hip, vexhip_csr_split_*: what vexcl/spmat.hpp's set-up calls) -- local
        # part, remote part as a row-subset CSR with its columns renumbered to their rank in the sorted ghost set, the ghost set itself.
        # (Until then torch's boolean-mask indexing / unique / bincount did it: correct, several GB of temporaries, and with FOUR
        # processes sharing one GPU its rocPRIM partition stalled -- tools/r06_n4_probe.py.)
        if (dev.type == "cuda" and kernels is None and ptr.dtype == torch.int32 and col.dtype == torch.int32
                and val.dtype in (torch.float64, torch.float32) and os.environ.get("VEXCL_AMD_TORCH_SPLIT") != "1"):
            self._split_native(ptr, col, val, c0, c1, local_fmt)
            ghosts = self.ghosts
        else:
            ghosts = self._split_torch(ptr, col, val, c0, c1, local_fmt)
        self._plan_exchange(ghosts, c0, val, dev)

    def _split_native(self, ptr, col, val, c0, c1, local_fmt):
        import ctypes
        from . import _capi
        L = _capi.lib()
        dev = val.device
        d = dev.index or 0
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        ptr, col, val = ptr.contiguous(), col.contiguous(), val.contiguous()
        sizes = (ctypes.c_int64 * 4)()
        L.csr_split_sizes_i32(d, stream, self.rows, p(ptr), p(col), c0, c1, sizes)
        nloc, nrem, nrows = int(sizes[0]), int(sizes[1]), int(sizes[2])
        i32 = dict(dtype=torch.int32, device=dev)
        lptr = torch.empty(self.rows + 1, **i32); lcol = torch.empty(max(nloc, 1), **i32); lval = torch.empty(max(nloc, 1), dtype=val.dtype, device=dev)
        rrows = torch.empty(max(nrows, 1), **i32); rptr = torch.empty(nrows + 1, **i32)
        rcol = torch.empty(max(nrem, 1), **i32); rval = torch.empty(max(nrem, 1), dtype=val.dtype, device=dev); gh = torch.empty(max(nrem, 1), **i32)
        split = L.csr_split_f64_i32 if val.dtype == torch.float64 else L.csr_split_f32_i32
        split(d, stream, self.rows, p(ptr), p(col), p(val), c0, c1, sizes, p(lptr), p(lcol), p(lval), p(rrows), p(rptr), p(rcol), p(rval), p(gh))
        torch.cuda.current_stream(dev).synchronize()
        nghost = int(sizes[3])
        self.ghosts = gh[:nghost].to(torch.int64)                  # sorted global ids
        self.loc = self.k.make_matrix(lptr, lcol[:nloc], lval[:nloc], self.local_cols, local_fmt) if nloc else None
        self.rem = self.k.make_remote(rrows[:nrows], rptr, rcol[:nrem], rval[:nrem]) if nghost else None

    def _split_torch(self, ptr, col, val, c0, c1, local_fmt):
        dev = val.device
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
        return ghosts

    def _plan_exchange(self, ghosts, c0, val, dev):
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
        self._native, self._comm, self._window, self.native_error, self.native_transport = None, None, None, None, None

    # ---- the product step issued from C++ (include/vexhip.h vexhip_dist_spmv_*) --------------------------------------
    def _coll_device(self):
        """where tensors of set-up collectives live: the GPU for nccl, the host for gloo"""
        return self.dev if (dist.is_initialized() and dist.get_backend(self.group) == "nccl") else torch.device("cpu")

    def _agree(self, ok):
        """True only if EVERY rank says ok (all-reduce MIN).  Every stage of enable_native ends in one of these, so a rank
        that fails early never leaves the others blocked in a later broadcast / ncclCommInitRank / all-gather."""
        if self.world == 1 or not dist.is_initialized():
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self._coll_device())
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.cpu()[0]))

    def _stage(self, fn):
        """run one local set-up stage; remember why it failed; agree with the other ranks"""
        ok = True
        try:
            fn()
        except Exception as e:
            self.native_error = repr(e)
            ok = False
        if not self._agree(ok):
            if ok:
                self.native_error = self.native_error or "another rank could not set the step up"
            return False
        return True

    def enable_native(self, graph=False, transport="rccl", strip=None):
        """Replace the per-product Python step (gather launch, torch.distributed request objects, two ctypes launches)
        by ONE call into libvexhip.  transport "rccl": pack -> grouped ncclSend/ncclRecv on a second stream -> local
        part -> remote part, over its own RCCL communicator (the unique id travels over this process group);
        transport "ipc": peer-mapped ghost windows -- every owner writes its neighbours' shares straight into their
        windows (hipIpcGetMemHandle; the handles travel over this process group), no communicator at all.
        Collective: every rank must call it; it becomes active on all ranks or on none (`native_error` says why)."""
        import ctypes
        from . import _capi
        self.disable_native()
        if strip is not None:
            self._strip = tuple(strip)
        self.native_error = None
        L = None
        st = {}

        def pre():
            nonlocal L
            if transport not in ("rccl", "ipc", "halo", "pull"):
                raise ValueError("unknown transport %r" % (transport,))
            if self.dev.type != "cuda" or not hasattr(self.k, "make_remote"):
                raise RuntimeError("native step needs the device kernels")
            if self.loc is not None and not getattr(self.loc, "handle", None):
                raise RuntimeError("local part is not a vexhip_spmat")
            L = _capi.lib()
        if not self._stage(pre):
            return False

        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else None
        sc = (ctypes.c_int64 * self.world)(*[int(c) for c in self.send_counts])
        rc = (ctypes.c_int64 * self.world)(*[int(c) for c in self.recv_counts])
        rem = self.rem
        dt = _capi.F64 if self.send_buf.dtype == torch.float64 else _capi.F32
        rem_args = (rem.rows.numel() if rem is not None else 0,
                    p(rem.rows) if rem is not None else None, p(rem.ptr) if rem is not None else None,
                    p(rem.col) if rem is not None else None, p(rem.val) if rem is not None else None)
        cdev = self._coll_device()
        try:
            if transport == "rccl":
                def make_id():
                    st["id"] = torch.zeros(128, dtype=torch.uint8)
                    if self.rank == 0:
                        raw = (ctypes.c_char * 128)()
                        L.comm_unique_id(ctypes.cast(raw, ctypes.c_void_p))
                        st["id"] = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
                if not self._stage(make_id):
                    return False
                idbuf = st["id"]
                if self.world > 1:
                    t = idbuf.to(cdev)
                    dist.broadcast(t, src=self._global_rank(0), group=self.group)
                    idbuf = t.cpu()
                idbytes = (ctypes.c_char * 128).from_buffer_copy(bytes(idbuf.numpy().tobytes()))

                def make_step():
                    comm = ctypes.c_void_p()
                    L.comm_init_rank(self.dev.index or 0, self.rank, self.world, ctypes.cast(idbytes, ctypes.c_void_p), ctypes.byref(comm))
                    self._comm = comm
                    step = ctypes.c_void_p()
                    L.dist_spmv_create(comm, dt, self.rows, self.loc.handle if self.loc is not None else None, *rem_args,
                                       self.send_idx.numel(), p(self.send_idx), p(self.send_buf), sc,
                                       self.ghost_buf.numel(), p(self.ghost_buf), rc, ctypes.byref(step))
                    st["step"] = step
                if not self._stage(make_step):
                    self._drop_native(st.get("step"))
                    return False
            elif transport in ("halo", "pull"):
                # ---- the whole step in ONE launch (include/vexhip.h vexhip_dist_spmv_create_halo; csrc/halo.hpp): every remote
                #      column of this rank lies in the plane below its first row or the plane above its last one, the strip is
                #      stored as one grid matrix with those two ghost planes, and the plane product reads them from the window
                def make_ext():
                    st["halo"] = self._halo_plan(pull=(transport == "pull"))
                if not self._stage(make_ext):
                    return False
                H, lower, upper = st["halo"]

                def make_window():
                    win = ctypes.c_void_p()
                    # (pull, round 6: the window carries flags only -- the ghost planes are read from the neighbours' x where it lies)
                    L.ipc_window_create(self.dev.index or 0, self.rank, self.world, 0 if transport == "pull" else 2 * H * 8, ctypes.byref(win))
                    self._window = win
                    raw = (ctypes.c_char * 64)()
                    L.ipc_window_export(win, ctypes.cast(raw, ctypes.c_void_p))
                    st["handle"] = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
                if not self._stage(make_window):
                    self._drop_native(None)
                    return False
                handles = [st["handle"]]
                geo = [torch.tensor([H, self.rows, lower, upper], dtype=torch.int64)]
                if self.world > 1:
                    hs = [torch.empty(64, dtype=torch.uint8, device=cdev) for _ in range(self.world)]
                    dist.all_gather(hs, st["handle"].to(cdev), group=self.group)
                    gs = [torch.empty(4, dtype=torch.int64, device=cdev) for _ in range(self.world)]
                    dist.all_gather(gs, geo[0].to(cdev), group=self.group)
                    handles, geo = [h.cpu() for h in hs], [g.cpu() for g in gs]

                def make_step():
                    for peer in (lower, upper):
                        if peer >= 0 and peer != self.rank:
                            if int(geo[peer][0]) != H or int(geo[peer][1]) < H:
                                raise RuntimeError("the neighbours' strips do not have planes of the same size")
                            # the neighbour must point back at this rank (its own plan was made from ITS ghosts)
                            if int(geo[peer][3 if peer == lower else 2]) != self.rank:
                                raise RuntimeError("rank %d does not exchange with this rank: the neighbour relation is not symmetric" % peer)
                            hb = (ctypes.c_char * 64).from_buffer_copy(bytes(handles[peer].numpy().tobytes()))
                            L.ipc_window_open(self._window, peer, ctypes.cast(hb, ctypes.c_void_p))
                    step = ctypes.c_void_p()
                    if transport == "pull":
                        L.dist_spmv_create_halo_pull(self._window, self._ext.handle, self.rows, H, lower, upper, 1, ctypes.byref(step))
                        self._pull = {"H": H, "lower": lower, "upper": upper, "vectors": {}, "opened": {}}
                    else:
                        L.dist_spmv_create_halo(self._window, self._ext.handle, self.rows, H, lower, upper, ctypes.byref(step))
                    st["step"] = step
                if not self._stage(make_step):
                    self._drop_native(st.get("step"))
                    return False
            else:
                def make_window():
                    win = ctypes.c_void_p()
                    L.ipc_window_create(self.dev.index or 0, self.rank, self.world,
                                        int(self.ghost_buf.numel()) * self.ghost_buf.element_size(), ctypes.byref(win))
                    self._window = win
                    raw = (ctypes.c_char * 64)()
                    L.ipc_window_export(win, ctypes.cast(raw, ctypes.c_void_p))
                    st["handle"] = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
                if not self._stage(make_window):
                    self._drop_native(None)
                    return False
                handles = [st["handle"]]
                counts = [torch.tensor([int(c) for c in self.recv_counts], dtype=torch.int64)]
                if self.world > 1:
                    hs = [torch.empty(64, dtype=torch.uint8, device=cdev) for _ in range(self.world)]
                    dist.all_gather(hs, st["handle"].to(cdev), group=self.group)
                    cs = [torch.empty(self.world, dtype=torch.int64, device=cdev) for _ in range(self.world)]
                    dist.all_gather(cs, counts[0].to(cdev), group=self.group)
                    handles, counts = [h.cpu() for h in hs], [c.cpu() for c in cs]

                def make_step():
                    for peer in range(self.world):
                        if peer != self.rank and (self.send_counts[peer] or self.recv_counts[peer]):
                            hb = (ctypes.c_char * 64).from_buffer_copy(bytes(handles[peer].numpy().tobytes()))
                            L.ipc_window_open(self._window, peer, ctypes.cast(hb, ctypes.c_void_p))
                    # my share sits in peer p's ghost set behind the shares of the owners before me
                    offs = (ctypes.c_int64 * self.world)(*[int(counts[peer][:self.rank].sum()) for peer in range(self.world)])
                    if any(int(counts[peer][self.rank]) != int(self.send_counts[peer]) for peer in range(self.world)):
                        raise RuntimeError("exchange plans of the ranks do not match")
                    step = ctypes.c_void_p()
                    L.dist_spmv_create_ipc(self._window, dt, self.rows, self.loc.handle if self.loc is not None else None, *rem_args,
                                           self.send_idx.numel(), p(self.send_idx), sc, offs,
                                           self.ghost_buf.numel(), rc, ctypes.byref(step))
                    st["step"] = step
                if not self._stage(make_step):
                    self._drop_native(st.get("step"))
                    return False
        except Exception as e:              # a failed collective: nothing sensible is left to agree on
            self.native_error = repr(e)
            self._drop_native(st.get("step"))
            return False
        if graph:
            L.dist_spmv_set_graph(st["step"], 1)
        self._native = st["step"]
        self.native_transport = transport
        return True

    def register_vector(self, x):
        """Transport "pull" (COLLECTIVE: every rank calls it with its segment of the same vector): the neighbours learn where this rank's
        x lies -- the IPC handle of its allocation and x's offset in it travel over the process group -- and map it; products with
        this x then read its boundary planes in place.  apply() calls it for an x it has not seen (all ranks must then see a new x in
        the same product).  The mapping stays until disable_native()."""
        import ctypes
        from . import _capi
        L = _capi.lib()
        P = self._pull
        raw = (ctypes.c_char * 64)(); off = ctypes.c_int64()
        L.ipc_export(self.dev.index or 0, ctypes.c_void_p(x.data_ptr()), ctypes.cast(raw, ctypes.c_void_p), ctypes.byref(off))
        mine = torch.cat([torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone(),
                          torch.tensor([off.value, x.numel()], dtype=torch.int64).view(torch.uint8)])
        everyone = [mine]
        if self.world > 1:
            cdev = self._coll_device()
            got = [torch.empty(80, dtype=torch.uint8, device=cdev) for _ in range(self.world)]
            dist.all_gather(got, mine.to(cdev), group=self.group)
            everyone = [g.cpu() for g in got]
        item = x.element_size()
        ptrs = {}
        for side, peer in (("lower", P["lower"]), ("upper", P["upper"])):
            if peer < 0:
                ptrs[side] = None
                continue
            rec = everyone[peer if self.world > 1 else 0]
            hbytes = bytes(rec[:64].numpy().tobytes())
            poff, pn = [int(v) for v in rec[64:].view(torch.int64)]
            if peer == self.rank:
                base = x.data_ptr() - off.value                      # (one rank exchanging with itself: tools/r06_dist_step.py)
            else:
                base = P["opened"].get(hbytes)
                if base is None:
                    hb = (ctypes.c_char * 64).from_buffer_copy(hbytes); b = ctypes.c_void_p()
                    L.ipc_open(self.dev.index or 0, ctypes.cast(hb, ctypes.c_void_p), ctypes.byref(b))
                    base = b.value; P["opened"][hbytes] = base
            # the lower neighbour's LAST plane, the upper neighbour's FIRST plane
            ptrs[side] = base + poff + ((pn - P["H"]) * item if side == "lower" else 0)
        P["vectors"][x.data_ptr()] = (ptrs["lower"], ptrs["upper"])
        return P["vectors"][x.data_ptr()]

    def drop_strip(self):
        """release the references to the strip the constructor was given (only transport "halo" needs them)"""
        self._strip = None

    def _halo_plan(self, self_exchange=False, pull=False):
        """Transport "halo": (H, lower, upper) and self._ext = the strip stored with its ghost planes, or an exception that says
        why this matrix / partition does not qualify.  H = elements of a ghost plane; lower / upper = the neighbours (-1: none).
        self_exchange (one rank, tools/r05_dist_step.py): the rank is its own lower and upper neighbour."""
        from . import ops
        if self._strip is None:
            raise RuntimeError("the strip has been dropped")
        ptr, col, val = self._strip
        if val.dtype != torch.float64:
            raise RuntimeError("transport halo: fp64 only")
        if self.part != self.col_part:
            raise RuntimeError("transport halo: rows and columns must be partitioned alike")
        c0, c1 = self.col_part[self.rank], self.col_part[self.rank + 1]
        g = self.ghosts
        below, above = g[g < c0], g[g >= c1]
        has_lo, has_hi = below.numel() > 0, above.numel() > 0
        if self_exchange:
            lower = upper = self.rank
            has_lo = has_hi = True
        else:
            lower = self.rank - 1 if has_lo else -1
            upper = self.rank + 1 if has_hi else -1
            owners = [o for o in range(self.world) if self.recv_counts[o]]
            if any(o not in (lower, upper) for o in owners):
                raise RuntimeError("transport halo: ghosts come from other ranks than the two neighbours")
            # the protocol is symmetric: a rank pushes its boundary plane to exactly the neighbours it has ghosts from, and its push
            # workgroups wait for `sent` flags only such a neighbour raises.  A one-sided coupling (an upwind stencil: entries at
            # -far / -nx / -1 only) would leave rank r waiting for a share rank r - 1 never sends -- decline here, on every rank
            # (the staging all-reduce makes the ranks decline together), instead of running into the flag time-out
            takers = {o for o in range(self.world) if self.send_counts[o]}
            if takers != {o for o in (lower, upper) if o >= 0}:
                raise RuntimeError("transport halo: the coupling between neighbouring strips is not symmetric "
                                   "(this rank receives from %s and sends to %s)" % (sorted(o for o in (lower, upper) if o >= 0), sorted(takers)))
        reach = max(int(c0 - below.min()) if below.numel() else 0, int(above.max()) + 1 - c1 if above.numel() else 0)
        # the ghost range: a plane of the stored grid.  Planes of 512-point lines and slices of the SELL-512 storage are multiples of 1024
        # / 512 elements; a grid with other lines (500^3; round 6: the rows of a 2-D grid, cut into virtual lines) has planes of
        # exactly `reach` elements -- tried first where the two differ (pull only: the grid product has no push form)
        rounded = (reach + 1023) // 1024 * 1024
        why = "transport halo: the strip is not a whole number of planes of %d elements" % rounded
        exact = 0                      # the shortest plane the strip is a whole number of (the first and the last point of a plane are boundary rows: reach <= plane)
        if pull and reach > 0:
            for k in range(self.rows // reach, max(self.rows // reach - 4096, 0), -1):
                if self.rows % k == 0:
                    exact = self.rows // k
                    break
        for H in ([exact] if exact and exact != rounded and exact % 2 == 0 else []) + [rounded]:
            if H <= 0 or self.rows % H:
                continue
            lo, hi = (H if has_lo else 0), (H if has_hi else 0)
            ptr_ext, col_ext = halo_extended_csr(ptr, col, 0 if self_exchange else c0, self.rows, lo, hi)
            ext = ops.SpMat(ptr_ext, col_ext, val, n_cols=lo + self.rows + hi)
            # pushed shares (halo): the plane product only; shares read in place (pull): also the grid product and -- round 6 -- any strip
            # stored with diagonal codes whose diagonals stay within one ghost range (the library checks: vexhip_dist_spmv_create_halo_pull)
            plane, grid = getattr(ext, "plane", None), getattr(ext, "grid", None)
            if getattr(ext, "handle", None):
                if plane and H == plane["lines_per_plane"] * 512:
                    break
                if pull and grid and not plane and H == grid["nx"] * grid["lines_per_plane"]:
                    break
                if pull and not plane and not grid and H % 512 == 0 and getattr(ext, "storage", "") in ("sell8", "sell8v"):
                    break
            why = "transport %s: the stored strip did not get a plan for the one-launch step with ghost ranges of %d elements (storage %s)" % ("pull" if pull else "halo", H, getattr(ext, "storage", "?"))
            del ext
        else:
            raise RuntimeError(why)
        ext.ptr = ext.col = ext.val = None
        self._ext = ext
        return H, lower, upper

    def _drop_native(self, step):
        import ctypes
        from . import _capi
        L = _capi.lib()
        if step:
            L.dist_spmv_destroy(step)
        if getattr(self, "_comm", None):
            L.comm_destroy(self._comm)
            self._comm = None
        if getattr(self, "_window", None):
            L.ipc_window_destroy(self._window)
            self._window = None
        self._ext = None            # the strip stored with its ghost planes belongs to the step that has just gone
        P, self._pull = getattr(self, "_pull", None), None
        if P:
            for base in P["opened"].values():
                try:
                    L.ipc_close(self.dev.index or 0, ctypes.c_void_p(base))
                except Exception:           # noqa: BLE001 -- the owner may be gone already
                    pass

    def disable_native(self):
        step, self._native = getattr(self, "_native", None), None
        self.native_transport = None
        if step or getattr(self, "_comm", None) or getattr(self, "_window", None):
            if self.dev.type == "cuda":
                torch.cuda.synchronize(self.dev)
            self._drop_native(step)

    def native_status(self):
        """(timed_out, transport, direct) of the active C++ step: timed_out != 0 means a flag wait of the IPC transport ran
        into its bound (a peer did not show up)."""
        import ctypes
        from . import _capi
        if not self._native:
            return None
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _capi.lib().dist_spmv_status(self._native, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return {"timed_out": a.value, "transport": "pull" if getattr(self, "_pull", None) else {1: "rccl", 3: "ipc", 4: "halo"}.get(b.value, b.value), "direct": bool(c.value)}

    def rccl_info(self):
        import ctypes
        from . import _capi
        if not getattr(self, "_comm", None):
            return None
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _capi.lib().comm_rccl_info(self._comm, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return {"ncclCommCount": a.value, "ncclCommCuDevice": b.value, "ncclCommUserRank": c.value}

    def profile_step(self, x, y, alpha=1.0, append=False):
        """One product through the C++ step with its phases timed (ms): total, local, wait, remote, pack, exchange."""
        import ctypes
        from . import _capi
        if not self._native:
            return None
        ms = (ctypes.c_float * 6)()
        _capi.lib().dist_spmv_profile(self._native, ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream),
                                      float(alpha), int(bool(append)), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ms)
        return dict(zip(("total", "local", "wait_for_ghosts", "remote", "pack", "exchange"), [float(v) for v in ms]))

    def __del__(self):
        try:
            self._drop_native(getattr(self, "_native", None))
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
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
            if getattr(self, "_pull", None):
                peers = self._pull["vectors"].get(x.data_ptr()) or self.register_vector(x)
                _capi.lib().dist_spmv_apply_pull(self._native, stream, float(alpha), int(bool(append)), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                                 ctypes.c_void_p(peers[0]) if peers[0] else None, ctypes.c_void_p(peers[1]) if peers[1] else None)
                return y
            _capi.lib().dist_spmv_apply(self._native, stream,
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
