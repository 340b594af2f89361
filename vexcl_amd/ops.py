"""Host-side mirror of the reference's operator interface for the hot path,
over torch device tensors (torch supplies HBM allocations and streams only;
every computation goes through libvexhip.so, include/vexhip.h).

Names follow the reference: ``SpMat`` (vexcl/spmat.hpp:56-185), ``Reductor``
(vexcl/reductor.hpp:289-439), ``sort`` / ``sort_by_key`` (vexcl/sort.hpp:
2158-2182), ``inclusive_scan`` / ``exclusive_scan`` (vexcl/scan.hpp:461-518).
The C++ header API (``vexcl/*.hpp``) is the primary host side; this module is
what the Python parity tests and ``bench.py`` drive.
"""
import ctypes
import os
import sys
import time

import torch

from . import _capi
from ._capi import Error, lib

_DT = {torch.float64: _capi.F64, torch.float32: _capi.F32, torch.int32: _capi.I32, torch.int64: _capi.I64}
for _name, _code in (("uint32", _capi.U32), ("uint64", _capi.U64)):
    if hasattr(torch, _name):
        _DT[getattr(torch, _name)] = _code


def _dtype_code(t, unsigned=False):
    code = _DT.get(t.dtype)
    if code is None:
        raise Error("unsupported element type %s" % t.dtype)
    if unsigned and code == _capi.I32:
        code = _capi.U32
    if unsigned and code == _capi.I64:
        code = _capi.U64
    return code


def _dev(t):
    if not t.is_cuda:
        raise Error("vexcl_amd operates on HBM-resident tensors only (got a %s tensor); "
                    "there is no CPU fallback" % t.device.type)
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk(t, name):
    if not t.is_contiguous():
        raise Error("%s must be contiguous" % name)
    return t


# --------------------------------------------------------------------------
# raw kernels
# --------------------------------------------------------------------------
def csr_traversal(ptr, col):
    """Strip traversal for the CSR kernel of a banded / stencil matrix (include/vexhip.h
    `vexhip_csr_traversal_i32`); grid_blocks == 0 when no reordering pays."""
    trav = _capi.Traversal()
    if ptr.dtype == torch.int32 and col.dtype == torch.int32 and ptr.numel() > 1:
        lib().csr_traversal_i32(_dev(col), _stream(col), ptr.numel() - 1, _p(ptr), _p(col), 256, ctypes.byref(trav))
    return trav


def spmv_csr(ptr, col, val, x, y, alpha=1.0, append=False, traversal=None):
    """y (=|+=) alpha * A x  -- `csr_spmv`, vexcl/spmat/csr.inl:153-185."""
    n = ptr.numel() - 1
    for t, nm in ((ptr, "ptr"), (col, "col"), (val, "val"), (x, "x"), (y, "y")):
        _chk(t, nm)
    if y.numel() != n:
        raise Error("y has %d elements, matrix has %d rows" % (y.numel(), n))
    L = lib()
    if val.dtype == torch.float64 and ptr.dtype == torch.int32 and col.dtype == torch.int32:
        fn, a = L.spmv_csr_f64_i32, ctypes.c_double(alpha)
    elif val.dtype == torch.float32 and ptr.dtype == torch.int32 and col.dtype == torch.int32:
        fn, a = L.spmv_csr_f32_i32, ctypes.c_float(alpha)
    elif val.dtype == torch.float64 and ptr.dtype == torch.int64 and col.dtype == torch.int64:
        fn, a = L.spmv_csr_f64_i64, ctypes.c_double(alpha)
    else:
        raise Error("unsupported CSR type combination %s/%s/%s" % (val.dtype, ptr.dtype, col.dtype))
    if traversal is not None and traversal.grid_blocks > 0 and ptr.dtype == torch.int32:
        fo = L.spmv_csr_ordered_f64_i32 if val.dtype == torch.float64 else L.spmv_csr_ordered_f32_i32
        fo(_dev(y), _stream(y), n, a, int(bool(append)), _p(ptr), _p(col), _p(val), _p(x), _p(y), ctypes.byref(traversal))
        return y
    fn(_dev(y), _stream(y), n, a, int(bool(append)), _p(ptr), _p(col), _p(val), _p(x), _p(y))
    return y


class RowSubsetCSR:
    """`y[rows[k]] += alpha * A_k . x`: a CSR matrix that has entries only in the listed rows
    (the remote part of a partitioned matrix).  ptr has len(rows) + 1 entries."""

    def __init__(self, rows, ptr, col, val):
        for t, nm in ((rows, "rows"), (ptr, "ptr"), (col, "col"), (val, "val")):
            _chk(t, nm)
        if rows.dtype != torch.int32 or ptr.dtype != torch.int32 or col.dtype != torch.int32:
            raise Error("RowSubsetCSR needs int32 indices")
        self.rows, self.ptr, self.col, self.val = rows, ptr, col, val
        self.fmt = "csr-rows"

    def apply(self, x, y, alpha=1.0, append=True):
        if not append:
            raise Error("a row-subset product only adds to y")
        L = lib()
        f64 = self.val.dtype == torch.float64
        a = ctypes.c_double(alpha) if f64 else ctypes.c_float(alpha)
        (L.spmv_csr_rows_f64_i32 if f64 else L.spmv_csr_rows_f32_i32)(
            _dev(y), _stream(y), self.rows.numel(), a, _p(self.rows), _p(self.ptr), _p(self.col), _p(self.val), _p(x), _p(y))
        return y


def gather(idx, src, dst=None):
    """dst[i] = src[idx[i]] -- spmat.hpp:129-133 `permutation(cols_to_send)(x)`."""
    if dst is None:
        dst = torch.empty(idx.numel(), dtype=src.dtype, device=src.device)
    L = lib()
    fn = {torch.float64: L.gather_f64_i32, torch.float32: L.gather_f32_i32}.get(src.dtype)
    if fn is None or idx.dtype != torch.int32:
        raise Error("gather: unsupported types")
    fn(_dev(src), _stream(src), idx.numel(), _p(idx), _p(src), _p(dst))
    return dst


def poisson3d(n, device="cuda", rows=None, ptr64=None):
    """3-D Poisson matrix of examples/benchmark.cpp:364-415, built in HBM.
    rows=(r0, r1): only that row strip (global column ids, strip-local ptr).
    ptr64: row pointers as int64 (default: when the strip holds 2^31 entries or more); columns are int32."""
    L = lib()
    dev = torch.device(device)
    N = n ** 3
    r0, r1 = (0, N) if rows is None else rows
    nnz = L.poisson3d_strip_nnz(n, r0, r1)
    if ptr64 is None:
        ptr64 = nnz >= 2 ** 31
    ptr = torch.empty(r1 - r0 + 1, dtype=torch.int64 if ptr64 else torch.int32, device=dev)
    col = torch.empty(nnz, dtype=torch.int32, device=dev)
    val = torch.empty(nnz, dtype=torch.float64, device=dev)
    (L.poisson3d_strip_f64_p64 if ptr64 else L.poisson3d_strip_f64_i32)(_dev(ptr), _stream(ptr), n, r0, r1, _p(ptr), _p(col), _p(val))
    return ptr, col, val


def diffusion3d(n, device="cuda", rows=None, seed=7):
    """Variable-coefficient 7-point operator -div(k grad u) on the n^3 grid (k differs on every face: about 4 N
    distinct values), built in HBM; same pattern, strip convention and nnz as `poisson3d`."""
    L = lib()
    dev = torch.device(device)
    N = n ** 3
    r0, r1 = (0, N) if rows is None else rows
    nnz = L.poisson3d_strip_nnz(n, r0, r1)
    ptr = torch.empty(r1 - r0 + 1, dtype=torch.int32, device=dev)
    col = torch.empty(nnz, dtype=torch.int32, device=dev)
    val = torch.empty(nnz, dtype=torch.float64, device=dev)
    L.diffusion3d_strip_f64_i32(_dev(ptr), _stream(ptr), n, r0, r1, ctypes.c_uint64(seed), _p(ptr), _p(col), _p(val))
    return ptr, col, val


_vector_ordinal = [0]


def device_vector(n, dtype=torch.float64, device="cuda", zero=False):
    """A vector placed as the library places a vex::vector (csrc/runtime.hip vexhip_malloc, round 6): allocations of 64 MiB and more
    start at a multiple of 64 MiB plus a stagger of 0 / 2 / 4 / 6 / 8 MiB -- where x and y of a product lie relative to each other is
    worth up to 11 % of the headline (profiles/r06_xy_gap.json).  The memory is torch's (a view into a larger allocation, which the
    view keeps alive); the offset is the library's rule (vexhip_malloc_placement), not a copy of it."""
    dev = torch.device(device)
    item = torch.empty(0, dtype=dtype).element_size()
    nbytes = n * item
    slack = (64 << 20) + (8 << 20) if nbytes >= (64 << 20) else 0
    raw = torch.empty(nbytes + slack, dtype=torch.uint8, device=dev)
    skip = lib().malloc_placement(nbytes, raw.data_ptr(), _vector_ordinal[0]) if slack else 0
    if slack:
        _vector_ordinal[0] += 1
    v = raw[skip:skip + nbytes].view(dtype)
    if zero:
        v.zero_()
    return v


def fill_hash(t, seed):
    lib().fill_hash(_dev(t), _stream(t), _dtype_code(t), ctypes.c_uint64(seed), _p(t), t.numel())
    return t


# --------------------------------------------------------------------------
# SpMat: the reference's GPU path is hybrid ELL (spmat.hpp:98-103)
# --------------------------------------------------------------------------
class HybridELL:
    """Device-resident ELL(+CSR tail) built from device CSR
    (spmat/hybrid_ell.inl:55-216; device-side conversion sparse/ell.hpp:400-508)."""

    def __init__(self, ptr, col, val, tiled=True, order_mode=0):
        L = lib()
        self.n = n = ptr.numel() - 1
        self.dtype = val.dtype
        dev, s = _dev(val), _stream(val)
        w, tail = ctypes.c_int64(0), ctypes.c_int64(0)
        L.hell_analyze_i32(dev, s, n, _p(ptr), ctypes.byref(w), ctypes.byref(tail))
        self.width, self.tail_nnz = int(w.value), int(tail.value)
        self.pitch = (n + 15) // 16 * 16
        d = val.device
        self.ell_col = torch.empty(self.pitch * self.width, dtype=torch.int32, device=d)
        self.ell_val = torch.empty(self.pitch * self.width, dtype=val.dtype, device=d)
        if self.tail_nnz:
            self.csr_ptr = torch.empty(n + 1, dtype=torch.int32, device=d)
            self.csr_col = torch.empty(self.tail_nnz, dtype=torch.int32, device=d)
            self.csr_val = torch.empty(self.tail_nnz, dtype=val.dtype, device=d)
        else:
            self.csr_ptr = self.csr_col = self.csr_val = None
        fill = L.hell_fill_f64_i32 if val.dtype == torch.float64 else L.hell_fill_f32_i32
        fill(dev, s, n, _p(ptr), _p(col), _p(val), self.width, self.pitch,
             _p(self.ell_col), _p(self.ell_val), _p(self.csr_ptr), _p(self.csr_col), _p(self.csr_val))
        # L2-tiled traversal order for banded matrices (0 blocks = plain order)
        self.order, self.trav = None, _capi.Traversal()
        if self.width and tiled:
            cap = L.hell_order_capacity(n) if order_mode in (1, 2) else 0
            self.order = torch.empty(cap, dtype=torch.int32, device=d) if cap else None
            L.hell_order_i32(dev, s, n, self.width, self.pitch, _p(self.ell_col), order_mode, _p(self.order), cap,
                             ctypes.byref(self.trav))
        self.order_grid = int(self.trav.grid_blocks)

    def mul(self, x, y, alpha=1.0, append=False, tiled=True):
        L = lib()
        f64 = self.dtype == torch.float64
        a = ctypes.c_double(alpha) if f64 else ctypes.c_float(alpha)
        args = (_dev(y), _stream(y), self.n, a, int(bool(append)), self.width, self.pitch,
                _p(self.ell_col), _p(self.ell_val), _p(self.csr_ptr), _p(self.csr_col), _p(self.csr_val),
                _p(x), _p(y))
        if tiled and self.order_grid:
            (L.spmv_hell_ordered_f64_i32 if f64 else L.spmv_hell_ordered_f32_i32)(*args, ctypes.byref(self.trav))
        else:
            (L.spmv_hell_f64_i32 if f64 else L.spmv_hell_f32_i32)(*args)
        return y


class SlicedELL:
    """SELL-512 storage of the ELL part (include/vexhip.h `vexhip_spmv_sell_*`):
    slice-major, one slice = the 512 rows of one workgroup = one contiguous region
    (its columns, then its values).  Width rule and CSR tail are those of hybrid ELL
    (spmat/hybrid_ell.inl:66-216); same arithmetic, same summation order.
    ``codes=True``: when the ELL part uses <= 255 distinct diagonals (banded /
    stencil matrices) the columns are stored as 1-byte diagonal codes ("SELL8",
    9 instead of 12 bytes per fp64 entry); other matrices keep 32-bit columns."""

    def __init__(self, ptr, col, val, tiled=True, order_mode=0, codes=True, value_codes=True):
        L = lib()
        self.n = n = ptr.numel() - 1
        self.dtype = val.dtype
        dev, s, d = _dev(val), _stream(val), val.device
        w, tail = ctypes.c_int64(0), ctypes.c_int64(0)
        L.hell_analyze_i32(dev, s, n, _p(ptr), ctypes.byref(w), ctypes.byref(tail))
        self.width, self.tail_nnz = int(w.value), int(tail.value)
        if not self.width:
            raise Error("SlicedELL needs a non-empty ELL part")
        self.deltas, self.ndeltas = None, -1
        self.values, self.nvalues = None, -1
        self.csr_ptr = self.csr_col = self.csr_val = None
        f64 = val.dtype == torch.float64
        if self.tail_nnz:
            self.csr_ptr = torch.empty(n + 1, dtype=torch.int32, device=d)
            self.csr_col = torch.empty(self.tail_nnz, dtype=torch.int32, device=d)
            self.csr_val = torch.empty(self.tail_nnz, dtype=val.dtype, device=d)
            (L.hell_fill_f64_i32 if f64 else L.hell_fill_f32_i32)(
                dev, s, n, _p(ptr), _p(col), _p(val), self.width, (n + 15) // 16 * 16, None, None,
                _p(self.csr_ptr), _p(self.csr_col), _p(self.csr_val))
        vb = val.element_size()
        # traversal order for banded / stencil matrices (0 blocks = plain order)
        self.order, self.trav = None, _capi.Traversal()
        if codes:
            # banded / stencil matrix (<= 255 distinct diagonals in the ELL part): 1-byte diagonal codes
            deltas = torch.empty(256, dtype=torch.int32, device=d)
            nd = ctypes.c_int(-1)
            L.sell8_analyze_i32(dev, s, n, _p(ptr), _p(col), self.width, _p(deltas), ctypes.byref(nd))
            if nd.value > 0:
                self.deltas, self.ndeltas = deltas, int(nd.value)
                trav = _capi.Traversal()
                if value_codes:
                    # ... and at most 255 distinct VALUES (constant-coefficient stencils): 1-byte value codes too
                    values = torch.empty(256, dtype=val.dtype, device=d)
                    nv = ctypes.c_int(-1)
                    (L.sell8v_analyze_f64_i32 if f64 else L.sell8v_analyze_f32_i32)(
                        dev, s, n, _p(ptr), _p(val), self.width, _p(values), ctypes.byref(nv))
                    if nv.value > 0:
                        self.values, self.nvalues = values, int(nv.value)
                        self.sell = torch.empty(L.sell8v_bytes(n, self.width), dtype=torch.uint8, device=d)
                        (L.sell8v_fill_f64_i32 if f64 else L.sell8v_fill_f32_i32)(
                            dev, s, n, _p(ptr), _p(col), _p(val), self.width, _p(deltas), self.ndeltas, _p(values), self.nvalues,
                            _p(self.sell), ctypes.byref(trav))
                        if tiled and order_mode == 0:
                            self.trav = trav
                        self.order_grid = int(self.trav.grid_blocks)
                        return
                self.sell = torch.empty(L.sell8_bytes(n, self.width, vb), dtype=torch.uint8, device=d)
                (L.sell8_fill_f64_i32 if f64 else L.sell8_fill_f32_i32)(
                    dev, s, n, _p(ptr), _p(col), _p(val), self.width, _p(deltas), self.ndeltas, _p(self.sell),
                    ctypes.byref(trav))
                if tiled and order_mode == 0:
                    self.trav = trav
                self.order_grid = int(self.trav.grid_blocks)
                return
        self.sell = torch.empty(L.sell_bytes(n, self.width, vb), dtype=torch.uint8, device=d)   # one region per slice
        (L.sell_fill_f64_i32 if f64 else L.sell_fill_f32_i32)(
            dev, s, n, _p(ptr), _p(col), _p(val), self.width, _p(self.sell))
        if tiled:
            cap = L.hell_order_capacity(n) if order_mode in (1, 2) else 0
            self.order = torch.empty(cap, dtype=torch.int32, device=d) if cap else None
            L.sell_order_i32(dev, s, n, self.width, vb, _p(self.sell), order_mode, _p(self.order), cap,
                             ctypes.byref(self.trav))
        self.order_grid = int(self.trav.grid_blocks)

    def mul(self, x, y, alpha=1.0, append=False, tiled=True):
        L = lib()
        f64 = self.dtype == torch.float64
        a = ctypes.c_double(alpha) if f64 else ctypes.c_float(alpha)
        use = bool(tiled and self.order_grid)
        if self.values is not None:
            (L.spmv_sell8v_f64_i32 if f64 else L.spmv_sell8v_f32_i32)(
                _dev(y), _stream(y), self.n, a, int(bool(append)), self.width, _p(self.sell), _p(self.deltas), _p(self.values),
                _p(self.csr_ptr), _p(self.csr_col), _p(self.csr_val), _p(x), _p(y),
                ctypes.byref(self.trav) if use else None)
            return y
        if self.deltas is not None:
            (L.spmv_sell8_f64_i32 if f64 else L.spmv_sell8_f32_i32)(
                _dev(y), _stream(y), self.n, a, int(bool(append)), self.width, _p(self.sell), _p(self.deltas),
                _p(self.csr_ptr), _p(self.csr_col), _p(self.csr_val), _p(x), _p(y),
                ctypes.byref(self.trav) if use else None)
            return y
        (L.spmv_sell_f64_i32 if f64 else L.spmv_sell_f32_i32)(
            _dev(y), _stream(y), self.n, a, int(bool(append)), self.width, _p(self.sell),
            _p(self.csr_ptr), _p(self.csr_col), _p(self.csr_val), _p(x), _p(y),
            ctypes.byref(self.trav) if use else None)
        return y

    def mul_multi(self, xs, ys, alpha=1.0, append=False, tiled=True):
        """ys[k] (+)= alpha * A * xs[k] for all k in ONE pass over the matrix per group of four
        right-hand sides (`SpMat * multivector`, spmat.hpp:388-398); each ys[k] is
        bit-identical to ``mul(xs[k], ys[k])``."""
        if len(xs) != len(ys) or not xs:
            raise Error("mul_multi: need as many results as right-hand sides (at least one)")
        L = lib()
        f64 = self.dtype == torch.float64
        a = ctypes.c_double(alpha) if f64 else ctypes.c_float(alpha)
        use = bool(tiled and self.order_grid)
        k = len(xs)
        xp = (ctypes.c_void_p * k)(*[_p(x) for x in xs])
        yp = (ctypes.c_void_p * k)(*[_p(y) for y in ys])
        trav = ctypes.byref(self.trav) if use else None
        if self.values is not None:
            (L.spmm_sell8v_f64_i32 if f64 else L.spmm_sell8v_f32_i32)(
                _dev(ys[0]), _stream(ys[0]), self.n, k, a, int(bool(append)), self.width, _p(self.sell), _p(self.deltas),
                _p(self.values), _p(self.csr_ptr), _p(self.csr_col), _p(self.csr_val), xp, yp, trav)
            return ys
        if self.deltas is not None:
            (L.spmm_sell8_f64_i32 if f64 else L.spmm_sell8_f32_i32)(
                _dev(ys[0]), _stream(ys[0]), self.n, k, a, int(bool(append)), self.width, _p(self.sell), _p(self.deltas),
                _p(self.csr_ptr), _p(self.csr_col), _p(self.csr_val), xp, yp, trav)
        else:
            (L.spmm_sell_f64_i32 if f64 else L.spmm_sell_f32_i32)(
                _dev(ys[0]), _stream(ys[0]), self.n, k, a, int(bool(append)), self.width, _p(self.sell),
                _p(self.csr_ptr), _p(self.csr_col), _p(self.csr_val), xp, yp, trav)
        return ys


class SpMat:
    """vex::SpMat<val_t, col_t, idx_t> on one GPU (spmat.hpp:56-185).

    Built from device CSR arrays.  Formats: ``'sell'`` -- hybrid ELL with the ELL
    part stored slice-major (default for int32 indices: the reference picks hybrid
    ELL for GPU devices, spmat.hpp:98-103; SELL-512 is its MI355X layout),
    ``'hell'`` -- the reference's column-major hybrid ELL, ``'csr'`` -- the CSR
    arrays as given (LDS-staged CSR kernel).  ``apply(x, y, alpha, append)`` has
    the semantics of ``SpMat::apply`` (spmat.hpp:120-121):
    ``y = alpha*A*x`` or ``y += alpha*A*x``.
    """

    _FORMATS = {"sell": _capi.SPMAT_AUTO, "sell8": _capi.SPMAT_SELL8, "sell32": _capi.SPMAT_SELL, "csr": _capi.SPMAT_CSR}

    def __init__(self, ptr, col, val, n_cols=None, fmt="auto", dictionary=True, march=True, plane=True, direct=True, plain_order=False):
        """dictionary=False keeps one code block per slice even when the slices of a value-coded matrix repeat
        (VEXHIP_SPMAT_NO_DICTIONARY: A/B and tests); march=False keeps the pair product where the march product (x window
        of the near diagonals in an LDS ring carried along a run of slices) or the plane product would apply
        (VEXHIP_SPMAT_NO_MARCH); plane=False keeps the march product where the plane product (round 4: two grid lines per
        workgroup walked through the planes, neighbours in registers) or the grid product (the same walk for lines of any
        length) would apply (VEXHIP_SPMAT_NO_PLANE); direct=False builds the SELL-512 storage (slices, dictionary, plans) even where
        the matrix could be stored by grid line straight from the CSR arrays (VEXHIP_SPMAT_NO_GRID_BUILD); plain_order=True deals the
        slices of an unstructured matrix to the XCDs round-robin instead of giving every XCD a contiguous eighth (VEXHIP_SPMAT_PLAIN_ORDER, A/B)."""
        self.ptr, self.col, self.val = ptr, col, val
        self.n = ptr.numel() - 1
        self.m = self.n if n_cols is None else n_cols
        p64 = ptr.dtype == torch.int64 and col.dtype == torch.int32       # 64-bit row pointers, 32-bit columns (round 3)
        i32 = (ptr.dtype == torch.int32 and col.dtype == torch.int32) or p64
        if fmt == "auto":
            fmt = "sell" if i32 else "csr"
        if fmt not in ("sell", "sell8", "sell32", "hell", "csr"):
            raise Error("unknown SpMat format %r" % fmt)
        self.hell, self.handle, self.csr_trav = None, None, None
        self.dictionary_blocks = 0
        self.march = None
        self.plane = None
        self.grid = None
        self.direct = False
        self.dtype = val.dtype
        if fmt == "hell":                        # the reference's column-major hybrid ELL (kept for A/B and sparse::ell)
            self.hell = HybridELL(ptr, col, val)
            self.fmt = self.storage = "hell"
            return
        if not i32 or not val.is_cuda:           # 64-bit indices: the CSR kernel on the arrays as given
            self.fmt = self.storage = "csr"
            return
        # one C-ABI object owns the storage selection (include/vexhip.h vexhip_spmat_*): the C++ vex::SpMat calls the same
        L = lib()
        f64 = val.dtype == torch.float64
        h = ctypes.c_void_p()
        create = ((L.spmat_create_f64_p64 if f64 else L.spmat_create_f32_p64) if p64 else
                  (L.spmat_create_f64_i32 if f64 else L.spmat_create_f32_i32))
        _t0 = time.perf_counter() if os.environ.get("VEXHIP_SETUP_TRACE") else None
        create(
            _dev(val), _stream(val), self.n, _p(ptr), _p(col), _p(val), self._FORMATS[fmt],
            _capi.SPMAT_BORROW_CSR | (0 if dictionary else _capi.SPMAT_NO_DICTIONARY) | (0 if march else _capi.SPMAT_NO_MARCH)
            | (0 if plane else _capi.SPMAT_NO_PLANE) | (0 if direct else _capi.SPMAT_NO_GRID_BUILD)
            | (_capi.SPMAT_PLAIN_ORDER if plain_order else 0)
            | (_capi.SPMAT_SQUARE if self.m >= self.n else 0), ctypes.byref(h))      # x has m >= n elements (vex::SpMat is told n and m: spmat.hpp:56-60)
        self.handle = h
        if _t0 is not None:
            sys.stderr.write("[vexhip set-up] python: create() returned after %.3f ms\n" % ((time.perf_counter() - _t0) * 1e3))
        info = _capi.SpMatInfo()
        L.spmat_get_info(h, ctypes.byref(info))
        self.info = info
        self.storage = _capi.SPMAT_NAMES[info.format]              # sell8v | sell8 | sell32 | csr
        self.product = info.product.decode()                       # the kernel a product launches ...
        self.reason = info.reason.decode()                         # ... and why this storage and that product (csrc/spmat.hip select_product)
        self.dictionary_blocks = int(info.dictionary_blocks)       # > 0: the value-coded slices are stored once per DISTINCT slice
        self.march = ({"lo": int(info.march.lo), "hi": int(info.march.hi), "run": int(info.march.run), "x_last": int(info.march.x_last),
                       "far": [int(info.march.far[k]) for k in range(info.march.nfar)]}
                      if info.march.usable else None)              # not None: apply() runs the march product
        self.plane = ({"lines_per_plane": int(info.plane.lines_per_plane), "planes": int(info.plane.planes), "depth": int(info.plane.depth),
                       "hot_block": int(info.plane.hot_block), "tile": int(info.plane.tile), "store_policy": int(info.plane.store_policy), "x_last": int(info.plane.x_last),
                       "table_pitch": int(info.plane.table_pitch), "flat": int(info.plane.flat)}
                      if info.plane.usable else None)              # not None: apply() runs the plane product (fp64: plane.hip; fp32: plane32.hip)
        g = info.grid
        self.grid = ({"nx": int(g.nx), "lines_per_plane": int(g.lines_per_plane), "planes": int(g.planes), "depth": int(g.depth),
                      "segments": int(g.segments), "segment_rows": int(g.segment_rows), "threads": int(g.threads), "hot_class": int(g.hot_class),
                      "classes": int(g.classes), "store_policy": int(g.store_policy), "flat": int(g.flat), "x_last": int(g.x_last)}
                     if g.usable else None)                        # not None: the matrix is stored by grid line; apply() runs the grid product (fp64; grids of
                                                                   # any line length) unless `plane` is set too (512-point lines: the plane kernel reads the same tables)
        self.direct = bool(g.usable and not info.sell and not info.code_pool)      # stored by grid line straight from the CSR arrays: no SELL-512 slices
        self.fmt = "csr" if self.storage == "csr" else "sell"      # SELL without an ELL part degrades to CSR
        if self.fmt == "sell":
            self.hell = _SellInfo(info)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                lib().spmat_destroy(h)
            except Exception:
                pass

    def rows(self):
        return self.n

    def cols(self):
        return self.m

    def nonzeros(self):
        return self.col.numel()

    def matrix_bytes(self):
        """Bytes of matrix data one product streams (the storage actually read)."""
        if self.handle:
            return int(self.info.matrix_bytes)
        if self.storage == "hell":
            return int(self.hell.width * self.hell.pitch * (4 + self.val.element_size()))
        return int(self.col.numel() * (self.col.element_size() + self.val.element_size()) + self.ptr.numel() * self.ptr.element_size())

    def apply(self, x, y, alpha=1.0, append=False):
        if x.numel() != self.m:
            raise Error("x has %d elements, matrix has %d columns" % (x.numel(), self.m))
        if self.handle:
            f64 = self.dtype == torch.float64
            (lib().spmat_apply_f64 if f64 else lib().spmat_apply_f32)(
                self.handle, _stream(y), ctypes.c_double(alpha) if f64 else ctypes.c_float(alpha), int(bool(append)), _p(x), _p(y))
            return y
        if self.hell is not None:
            return self.hell.mul(x, y, alpha, append)
        return spmv_csr(self.ptr, self.col, self.val, x, y, alpha, append, traversal=self.csr_trav)

    def apply_axpby(self, x, y, alpha, z, beta):
        """y = alpha * A * x + beta * z in one call (vexhip_spmat_apply_axpby_f64: one pass where the product takes the addend);
        z may be x or y, y must not be x."""
        if not self.handle:
            raise Error("apply_axpby: a matrix behind vexhip_spmat only")
        if x.numel() != self.m:
            raise Error("x has %d elements, matrix has %d columns" % (x.numel(), self.m))
        if self.dtype == torch.float64:
            lib().spmat_apply_axpby_f64(self.handle, _stream(y), ctypes.c_double(alpha), _p(x), ctypes.c_double(beta), _p(z), _p(y))
        else:
            lib().spmat_apply_axpby_f32(self.handle, _stream(y), ctypes.c_float(alpha), _p(x), ctypes.c_float(beta), _p(z), _p(y))
        return y

    def apply_multi(self, xs, ys, alpha=1.0, append=False):
        """`Y = alpha * A * X` for a multivector (lists of component vectors): the SELL
        formats read the matrix once for up to four components; other formats loop."""
        for x in xs:
            if x.numel() != self.m:
                raise Error("x has %d elements, matrix has %d columns" % (x.numel(), self.m))
        if len(xs) != len(ys) or not xs:
            raise Error("apply_multi: need as many results as right-hand sides (at least one)")
        if self.handle:
            f64 = self.dtype == torch.float64
            k = len(xs)
            xp = (ctypes.c_void_p * k)(*[_p(x) for x in xs])
            yp = (ctypes.c_void_p * k)(*[_p(y) for y in ys])
            (lib().spmat_apply_multi_f64 if f64 else lib().spmat_apply_multi_f32)(
                self.handle, _stream(ys[0]), k, ctypes.c_double(alpha) if f64 else ctypes.c_float(alpha), int(bool(append)), xp, yp)
            return ys
        for x, y in zip(xs, ys):
            self.apply(x, y, alpha, append)
        return ys

    def __matmul__(self, x):                      # y = A * x
        y = torch.empty(self.n, dtype=x.dtype, device=x.device)
        return self.apply(x, y, 1.0, False)


class _SellInfo:
    """What vexhip_spmat_get_info reports about a SELL storage (read-only view for tests and the bench)."""

    def __init__(self, info):
        self.width, self.tail_nnz = int(info.ell_width), int(info.tail_nnz)
        self.ndeltas, self.nvalues = int(info.ndeltas), int(info.nvalues)
        self.deltas = info.deltas if info.ndeltas > 0 else None        # device addresses (None = not coded)
        self.values = info.values if info.nvalues > 0 else None
        self.order_grid = int(info.traversal.grid_blocks)
        self.sell_bytes = int(info.sell_bytes)


# --------------------------------------------------------------------------
# Reductor
# --------------------------------------------------------------------------
class Reductor:
    """vex::Reductor<T, RDC> for plain-vector operands (reductor.hpp:289-439).
    Blocking, returns a host scalar like the reference (MIN_MAX: a pair)."""
    _OPS = {"SUM": _capi.SUM, "SUM_Kahan": _capi.SUM_KAHAN, "MIN": _capi.MIN, "MAX": _capi.MAX,
            "MIN_MAX": _capi.MIN_MAX}

    def __init__(self, op="SUM", device="cuda"):
        if op not in self._OPS:
            raise Error("unknown reduction %r" % op)
        self.op = op
        self._tmp = None

    def _buffers(self, t):
        if self._tmp is None or self._tmp.device != t.device:
            self._tmp = torch.empty(lib().reduce_tmp_bytes() // 8, dtype=torch.float64, device=t.device)
            self._out = torch.empty(2, dtype=torch.float64, device=t.device)
        return self._tmp, self._out

    def device_result(self, x, unsigned=False):
        """Runs both stages on the device; returns the 1- (or 2-) element device tensor."""
        tmp, out = self._buffers(x)
        lib().reduce(_dev(x), _stream(x), self._OPS[self.op], _dtype_code(x, unsigned), _p(_chk(x, "x")),
                     x.numel(), _p(out), _p(tmp))
        k = 2 if self.op == "MIN_MAX" else 1
        return out.view(torch.uint8)[: k * x.element_size()].view(x.dtype)

    def __call__(self, x, unsigned=False):
        r = self.device_result(x, unsigned).cpu()
        return (r[0].item(), r[1].item()) if self.op == "MIN_MAX" else r[0].item()

    def dot(self, a, b):
        """sum(a*b), examples/benchmark.cpp:224-246."""
        if self.op != "SUM":
            raise Error("dot is a SUM reduction")
        tmp, out = self._buffers(a)
        lib().reduce_dot(_dev(a), _stream(a), _dtype_code(a), _p(a), _p(b), a.numel(), _p(out), _p(tmp))
        return out.view(torch.uint8)[: a.element_size()].view(a.dtype).cpu()[0].item()


# --------------------------------------------------------------------------
# scan / sort
# --------------------------------------------------------------------------
def _scan(inp, out, exclusive, init, unsigned):
    if out is None:
        out = torch.empty_like(inp)
    _chk(inp, "input"); _chk(out, "output")
    if out.numel() != inp.numel() or out.dtype != inp.dtype:
        raise Error("scan: input and output must have the same size and type")
    L = lib()
    code = _dtype_code(inp, unsigned)
    n = inp.numel()
    tmp = torch.empty(max(1, L.scan_tmp_bytes(code, n)), dtype=torch.uint8, device=inp.device)
    host_init = None
    if exclusive:
        ct = {_capi.F64: ctypes.c_double, _capi.F32: ctypes.c_float, _capi.I32: ctypes.c_int32,
              _capi.U32: ctypes.c_uint32, _capi.I64: ctypes.c_int64, _capi.U64: ctypes.c_uint64}[code]
        host_init = ctypes.byref(ct(init))
    L.scan(_dev(inp), _stream(inp), code, int(exclusive), host_init, _p(inp), _p(out), n, _p(tmp))
    return out


def inclusive_scan(inp, out=None, unsigned=False):
    """vex::inclusive_scan(in, out) with vex::plus (scan.hpp:461-469). In-place allowed."""
    return _scan(inp, out, False, 0, unsigned)


def exclusive_scan(inp, out=None, init=0, unsigned=False):
    """vex::exclusive_scan(in, out, init) (scan.hpp:510-518)."""
    return _scan(inp, out, True, init, unsigned)


def _sort(keys, vals, descending, unsigned):
    _chk(keys, "keys")
    L = lib()
    n = keys.numel()
    code = _dtype_code(keys, unsigned)
    ktmp = torch.empty_like(keys)
    vb, vtmp = 0, None
    if vals is not None:
        _chk(vals, "vals")
        if vals.numel() != n:
            raise Error("sort_by_key: keys and values differ in size")
        vb = vals.element_size()
        if vb not in (4, 8):
            raise Error("sort_by_key: values must be 4 or 8 bytes wide")
        vtmp = torch.empty_like(vals)
    tmp = torch.empty(max(1, L.sort_tmp_bytes(code, n)), dtype=torch.uint8, device=keys.device)
    L.sort(_dev(keys), _stream(keys), code, int(bool(descending)), _p(keys), _p(ktmp), vb, _p(vals), _p(vtmp),
           n, _p(tmp))
    _last_sort_status[:] = [keys, tmp, n]
    return keys if vals is None else (keys, vals)


_last_sort_status = [None, None, 0]


def sort_status():
    """(tiles ranked a second time, tiles dropped) of the last sort / sort_by_key (include/vexhip.h vexhip_sort_status; waits for it).
    A dropped tile -- the input changed while the sort ran -- raises."""
    keys, tmp, n = _last_sort_status
    if keys is None:
        return 0, 0
    a, b = ctypes.c_int64(), ctypes.c_int64()
    lib().sort_status(_dev(keys), _stream(keys), n, _p(tmp), ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def sort(keys, descending=False, unsigned=False):
    """vex::sort(keys[, vex::greater]) -- in place, stable (sort.hpp:2158-2167)."""
    return _sort(keys, None, descending, unsigned)


def sort_by_key(keys, vals, descending=False, unsigned=False):
    """vex::sort_by_key(keys, vals) -- in place, stable (sort.hpp:2170-2182)."""
    return _sort(keys, vals, descending, unsigned)


# --------------------------------------------------------------------------
# FFT (vexcl/fft.hpp:69-148)
# --------------------------------------------------------------------------
FORWARD, INVERSE, NONE = 0, 1, 2        # vex::fft::direction


def fft_best_size(n):
    """vex::fft::planner::best_size: smallest 2^a 3^b 5^c 7^d >= n."""
    return int(lib().fft_best_size(n))


class FFT:
    """vex::FFT<cl_double2> / <cl_float2> over a complex torch tensor: ``sizes`` row-major, one direction per
    dimension (NONE = batch).  Like the reference, inverse dimensions are scaled by 1/n (fft/plan.hpp:236-241)."""

    def __init__(self, sizes, dirs=FORWARD, dtype=torch.complex128, device="cuda"):
        self.sizes = [int(s) for s in (sizes if hasattr(sizes, "__len__") else [sizes])]
        self.dirs = [int(d) for d in (dirs if hasattr(dirs, "__len__") else [dirs] * len(self.sizes))]
        if len(self.dirs) != len(self.sizes):
            raise Error("one direction per dimension is required")
        if dtype not in (torch.complex128, torch.complex64):
            raise Error("only complex64 / complex128 data are supported")
        self.dtype = dtype
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise Error("vexcl_amd.FFT runs on the GPU only; there is no CPU fallback")
        self.dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.scale = 1.0
        for s, d in zip(self.sizes, self.dirs):
            if d == INVERSE:
                self.scale /= s
        sz = (ctypes.c_size_t * len(self.sizes))(*self.sizes)
        dr = (ctypes.c_int * len(self.dirs))(*self.dirs)
        self._plan = ctypes.c_void_p()
        lib().fft_plan_create(self.dev, _capi.F64 if dtype == torch.complex128 else _capi.F32, len(self.sizes), sz, dr,
                              ctypes.byref(self._plan))

    def steps(self):
        r, t, o = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib().fft_plan_steps(self._plan, ctypes.byref(r), ctypes.byref(t), ctypes.byref(o))
        return r.value, t.value, o.value

    def __call__(self, x, out=None, scaled=True):
        total = 1
        for s in self.sizes:
            total *= s
        if x.dtype != self.dtype or x.numel() != total:
            raise Error("FFT plan is for %d %s elements" % (total, self.dtype))
        _chk(x, "x")
        if out is None:
            out = torch.empty_like(x)
        if total:
            lib().fft_exec(self._plan, _stream(x), _p(x), _p(out))
        if scaled and self.scale != 1.0:
            out.mul_(self.scale)
        return out

    def __del__(self):
        try:
            if self._plan:
                lib().fft_plan_destroy(self._plan)
                self._plan = ctypes.c_void_p()
        except Exception:
            pass
