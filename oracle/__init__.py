"""CPU oracle for the vexcl hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; the product (``vexcl_amd``) never does.

PARITY UNPINNED: the reference ships no golden vectors and cannot be built in
this image (Boost / CPU-OpenCL absent) -- see ``vex_oracle.c`` and DESIGN.md.

The arithmetic lives in ``vex_oracle.c`` (plain C, ``-ffp-contract=off``); this
module is its ctypes binding plus numpy restatements of the integer-only
pieces (partitioning, local/remote split, scan, stable sort), each citing the
reference file:line it follows.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvexoracle.so")


def build(force=False):
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "vex_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.vxo_poisson3d_nnz.restype = ctypes.c_int64
        _lib.vxo_poisson3d_nnz.argtypes = [ctypes.c_int64]
        _lib.vxo_hell_width_i32.restype = ctypes.c_int64
        _lib.vxo_hell_pitch.restype = ctypes.c_int64
        _lib.vxo_hell_build_f64_i32.restype = ctypes.c_int64
        _lib.vxo_random_matrix_f64_i32.restype = ctypes.c_int64
        for f in ("vxo_sum_f64", "vxo_sum_kahan_f64", "vxo_dot_kahan_f64"):
            getattr(_lib, f).restype = ctypes.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


_i64 = ctypes.c_int64
_f64 = ctypes.c_double


# --------------------------------------------------------------------------
# Poisson matrix (examples/benchmark.cpp:364-415)
# --------------------------------------------------------------------------
def poisson3d(n, index_dtype=np.int32, val_dtype=np.float64):
    L = lib()
    N = n ** 3
    nnz = L.vxo_poisson3d_nnz(_i64(n))
    ptr = np.empty(N + 1, dtype=index_dtype)
    col = np.empty(nnz, dtype=index_dtype)
    val = np.empty(nnz, dtype=val_dtype)
    if index_dtype == np.int32 and val_dtype == np.float64:
        L.vxo_poisson3d_csr_i32(_i64(n), _p(ptr), _p(col), _p(val))
    elif index_dtype == np.int64 and val_dtype == np.float64:
        L.vxo_poisson3d_csr_i64(_i64(n), _p(ptr), _p(col), _p(val))
    elif index_dtype == np.int32 and val_dtype == np.float32:
        L.vxo_poisson3d_csr_f32_i32(_i64(n), _p(ptr), _p(col), _p(val))
    else:
        raise TypeError("unsupported oracle dtype combination")
    return ptr, col, val


def poisson3d_nnz(n):
    return int(lib().vxo_poisson3d_nnz(_i64(n)))


# --------------------------------------------------------------------------
# CSR SpMV (vexcl/spmat/csr.inl:163-170; tests/spmv.cpp:28-32)
# --------------------------------------------------------------------------
def spmv_csr(ptr, col, val, x, y=None, alpha=1.0, append=False, omp=False):
    """y (=|+=) alpha * A x; returns y (a new array when y is None)."""
    n = len(ptr) - 1
    L = lib()
    x = np.ascontiguousarray(x)
    if y is None:
        assert not append
        y = np.empty(n, dtype=val.dtype)
    if val.dtype == np.float64 and ptr.dtype == np.int32:
        fn = L.vxo_spmv_csr_f64_i32_omp if omp else L.vxo_spmv_csr_f64_i32
        fn(_i64(n), _f64(alpha), ctypes.c_int(int(append)), _p(ptr), _p(col), _p(val), _p(x), _p(y))
    elif val.dtype == np.float64 and ptr.dtype == np.int64:
        L.vxo_spmv_csr_f64_i64(_i64(n), _f64(alpha), ctypes.c_int(int(append)),
                               _p(ptr), _p(col), _p(val), _p(x), _p(y))
    elif val.dtype == np.float32 and ptr.dtype == np.int32:
        L.vxo_spmv_csr_f32_i32(_i64(n), ctypes.c_float(alpha), ctypes.c_int(int(append)),
                               _p(ptr), _p(col), _p(val), _p(x), _p(y))
    else:
        raise TypeError("unsupported oracle dtype combination")
    return y


def num_threads():
    return int(lib().vxo_num_threads())


def spmv_abs_bound(ptr, col, val, x):
    """sum_j |a_ij x_j| per row: the scale the fp64 tolerance is stated against
    (SURVEY section 7 'Parity definition')."""
    return spmv_csr(ptr, col, np.abs(val), np.abs(x))


# --------------------------------------------------------------------------
# Hybrid ELL (vexcl/spmat/hybrid_ell.inl:66-114,138-198,238-269)
# --------------------------------------------------------------------------
def hell_width(ptr):
    return int(lib().vxo_hell_width_i32(_i64(len(ptr) - 1), _p(ptr)))


def hell_build(ptr, col, val, width=None):
    L = lib()
    n = len(ptr) - 1
    if width is None:
        width = hell_width(ptr)
    pitch = int(L.vxo_hell_pitch(_i64(n)))
    ell_col = np.empty(pitch * width, dtype=np.int32)
    ell_val = np.empty(pitch * width, dtype=np.float64)
    csr_ptr = np.empty(n + 1, dtype=np.int32)
    tail = L.vxo_hell_build_f64_i32(_i64(n), _p(ptr), _p(col), _p(val), _i64(width), _i64(pitch),
                                    _p(ell_col), _p(ell_val), _p(csr_ptr), None, None)
    csr_col = np.empty(tail, dtype=np.int32)
    csr_val = np.empty(tail, dtype=np.float64)
    L.vxo_hell_build_f64_i32(_i64(n), _p(ptr), _p(col), _p(val), _i64(width), _i64(pitch),
                             _p(ell_col), _p(ell_val), _p(csr_ptr), _p(csr_col), _p(csr_val))
    return dict(n=n, width=width, pitch=pitch, ell_col=ell_col, ell_val=ell_val,
                csr_ptr=csr_ptr, csr_col=csr_col, csr_val=csr_val, tail=int(tail))


def spmv_hell(h, x, y=None, alpha=1.0, append=False):
    n = h["n"]
    if y is None:
        y = np.empty(n, dtype=np.float64)
    lib().vxo_spmv_hell_f64_i32(_i64(n), _f64(alpha), ctypes.c_int(int(append)),
                                _i64(h["width"]), _i64(h["pitch"]),
                                _p(h["ell_col"]), _p(h["ell_val"]),
                                _p(h["csr_ptr"]) if h["tail"] else None,
                                _p(h["csr_col"]), _p(h["csr_val"]), _p(x), _p(y))
    return y


# --------------------------------------------------------------------------
# Elementwise / reductions
# --------------------------------------------------------------------------
def ew_mul_add_sin(b, c, d):
    a = np.empty_like(b)
    lib().vxo_ew_mul_add_sin_f64(_i64(len(b)), _p(b), _p(c), _p(d), _p(a))
    return a


def sum_kahan(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return float(lib().vxo_sum_kahan_f64(_p(x), _i64(len(x))))


def dot_kahan(a, b):
    return float(lib().vxo_dot_kahan_f64(_p(a), _p(b), _i64(len(a))))


# --------------------------------------------------------------------------
# Generators (shape of tests/random_matrix.hpp / tests/random_vector.hpp)
# --------------------------------------------------------------------------
def random_f64(seed, n):
    x = np.empty(n, dtype=np.float64)
    lib().vxo_random_f64(ctypes.c_uint64(seed), _i64(n), _p(x))
    return x


def random_i32(seed, n, lo=0, hi=100):
    x = np.empty(n, dtype=np.int32)
    lib().vxo_random_i32(ctypes.c_uint64(seed), _i64(n), ctypes.c_int32(lo), ctypes.c_int32(hi), _p(x))
    return x


def random_u32(seed, n):
    x = np.empty(n, dtype=np.uint32)
    lib().vxo_random_u32(ctypes.c_uint64(seed), _i64(n), _p(x))
    return x


def random_matrix(seed, n, m, nnz_per_row=16, empty_tail=0):
    ptr = np.empty(n + 1, dtype=np.int32)
    col = np.empty(max(1, n * nnz_per_row), dtype=np.int32)
    val = np.empty(max(1, n * nnz_per_row), dtype=np.float64)
    nnz = lib().vxo_random_matrix_f64_i32(ctypes.c_uint64(seed), _i64(n), _i64(m), _i64(nnz_per_row),
                                          _i64(empty_tail), _p(ptr), _p(col), _p(val))
    return ptr, col[:nnz].copy(), val[:nnz].copy()


# --------------------------------------------------------------------------
# Partitioning (vexcl/vector.hpp:131-167 with equal_weights, util.hpp:91-93)
# --------------------------------------------------------------------------
def partition(n, ndev, weights=None):
    if weights is None:
        weights = [1.0] * ndev
    part = [0]
    if ndev > 1:
        cum = np.concatenate([[0.0], np.cumsum(np.asarray(weights, dtype=np.float64))])
        for d in range(1, ndev):
            b = int(n * cum[d] / cum[-1])
            b = (b + 15) // 16 * 16
            part.append(min(n, b))
    part.append(n)
    return part


# --------------------------------------------------------------------------
# Multi-device split (vexcl/spmat.hpp:291-378 setup_exchange;
# spmat/csr.inl:92-131 local/remote sub-matrices; apply spmat.hpp:120-185)
# --------------------------------------------------------------------------
def split_rows(ptr, col, val, n_cols, ndev):
    """Returns per-device dicts with local/remote CSR parts and the exchange
    lists, exactly as the reference lays them out:
      * ghost columns of device d = sorted set of its non-local columns;
      * remote sub-matrix columns are renumbered to the rank in that set;
      * cols_to_send = sorted union over devices of all ghosts (global ids),
        cols_to_recv[d] = positions of d's ghosts inside that union."""
    n = len(ptr) - 1
    part = partition(n, ndev)
    cpart = partition(n_cols, ndev)
    devs = []
    union = set()
    for d in range(ndev):
        r0, r1 = part[d], part[d + 1]
        c0, c1 = cpart[d], cpart[d + 1]
        j0, j1 = int(ptr[r0]), int(ptr[r1])
        c = col[j0:j1].astype(np.int64)
        v = val[j0:j1]
        rows = np.repeat(np.arange(r1 - r0), np.diff(ptr[r0:r1 + 1]).astype(np.int64))
        is_loc = (c >= c0) & (c < c1)
        ghosts = np.unique(c[~is_loc])
        union.update(ghosts.tolist())

        def csr_of(mask, cols):
            cnt = np.bincount(rows[mask], minlength=r1 - r0)
            p = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
            return p, cols.astype(np.int32), v[mask].copy()

        loc = csr_of(is_loc, c[is_loc] - c0)
        rem = csr_of(~is_loc, np.searchsorted(ghosts, c[~is_loc]))
        devs.append(dict(rows=(r0, r1), cols=(c0, c1), loc=loc, rem=rem, ghosts=ghosts))
    send = np.array(sorted(union), dtype=np.int64)
    for d in range(ndev):
        devs[d]["cols_to_recv"] = np.searchsorted(send, devs[d]["ghosts"])
        c0, c1 = devs[d]["cols"]
        lo, hi = np.searchsorted(send, c0), np.searchsorted(send, c1)
        devs[d]["cidx"] = (int(lo), int(hi))
        devs[d]["cols_to_send"] = (send[lo:hi] - c0).astype(np.int32)
    return dict(part=part, col_part=cpart, devs=devs, cols_to_send=send)


def spmv_split(split, x, y=None, alpha=1.0, append=False):
    """The 5-phase apply of spmat.hpp:120-185, on the host."""
    part, cpart, devs = split["part"], split["col_part"], split["devs"]
    n = part[-1]
    if y is None:
        y = np.zeros(n, dtype=np.float64)
        append = False
    rx = np.empty(len(split["cols_to_send"]), dtype=np.float64)
    for d, D in enumerate(devs):                        # gather + "D2H"
        lo, hi = D["cidx"]
        rx[lo:hi] = x[cpart[d]:cpart[d + 1]][D["cols_to_send"]]
    for d, D in enumerate(devs):
        r0, r1 = D["rows"]
        xl = np.ascontiguousarray(x[cpart[d]:cpart[d + 1]])
        yl = np.ascontiguousarray(y[r0:r1])
        p, c, v = D["loc"]
        if len(v):
            spmv_csr(p, c, v, xl, yl, alpha, append)
        elif not append:
            yl[:] = 0                                   # csr.inl:196-199
        p, c, v = D["rem"]
        if len(v):
            ghost = np.ascontiguousarray(rx[D["cols_to_recv"]])
            spmv_csr(p, c, v, ghost, yl, alpha, True)
        y[r0:r1] = yl
    return y


# --------------------------------------------------------------------------
# scan (semantics = std::partial_sum, vexcl/scan.hpp:427-518; tests/scan.cpp)
# sort (semantics = std::stable_sort, vexcl/sort.hpp:2158-2182; tests/sort.cpp)
# --------------------------------------------------------------------------
def inclusive_scan(x):
    return np.cumsum(x, dtype=x.dtype)


def exclusive_scan(x, init=0):
    out = np.empty_like(x)
    if len(x):
        out[0] = init
        if len(x) > 1:
            with np.errstate(over="ignore"):
                out[1:] = np.cumsum(x[:-1], dtype=x.dtype) + np.asarray(init, dtype=x.dtype)
    return out


def stable_sort(keys):
    return np.sort(keys, kind="stable")


def stable_sort_by_key(keys, vals):
    p = np.argsort(keys, kind="stable")
    return keys[p], vals[p]
