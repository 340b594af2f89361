"""CPU oracle for the vexcl hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; the product (``vexcl_amd``) never does.

PARITY UNPINNED: the reference ships no golden vectors and cannot be built in
this image (Boost / CPU-OpenCL absent) -- see ``vex_oracle.c`` and DESIGN.md.
One exception: the Philox / Threefry generators at the end of this file ARE pinned,
by the known-answer vectors published with those algorithms (tests/test_oracle.py).

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


def diffusion3d(n, seed=7):
    """Variable-coefficient 7-point operator (about 4 N distinct values); restates vexhip_diffusion3d_* bit for bit."""
    L = lib()
    N = n ** 3
    nnz = L.vxo_poisson3d_nnz(_i64(n))
    ptr = np.empty(N + 1, dtype=np.int32)
    col = np.empty(nnz, dtype=np.int32)
    val = np.empty(nnz, dtype=np.float64)
    L.vxo_diffusion3d_csr_i32(_i64(n), ctypes.c_uint64(seed), _p(ptr), _p(col), _p(val))
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


def cpu_baseline_poisson(n, seconds, single_reps=1):
    """bench.py's CPU leg: the reference's CPU-device csr_spmv shape (8 x threads contiguous row chunks, OpenMP) on the
    n^3 Poisson matrix, arrays first-touched by the threads that use them; plus the single-thread loop of the
    reference harness (examples/benchmark.cpp:447-453).  Returns a dict, or None if the arrays do not fit."""
    L = lib()
    out = (ctypes.c_double * 5)()
    L.vxo_cpu_baseline_poisson.restype = ctypes.c_int
    rc = L.vxo_cpu_baseline_poisson(_i64(n), ctypes.c_double(seconds), ctypes.c_int(single_reps), out)
    if rc != 0:
        return None
    return {"seconds_per_product": out[0], "threads": int(out[1]), "single_thread_seconds_per_product": out[2],
            "sum_y": out[3], "products": int(out[4])}


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
# Row-pair placement of the SELL storages of THIS implementation (not a
# reference format): vexcl_amd/csrc/pairing.hpp.  The first w entries of rows
# 2k and 2k+1 are merged by diagonal (column - row); equal diagonals share an
# ELL column, otherwise the smaller takes the column alone.  If the merged
# list is longer than w the pair keeps the plain packing (entry j in column j).
# The empty half of a column is PAD_SAFE when a 16-byte load of x that starts
# at (partner column - partner half) stays inside [0, max_col], else PAD_UNSAFE.
# Restated in plain Python so that the layout tests have something to compare.
# --------------------------------------------------------------------------
def sell_pair_slots(ptr, col, row0, w, max_col):
    """Returns for rows row0, row0+1 two lists of length w holding CSR entry offsets, 'safe' or 'unsafe'."""
    n = len(ptr) - 1
    ent = []
    for q in (0, 1):
        r = row0 + q
        b, e = (int(ptr[r]), int(ptr[r + 1])) if r < n else (0, 0)
        ent.append(list(range(b, min(e, b + w))))
    diag = [[int(col[k]) - (row0 + q) for k in ent[q]] for q in (0, 1)]
    slots, pa, pb = [], 0, 0
    while pa < len(ent[0]) or pb < len(ent[1]):
        if pa < len(ent[0]) and pb < len(ent[1]):
            da, db = diag[0][pa], diag[1][pb]
            if da == db:
                slots.append((ent[0][pa], ent[1][pb])); pa += 1; pb += 1
            elif da < db:
                slots.append((ent[0][pa], None)); pa += 1
            else:
                slots.append((None, ent[1][pb])); pb += 1
        elif pa < len(ent[0]):
            slots.append((ent[0][pa], None)); pa += 1
        else:
            slots.append((None, ent[1][pb])); pb += 1
    if len(slots) > w:          # plain packing
        slots = [(ent[0][j] if j < len(ent[0]) else None, ent[1][j] if j < len(ent[1]) else None) for j in range(w)]
    slots += [(None, None)] * (w - len(slots))
    out = [[], []]
    for e0, e1 in slots:
        for q, (mine, other) in enumerate(((e0, e1), (e1, e0))):
            if mine is not None:
                out[q].append(mine)
            elif other is None:
                out[q].append("safe")
            else:
                first = int(col[other]) - (1 - q)
                out[q].append("safe" if first >= 0 and first + 1 <= max_col else "unsafe")
    return out


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


# ---------------------------------------------------------------------------
# Counter-based random numbers (vexcl/random.hpp:60-275; generators vexcl/random/philox.hpp:95-160,
# vexcl/random/threefry.hpp:150-215 -- Philox / Threefry of Salmon, Moraes, Dror, Shaw, SC'11).
# numpy restatement; pinned by the known-answer vectors published with the algorithms
# (tests/test_oracle.py).  Word arrays are uint32 with the counter words in the last axis.
# ---------------------------------------------------------------------------
def _mulhilo(a, b):
    """(high, low) words of a * b for uint32 or uint64 arrays."""
    if b.dtype == np.uint32:
        p = np.uint64(a) * b.astype(np.uint64)
        return (p >> np.uint64(32)).astype(np.uint32), p.astype(np.uint32)
    m = np.uint64(0xFFFFFFFF)
    a = np.uint64(a)
    a0, a1 = a & m, a >> np.uint64(32)
    b0, b1 = b & m, b >> np.uint64(32)
    with np.errstate(over="ignore"):
        lo = a * b
        t = a1 * b0 + ((a0 * b0) >> np.uint64(32))          # < 2^64: (2^32-1)^2 + 2^32 - 1
        w = a0 * b1 + (t & m)
        hi = a1 * b1 + (t >> np.uint64(32)) + (w >> np.uint64(32))
    return hi, lo


_PHILOX = {
    np.dtype(np.uint32): dict(weyl=(0x9E3779B9, 0xBB67AE85), m2=0xD256D193, m4=(0xD2511F53, 0xCD9E8D57)),
    np.dtype(np.uint64): dict(weyl=(0x9E3779B97F4A7C15, 0xBB67AE8584CAA73B), m2=0xD2B74407B1CE6E93,
                              m4=(0xD2E7470EE14C6C93, 0xCA5A826395121157)),
}


def philox(ctr, key, rounds=10):
    """Philox-NxW-R, N = ctr.shape[-1] in (2, 4), W = 32 or 64 bits by the dtype of ctr (uint32 unless it is
    uint64); key has N/2 words.  Returns the output words."""
    dt = np.dtype(np.uint64) if np.asarray(ctr).dtype == np.uint64 else np.dtype(np.uint32)
    ctr = np.array(ctr, dtype=dt, copy=True)
    key = np.array(np.broadcast_to(np.asarray(key, dtype=dt), ctr.shape[:-1] + (ctr.shape[-1] // 2,)), copy=True)
    n = ctr.shape[-1]
    k = _PHILOX[dt]
    W = tuple(dt.type(w) for w in k["weyl"])
    with np.errstate(over="ignore"):
        for r in range(rounds):
            if r:
                for i in range(n // 2):
                    key[..., i] += W[i]
            if n == 2:
                hi, lo = _mulhilo(dt.type(k["m2"]), ctr[..., 0])
                ctr[..., 0], ctr[..., 1] = hi ^ key[..., 0] ^ ctr[..., 1], lo
            else:
                hi0, lo0 = _mulhilo(dt.type(k["m4"][0]), ctr[..., 0])
                hi1, lo1 = _mulhilo(dt.type(k["m4"][1]), ctr[..., 2])
                c0 = hi1 ^ ctr[..., 1] ^ key[..., 0]
                c2 = hi0 ^ ctr[..., 3] ^ key[..., 1]
                ctr[..., 0], ctr[..., 1], ctr[..., 2], ctr[..., 3] = c0, lo1, c2, lo0
    return ctr


_THREEFRY_ROT = {
    (32, 2): (13, 15, 26, 6, 17, 29, 16, 24), (32, 4): (10, 26, 11, 21, 13, 27, 23, 5, 6, 20, 17, 11, 25, 10, 18, 20),
    (64, 2): (16, 42, 12, 31, 16, 32, 24, 21), (64, 4): (14, 16, 52, 57, 23, 40, 5, 37, 25, 33, 46, 12, 58, 22, 32, 32),
}


def threefry(ctr, key, rounds=20):
    """Threefry-NxW-R as the reference generates it (W by the dtype of ctr, as for philox): N = 2 is the published
    Threefry-2xW; N = 4 mixes the word pairs (0,1) and (2,3) WITHOUT Threefish's word permutation
    (threefry.hpp:178-215)."""
    dt = np.dtype(np.uint64) if np.asarray(ctr).dtype == np.uint64 else np.dtype(np.uint32)
    bits = dt.itemsize * 8
    ctr = np.array(ctr, dtype=dt, copy=True)
    n = ctr.shape[-1]
    key = np.array(np.broadcast_to(np.asarray(key, dtype=dt), ctr.shape[:-1] + (n,)), copy=True)
    rot = _THREEFRY_ROT[(bits, n)]
    p = np.full(ctr.shape[:-1], 0x1BD11BDA if bits == 32 else 0x1BD11BDAA9FC1A22, dtype=dt)
    for i in range(n):
        p = p ^ key[..., i]
    ks = [key[..., i] for i in range(n)] + [p]

    def rotl(x, b):
        return (x << dt.type(b)) | (x >> dt.type(bits - b))

    with np.errstate(over="ignore"):
        for i in range(n):
            ctr[..., i] += ks[i]
        for r in range(rounds):
            if n == 2:
                ctr[..., 0] += ctr[..., 1]; ctr[..., 1] = rotl(ctr[..., 1], rot[r % 8]); ctr[..., 1] ^= ctr[..., 0]
            else:
                b = 2 * (r % 8)
                r0, r1 = rot[b + (r % 2)], rot[b + ((r + 1) % 2)]
                ctr[..., 0] += ctr[..., 1]; ctr[..., 1] = rotl(ctr[..., 1], r0); ctr[..., 1] ^= ctr[..., 0]
                ctr[..., 2] += ctr[..., 3]; ctr[..., 3] = rotl(ctr[..., 3], r1); ctr[..., 3] ^= ctr[..., 2]
            if (r + 1) % 4 == 0:
                j = r // 4 + 1
                for i in range(n):
                    ctr[..., i] += ks[(j + i) % (n + 1)]
                ctr[..., n - 1] += dt.type(j)
    return ctr


def _rng_words(idx, seed, n, generator, bits=32):
    dt = np.uint32 if bits == 32 else np.uint64
    idx = np.asarray(idx, dtype=np.uint64)
    c = np.empty(idx.shape + (n,), dtype=dt)
    for i in range(0, n, 2):
        c[..., i] = idx.astype(dt)                        # (word)prm1
        c[..., i + 1] = dt(np.uint64(seed) & np.uint64(0xFFFFFFFF if bits == 32 else 0xFFFFFFFFFFFFFFFF))
    gen = {"philox": philox, "threefry": threefry}[generator]
    nk = n // 2 if generator == "philox" else n
    return gen(c, np.full(nk, 0x12345678, dtype=dt))


def random_vector(idx, seed, dtype, length, generator="philox"):
    """vex::Random<cl_<T>N, Generator>()(idx, seed): shape idx.shape + (length,) (random.hpp:85-152).
    Outputs below 32 bytes come from 32-bit words (two for up to 8 bytes, else four), 32-byte outputs from four
    64-bit words; the words are the bytes of the result; reals are then word / (2^W - 1) of their own width."""
    dtype = np.dtype(dtype)
    nbytes = dtype.itemsize * length
    assert nbytes in (1, 2, 4, 8, 16, 32)
    bits = 32 if nbytes < 32 else 64
    w = _rng_words(idx, seed, 2 if nbytes <= 8 else 4, generator, bits)
    raw = np.ascontiguousarray(w).view(np.uint8)[..., :nbytes]
    if dtype.kind == "f":
        ints = np.ascontiguousarray(raw).view(np.uint32 if dtype.itemsize == 4 else np.uint64)
        if dtype.itemsize == 4:
            return ints.astype(np.float32) / np.float32(4294967295.0)
        return ints.astype(np.float64) / np.float64(18446744073709551615.0)
    return np.ascontiguousarray(raw).view(dtype)


def random_uniform(idx, seed, dtype=np.float64, generator="philox"):
    """vex::Random<T, Generator>()(idx, seed) (random.hpp:60-150)."""
    w = _rng_words(idx, seed, 2, generator)
    dtype = np.dtype(dtype)
    u64 = (w[..., 1].astype(np.uint64) << np.uint64(32)) | w[..., 0].astype(np.uint64)
    if dtype == np.float32:
        return w[..., 0].astype(np.float32) / np.float32(4294967295.0)
    if dtype == np.float64:
        return u64.astype(np.float64) / np.float64(18446744073709551615.0)
    if dtype.itemsize == 4:
        return w[..., 0].view(dtype) if dtype != np.uint32 else w[..., 0]
    return u64.view(dtype) if dtype != np.uint64 else u64


def random_normal(idx, seed, dtype=np.float64, generator="philox"):
    """vex::RandomNormal<T, Generator>()(idx, seed): Box-Muller (random.hpp:155-275)."""
    if np.dtype(dtype) == np.float32:
        w = _rng_words(idx, seed, 2, generator)
        u0 = w[..., 0].astype(np.float32) / np.float32(4294967295.0)
        u1 = w[..., 1].astype(np.float32) / np.float32(4294967295.0)
        return (np.sqrt(np.float32(-2) * np.log(u0)) * np.cos(np.float32(np.pi) * (np.float32(2) * u1))).astype(np.float32)
    w = _rng_words(idx, seed, 4, generator)
    a = (w[..., 1].astype(np.uint64) << np.uint64(32)) | w[..., 0].astype(np.uint64)
    b = (w[..., 3].astype(np.uint64) << np.uint64(32)) | w[..., 2].astype(np.uint64)
    u0 = a.astype(np.float64) / 18446744073709551615.0
    u1 = b.astype(np.float64) / 18446744073709551615.0
    return np.sqrt(-2 * np.log(u0)) * np.cos(np.pi * (2 * u1))


# ---- FFT (vexcl/fft.hpp:42-66, fft/plan.hpp:214-257) ---------------------------------------------------
FFT_FORWARD, FFT_INVERSE, FFT_NONE = 0, 1, 2


def dft_definition(x, inverse=False):
    """The transform the reference computes along one dimension, by its definition:
    X[k] = sum_j x[j] exp(-+ 2 pi i j k / n); the inverse divided by n (plan.hpp:236-241: scale = 1 / prod of the
    inverse dimensions).  O(n^2), in extended precision: pins `fft_nd` for small n."""
    x = np.asarray(x, dtype=np.clongdouble)
    n = x.shape[-1]
    j = np.arange(n)
    ang = (-2 if not inverse else 2) * np.pi * ((np.outer(j, j) % n).astype(np.longdouble)) / np.longdouble(n)
    w = np.cos(ang) + 1j * np.sin(ang)
    y = x @ w
    return (y / n if inverse else y).astype(np.complex128)


def fft_nd(x, sizes, dirs):
    """vex::FFT over a row-major array of shape `sizes`, one direction per dimension (FFT_NONE: batch dimension).
    Real input is extended with a zero imaginary part (fft.hpp:44-48).  numpy's pocketfft per axis."""
    a = np.asarray(x).reshape(sizes).astype(np.complex128)
    for ax, d in enumerate(dirs):
        if d == FFT_FORWARD:
            a = np.fft.fft(a, axis=ax)
        elif d == FFT_INVERSE:
            a = np.fft.ifft(a, axis=ax)
    return a.reshape(-1)


def fft_best_size(n):
    """fft::planner::best_size with the primes 2, 3, 5, 7: the smallest such number >= n (plan.hpp:103-128)."""
    best = None
    p7 = 1
    while p7 < 2 * max(n, 1):
        p5 = p7
        while p5 < 2 * max(n, 1):
            p3 = p5
            while p3 < 2 * max(n, 1):
                v = p3
                while v < n:
                    v *= 2
                best = v if best is None else min(best, v)
                p3 *= 3
            p5 *= 5
        p7 *= 7
    return best
