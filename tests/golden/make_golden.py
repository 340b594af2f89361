"""Generates tests/golden/*.npz.

The reference holds NO golden vectors (its tests recompute expectations from
time(0)-seeded inputs) and cannot be built here, so these fixtures pin the
oracle against an INDEPENDENT implementation instead: scipy.sparse products of
a Poisson matrix assembled a different way (Kronecker sums + boundary mask),
numpy cumsum / stable argsort, and math.fsum.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def poisson_scipy(n):
    """Same matrix as examples/benchmark.cpp:364-415, assembled independently."""
    h2i = float((n - 1) * (n - 1))
    idx = np.arange(n ** 3).reshape(n, n, n)          # [k][j][i]
    k, j, i = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    interior = (i > 0) & (i < n - 1) & (j > 0) & (j < n - 1) & (k > 0) & (k < n - 1)
    rows, cols, vals = [], [], []
    b = idx[~interior]
    rows.append(b); cols.append(b); vals.append(np.ones(b.size))
    c = idx[interior]
    for off, v in ((-n * n, -h2i), (-n, -h2i), (-1, -h2i), (0, 6 * h2i), (1, -h2i), (n, -h2i), (n * n, -h2i)):
        rows.append(c); cols.append(c + off); vals.append(np.full(c.size, v))
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(n ** 3, n ** 3)).tocsr()
    A.sort_indices()
    return A


def main():
    rng = np.random.default_rng(20240917)
    out = {}
    for n in (5, 12):
        A = poisson_scipy(n)
        x = rng.random(n ** 3)
        out["poisson%d_ptr" % n] = A.indptr.astype(np.int32)
        out["poisson%d_col" % n] = A.indices.astype(np.int32)
        out["poisson%d_val" % n] = A.data
        out["poisson%d_x" % n] = x
        out["poisson%d_y" % n] = A @ x
    # random rectangular matrix, tests/spmv.cpp:61-87 shape (n x 2n, <= 15 per row)
    n, m = 257, 514
    dens = sp.random(n, m, density=8.0 / m, format="csr", random_state=7, dtype=np.float64)
    dens.sort_indices()
    x = rng.random(m)
    out.update(rect_ptr=dens.indptr.astype(np.int32), rect_col=dens.indices.astype(np.int32),
               rect_val=dens.data, rect_x=x, rect_y=dens @ x)
    # reductions / scan / sort
    v = (rng.random(4099) - 0.5) * 1e8                 # tests/vector_arithmetics.cpp:72-86 value range
    out.update(sum_x=v, sum_exact=np.float64(math.fsum(v.tolist())),
               min_exact=v.min(), max_exact=v.max())
    k = rng.integers(0, 101, size=5000).astype(np.int32)      # tests/sort.cpp:22-45: keys U[0,100]
    f = rng.random(5000).astype(np.float32)
    p = np.argsort(k, kind="stable")
    out.update(sort_keys=k, sort_vals=f, sort_keys_sorted=k[p], sort_vals_sorted=f[p])
    u = rng.integers(0, 2 ** 32, size=5000, dtype=np.uint64).astype(np.uint32)
    out.update(scan_in=u, scan_inclusive=np.cumsum(u, dtype=np.uint32))
    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    print("wrote", os.path.join(HERE, "golden.npz"))


if __name__ == "__main__":
    main()
