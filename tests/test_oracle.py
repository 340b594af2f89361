"""CPU tests: the oracle against the committed golden fixtures and against its
own alternative formulations (HELL vs CSR, split vs whole)."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden.npz"))


@pytest.mark.parametrize("n", [5, 12])
def test_poisson_generator_matches_independent_assembly(oracle, n):
    ptr, col, val = oracle.poisson3d(n)
    assert np.array_equal(ptr, G["poisson%d_ptr" % n])
    assert np.array_equal(col, G["poisson%d_col" % n])
    assert np.array_equal(val, G["poisson%d_val" % n])
    assert oracle.poisson3d_nnz(n) == len(col)
    p64, c64, v64 = oracle.poisson3d(n, index_dtype=np.int64)
    assert np.array_equal(p64, ptr) and np.array_equal(c64, col) and np.array_equal(v64, val)


def test_poisson_nnz_formula(oracle):
    # 512^3 -> 930 123 728; benchmark.cpp 128^3 -> 14 099 408
    assert oracle.poisson3d_nnz(512) == 930123728
    assert oracle.poisson3d_nnz(128) == 14099408


@pytest.mark.parametrize("name", ["poisson5", "poisson12", "rect"])
def test_spmv_against_scipy_golden(oracle, name):
    ptr, col, val, x, want = (G[name + s] for s in ("_ptr", "_col", "_val", "_x", "_y"))
    got = oracle.spmv_csr(ptr, col, val, x)
    bound = oracle.spmv_abs_bound(ptr, col, val, x)
    assert np.all(np.abs(got - want) <= 1e-12 * np.maximum(bound, 1e-300))


REF = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_fixtures.npz"))
REF_CASES = ("square", "nonsquare", "types", "emptyrows")


@pytest.mark.parametrize("case", REF_CASES)
@pytest.mark.parametrize("index_dtype", [np.int32, np.int64])
def test_oracle_against_reference_run_fixtures(oracle, case, index_dtype):
    """tests/golden/ref_fixtures.npz was written by oracle/ref_fixture_driver.cpp, which RUNS the reference's generators
    (/root/reference/tests/random_matrix.hpp, random_vector.hpp, fixed srand) and the host loop the reference's test asserts
    against (tests/spmv.cpp:28-32): the restatement must reproduce that loop's y bit for bit -- plain, '+= 42 *' on top of
    y0 (spmv.cpp:44-52), through hybrid ELL (hybrid_ell.inl:254-267: ELL entries then the CSR tail, i.e. the same order); the
    2- and 3-device split (spmat.hpp:120-185: local part, then remote part) within the reference's own tolerance.  A drift of vex_oracle.c from the reference's loop fails here."""
    row, col, val, x, y0, y, y42 = (REF[case + "_" + k] for k in ("row", "col", "val", "x", "y0", "y", "y42"))
    n, m = int(REF[case + "_shape"][0]), int(REF[case + "_shape"][1])
    ptr, c = row.astype(index_dtype), col.astype(index_dtype)
    assert np.array_equal(oracle.spmv_csr(ptr, c, val, x), y)
    got = y0.copy(); oracle.spmv_csr(ptr, c, val, x, got, alpha=42.0, append=True)
    assert np.array_equal(got, y42)
    if index_dtype is np.int32:
        assert np.array_equal(oracle.spmv_csr(ptr, c, val, x, omp=True), y)
        assert np.array_equal(oracle.spmv_hell(oracle.hell_build(ptr, c, val), x), y)
        bound = oracle.spmv_abs_bound(ptr, c, val, x)
        for ndev in (2, 3):      # local part first, then the remote part: another summation order -- the reference's own 1e-8 %
            got = oracle.spmv_split(oracle.split_rows(ptr, c, val, m, ndev), x)
            assert np.all(np.abs(got - y) <= 1e-10 * np.maximum(bound, 1e-300)), ndev
        if n == m:
            yx = x.copy(); oracle.spmv_csr(ptr, c, val, x, yx, alpha=1.0, append=True)
            assert np.array_equal(yx, REF[case + "_yx"])               # Y = X + A * X (spmv.cpp:54-58)


@pytest.mark.parametrize("case", REF_CASES)
def test_reference_generator_has_the_layout_the_restated_generator_promises(oracle, case):
    """What oracle.random_matrix states about tests/random_matrix.hpp (width uniform in [0, nnz_per_row - 1], distinct ascending
    columns in [0, m), values in [0, 1), trailing rows empty in the empty_rows case), checked on the output of the
    REFERENCE's generator itself, next to the restated generator's output for the same shape."""
    row, col, val = REF[case + "_row"], REF[case + "_col"], REF[case + "_val"]
    n, m, filled = (int(v) for v in REF[case + "_shape"][:3])
    mine = oracle.random_matrix(7, n, m, 16, empty_tail=n - filled)
    for ptr, cc, vv in ((row, col, val), mine):
        w = np.diff(ptr)
        assert len(ptr) == n + 1 and ptr[0] == 0 and ptr[-1] == len(cc) == len(vv)
        assert w.min() >= 0 and w.max() <= 15 and np.all(w[filled:] == 0) and w[:filled].max() >= 12
        assert cc.min() >= 0 and cc.max() < m and vv.min() >= 0.0 and vv.max() < 1.0
        for i in range(filled):
            assert np.all(np.diff(cc[ptr[i]:ptr[i + 1]]) > 0)
    # both generators draw the widths uniformly: about 7.5 entries per filled row
    assert abs(len(col) / filled - 7.5) < 0.6 and abs(len(mine[1]) / filled - 7.5) < 0.6


def test_spmv_alpha_append_semantics(oracle):
    # spmat.hpp:120-121 / tests/spmv.cpp:34-58
    ptr, col, val = oracle.random_matrix(1, 1024, 1024, 16)
    x = oracle.random_f64(2, 1024)
    y0 = oracle.random_f64(3, 1024)
    ax = oracle.spmv_csr(ptr, col, val, x)
    y = y0.copy(); oracle.spmv_csr(ptr, col, val, x, y, alpha=42.0, append=True)
    assert np.allclose(y, y0 + 42 * ax, rtol=1e-13, atol=0)
    y = ax.copy(); oracle.spmv_csr(ptr, col, val, x, y, alpha=-1.0, append=True)
    assert np.all(np.abs(y) <= 1e-8)
    assert np.array_equal(oracle.spmv_csr(ptr, col, val, x, omp=True), ax)


def test_random_matrix_shape(oracle):
    # tests/random_matrix.hpp: width in [0, nnz_per_row-1], distinct sorted columns
    ptr, col, val = oracle.random_matrix(5, 1024, 2048, 16, empty_tail=768)
    w = np.diff(ptr)
    assert w.max() <= 15 and np.all(w[-768:] == 0) and w[:256].max() > 0
    for i in range(256):
        c = col[ptr[i]:ptr[i + 1]]
        assert np.all(np.diff(c) > 0) and (len(c) == 0 or (c.min() >= 0 and c.max() < 2048))
    assert val.min() >= 0 and val.max() < 1


@pytest.mark.parametrize("seed,n,m,tail", [(11, 1024, 1024, 0), (12, 1024, 2048, 0), (13, 1024, 1024, 768)])
def test_hell_equals_csr(oracle, seed, n, m, tail):
    ptr, col, val = oracle.random_matrix(seed, n, m, 16, empty_tail=tail)
    x = oracle.random_f64(seed + 100, m)
    h = oracle.hell_build(ptr, col, val)
    assert h["pitch"] % 16 == 0 and h["pitch"] >= n
    wide = np.sum(np.diff(ptr) > h["width"])
    assert 3 * wide < n                                   # hybrid_ell.inl:103-110
    assert h["tail"] == np.sum(np.maximum(np.diff(ptr) - h["width"], 0))
    # ELL entries first, CSR tail after => same order as plain CSR => same bits
    assert np.array_equal(oracle.spmv_hell(h, x), oracle.spmv_csr(ptr, col, val, x))


def test_hell_width_poisson(oracle):
    ptr, col, val = oracle.poisson3d(12)
    assert oracle.hell_width(ptr) == 7


@pytest.mark.parametrize("ndev", [1, 2, 3, 8])
def test_partition(oracle, ndev):
    # vector.hpp:157-162: boundaries aligned up to 16, clamped to n
    for n in (0, 5, 1024, 1000, 134217728):
        p = oracle.partition(n, ndev)
        assert p[0] == 0 and p[-1] == n and len(p) == ndev + 1
        assert all(a <= b for a, b in zip(p, p[1:]))
        assert all(b % 16 == 0 or b == n for b in p[1:-1])
    assert oracle.partition(134217728, 8) == [16777216 * d for d in range(9)]


@pytest.mark.parametrize("ndev", [2, 3, 4])
def test_split_apply_equals_whole(oracle, ndev):
    # spmat.hpp:120-185 on the host: local + remote parts with ghost renumbering
    for ptr, col, val, m in [oracle.random_matrix(21, 1024, 1024, 16) + (1024,),
                             oracle.random_matrix(22, 1024, 2048, 16) + (2048,),
                             oracle.poisson3d(10) + (1000,)]:
        x = oracle.random_f64(23, m)
        S = oracle.split_rows(ptr, col, val, m, ndev)
        got = oracle.spmv_split(S, x, alpha=1.5)
        want = oracle.spmv_csr(ptr, col, val, x, alpha=1.5)
        bound = 1.5 * oracle.spmv_abs_bound(ptr, col, val, x)
        assert np.all(np.abs(got - want) <= 1e-13 * np.maximum(bound, 1e-300))
        for d, D in enumerate(S["devs"]):                 # ghosts are really non-local
            c0, c1 = D["cols"]
            assert np.all((D["ghosts"] < c0) | (D["ghosts"] >= c1))
            assert np.array_equal(S["cols_to_send"][D["cols_to_recv"]], D["ghosts"])


def test_reductions(oracle):
    v = G["sum_x"]
    assert abs(oracle.sum_kahan(v) - float(G["sum_exact"])) <= 1e-10 * np.abs(v).sum() * 1e-6
    assert v.min() == G["min_exact"] and v.max() == G["max_exact"]


def test_scan_sort_golden(oracle):
    assert np.array_equal(oracle.inclusive_scan(G["scan_in"]), G["scan_inclusive"])
    ex = oracle.exclusive_scan(G["scan_in"], 7)
    assert ex[0] == 7 and np.array_equal(ex[1:], (G["scan_inclusive"][:-1] + np.uint32(7)))
    k, v = oracle.stable_sort_by_key(G["sort_keys"], G["sort_vals"])
    assert np.array_equal(k, G["sort_keys_sorted"]) and np.array_equal(v, G["sort_vals_sorted"])


def test_elementwise(oracle):
    b, c, d = (oracle.random_f64(s, 1000) for s in (1, 2, 3))
    assert np.allclose(oracle.ew_mul_add_sin(b, c, d), b * c + np.sin(d), rtol=1e-15)


def test_random_generators_known_answer_vectors(oracle):
    """Philox / Threefry (vexcl/random/*.hpp) against the known-answer vectors published with the algorithms
    (Random123 kat_vectors: Salmon, Moraes, Dror, Shaw, SC'11) -- this part of the oracle is pinned."""
    import numpy as np
    u = lambda *w: np.array(w, dtype=np.uint32)
    pi4, pik2 = u(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), u(0xa4093822, 0x299f31d0)
    assert oracle.philox(u(0, 0), u(0)).tolist() == [0xff1dae59, 0x6cd10df2]
    assert oracle.philox(u(0, 0, 0, 0), u(0, 0)).tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox(u(*[0xffffffff] * 4), u(0xffffffff, 0xffffffff)).tolist() == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox(pi4, pik2).tolist() == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # 64-bit words (Random123 kat_vectors: philox2x64, philox4x64, threefry2x64)
    q = lambda *w: np.array(w, dtype=np.uint64)
    F = 0xffffffffffffffff
    assert oracle.philox(q(0, 0), q(0)).tolist() == [0xca00a0459843d731, 0x66c24222c9a845b5]
    assert oracle.philox(q(0, 0, 0, 0), q(0, 0)).tolist() == [0x16554d9eca36314c, 0xdb20fe9d672d0fdc, 0xd7e772cee186176b, 0x7e68b68aec7ba23b]
    assert oracle.philox(q(F, F, F, F), q(F, F)).tolist() == [0x87b092c3013fe90b, 0x438c3c67be8d0224, 0x9cc7d7c69cd777b6, 0xa09caebf594f0ba0]
    assert oracle.philox(q(0x243f6a8885a308d3, 0x13198a2e03707344, 0xa4093822299f31d0, 0x082efa98ec4e6c89),
                         q(0x452821e638d01377, 0xbe5466cf34e90c6c)).tolist() == [
        0xa528f45403e61d95, 0x38c72dbd566e9788, 0xa5a1610e72fd18b5, 0x57bd43b5e52b7fe6]
    assert oracle.threefry(q(0, 0), q(0, 0)).tolist() == [0xc2b6e3a8c2c69865, 0x6f81ed42f350084d]
    assert oracle.threefry(u(0, 0), u(0, 0)).tolist() == [0x6b200159, 0x99ba4efe]
    assert oracle.threefry(u(0xffffffff, 0xffffffff), u(0xffffffff, 0xffffffff)).tolist() == [0x1cb996fc, 0xbb002be7]
    assert oracle.threefry(u(0x243f6a88, 0x85a308d3), u(0x13198a2e, 0x03707344)).tolist() == [0xc4923a9c, 0x483df7a0]
    # the streams of vex::Random: shapes, ranges, vectorised over the index
    idx = np.arange(10000)
    d = oracle.random_uniform(idx, 42)
    assert d.dtype == np.float64 and d.min() >= 0 and d.max() <= 1 and abs(d.mean() - 0.5) < 0.02
    assert oracle.random_uniform(idx, 42, np.uint32).dtype == np.uint32
    z = oracle.random_normal(idx, 7)
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1) < 0.05
    assert not np.array_equal(oracle.random_uniform(idx, 1), oracle.random_uniform(idx, 2))


def test_fft_oracle_matches_the_definition(oracle):
    """oracle.fft_nd (numpy's pocketfft per axis) against the transform's definition in extended precision,
    forward and inverse, 1-D and along each axis of an n-D array with batch dimensions."""
    rng = np.random.default_rng(11)
    for n in [1, 2, 3, 4, 5, 7, 8, 12, 17, 30, 64, 101]:
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        assert np.abs(oracle.fft_nd(x, [n], [oracle.FFT_FORWARD]) - oracle.dft_definition(x)).max() <= 1e-13 * n
        assert np.abs(oracle.fft_nd(x, [n], [oracle.FFT_INVERSE]) - oracle.dft_definition(x, True)).max() <= 1e-13
    sizes, dirs = [3, 5, 4], [oracle.FFT_NONE, oracle.FFT_FORWARD, oracle.FFT_INVERSE]
    x = rng.standard_normal(60) + 1j * rng.standard_normal(60)
    a = x.reshape(sizes)
    want = np.empty_like(a)
    for b in range(3):
        t = np.stack([oracle.dft_definition(a[b, :, c]) for c in range(4)], axis=1)             # forward along dim 1
        want[b] = np.stack([oracle.dft_definition(t[r, :], True) for r in range(5)], axis=0)   # inverse along dim 2
    assert np.abs(oracle.fft_nd(x, sizes, dirs) - want.reshape(-1)).max() <= 1e-12
    assert [oracle.fft_best_size(n) for n in (1, 2, 11, 17, 1025, 4097)] == [1, 2, 12, 18, 1029, 4116]


def test_variable_coefficient_generator(oracle):
    """The general-matrix generator of bench.py's variable-coefficient row: the Poisson pattern, symmetric, interior rows
    summing to zero, (almost) every entry its own value; deterministic in the seed."""
    import scipy.sparse as sp
    n = 12
    ptr, col, val = oracle.diffusion3d(n)
    pp, cc, vv = oracle.poisson3d(n)
    assert np.array_equal(ptr, pp) and np.array_equal(col, cc)
    A = sp.csr_matrix((val, col, ptr), shape=(n ** 3, n ** 3))
    interior = np.diff(ptr) == 7
    B = A[interior][:, interior]
    assert abs(B - B.T).max() == 0.0                                  # couplings between interior points are symmetric
    assert np.abs(A @ np.ones(n ** 3))[interior].max() <= 1e-9 * np.abs(val).max()
    assert np.all(val[np.isin(np.arange(len(val)), ptr[:-1][~interior])] == 1.0)      # boundary rows: identity
    assert len(np.unique(val[val != 1.0])) > 0.45 * np.count_nonzero(val != 1.0)      # each coupling appears in two rows
    p2, c2, v2 = oracle.diffusion3d(n, seed=8)
    assert not np.array_equal(val, v2) and np.array_equal(oracle.diffusion3d(n)[2], val)


def test_sell_pair_slots_keep_row_order(oracle):
    """Row-pair placement of the SELL storages (restated for the layout tests): whatever the two rows look like, each
    row's entries keep their CSR order, equal diagonals share a column, and a pair that does not fit stays packed."""
    rng = np.random.default_rng(5)
    for trial in range(200):
        n, m, w = 2, int(rng.integers(3, 40)), int(rng.integers(1, 9))
        rows = [np.sort(rng.choice(m, size=int(rng.integers(0, min(m, 10))), replace=False)) if rng.random() < 0.7
                else rng.integers(0, m, size=int(rng.integers(0, 10))) for _ in range(n)]
        ptr = np.array([0, len(rows[0]), len(rows[0]) + len(rows[1])], dtype=np.int32)
        col = np.concatenate(rows).astype(np.int32) if ptr[-1] else np.zeros(0, dtype=np.int32)
        slots = oracle.sell_pair_slots(ptr, col, 0, w, m - 1)
        for q in (0, 1):
            assert len(slots[q]) == w
            real = [e for e in slots[q] if not isinstance(e, str)]
            assert real == list(range(int(ptr[q]), min(int(ptr[q + 1]), int(ptr[q]) + w)))
        for e0, e1 in zip(*slots):
            if isinstance(e0, str) or isinstance(e1, str):
                continue
            aligned = all(isinstance(a, str) or isinstance(b, str) or int(col[a]) - 0 == int(col[b]) - 1 for a, b in zip(*slots))
            packed = [e for e in slots[0] if not isinstance(e, str)] == slots[0][:len([e for e in slots[0] if not isinstance(e, str)])]
            assert aligned or packed


def test_cpu_baseline_runs(oracle):
    r = oracle.cpu_baseline_poisson(24, 0.05, 1)
    ptr, col, val = oracle.poisson3d(24)
    y = oracle.spmv_csr(ptr, col, val, np.full(24 ** 3, 0.01))
    assert r["threads"] >= 1 and r["products"] >= 1 and r["seconds_per_product"] > 0
    assert abs(r["sum_y"] - y.sum()) <= 1e-9 * np.abs(y).sum()
