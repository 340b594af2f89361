"""GPU parity tests for the SpMV path, through the C ABI (libvexhip.so),
against the CPU oracle on the cases the reference tests (tests/spmv.cpp:10-146,
tests/sparse_matrices.cpp:66-151) and on Poisson matrices
(examples/benchmark.cpp:364-415).

Tolerance: |y - y_ref| <= 1e-10 * sum_j |a_ij x_j| -- the reference's own
BOOST_CHECK_CLOSE(.., 1e-8 percent).  The kernels fold each row in CSR order
with unfused multiply-add, so most checks are in fact bit-exact and say so.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-10
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden.npz"))


@pytest.fixture(scope="module")
def T():
    import torch
    from vexcl_amd import ops
    assert torch.cuda.is_available()

    class NS:
        pass
    ns = NS()
    ns.torch, ns.ops, ns.dev = torch, ops, torch.device("cuda:0")
    from vexcl_amd import lib
    ns.L = lib()
    ns.up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ns.dev)
    return ns


def _check(got, want, bound, exact=False):
    if exact:
        assert np.array_equal(got, want)
    err = np.abs(got - want)
    assert np.all(err <= TOL * np.maximum(bound, 1e-300)), float(np.max(err / np.maximum(bound, 1e-300)))


CASES = {
    "square": lambda o: o.random_matrix(101, 1024, 1024, 16) + (1024,),          # spmv.cpp:10-59
    "nonsquare": lambda o: o.random_matrix(102, 1024, 2048, 16) + (2048,),       # spmv.cpp:61-87
    "empty_rows": lambda o: o.random_matrix(103, 1024, 1024, 16, 768) + (1024,),  # spmv.cpp:116-146
    "odd_size": lambda o: o.random_matrix(104, 1000 + 37, 911, 16) + (911,),
    "wide_rows": lambda o: o.random_matrix(105, 700, 5000, 900) + (5000,),        # rows span several LDS tiles
    "tiny": lambda o: o.random_matrix(106, 3, 5, 4) + (5,),
    "poisson32": lambda o: o.poisson3d(32) + (32 ** 3,),                          # spmv.cpp:148-231 grid size
}


@pytest.mark.parametrize("fmt", ["csr", "hell", "sell"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_spmv_set_append_scale(T, oracle, case, fmt):
    ptr, col, val, m = CASES[case](oracle)
    n = len(ptr) - 1
    x = oracle.random_f64(7, m)
    y0 = oracle.random_f64(8, n)
    A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), n_cols=m, fmt=fmt)
    assert (A.rows(), A.cols(), A.nonzeros()) == (n, m, len(col))
    dx = T.up(x)
    bound = oracle.spmv_abs_bound(ptr, col, val, x)

    # Y = A * X
    y = T.up(y0.copy()); A.apply(dx, y, 1.0, False)
    _check(y.cpu().numpy(), oracle.spmv_csr(ptr, col, val, x), bound, exact=True)
    # Y += 42 * (A * X)
    y = T.up(y0.copy()); A.apply(dx, y, 42.0, True)
    want = y0.copy(); oracle.spmv_csr(ptr, col, val, x, want, 42.0, True)
    _check(y.cpu().numpy(), want, 42 * bound + np.abs(y0), exact=True)
    # Y = A*X; Y -= A * X  -> 0   (spmv.cpp:34-40: |y| < 1e-8)
    y = A @ dx; A.apply(dx, y, -1.0, True)
    assert float(y.abs().max()) < 1e-8


REF = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_fixtures.npz"))


@pytest.mark.parametrize("fmt", ["csr", "hell", "sell"])
@pytest.mark.parametrize("case", ["square", "nonsquare", "types", "emptyrows"])
def test_spmv_against_reference_run_fixtures(T, case, fmt):
    """The HIP path against y computed by REFERENCE-EXECUTED code: matrices and vectors from the reference's own generators
    (tests/random_matrix.hpp, random_vector.hpp) and the host loop its test asserts against (tests/spmv.cpp:28-32), written
    by oracle/ref_fixture_driver.cpp into tests/golden/ref_fixtures.npz.  The kernels fold a row in storage order with
    unfused multiply and add, so the check is bit for bit (the reference's own assertion is 1e-8 %)."""
    row, col, val, x, y0, y, y42 = (REF[case + "_" + k] for k in ("row", "col", "val", "x", "y0", "y", "y42"))
    m = int(REF[case + "_shape"][1])
    A = T.ops.SpMat(T.up(row.astype(np.int32)), T.up(col.astype(np.int32)), T.up(val), n_cols=m, fmt=fmt)
    assert np.array_equal((A @ T.up(x)).cpu().numpy(), y)
    got = T.up(y0.copy()); A.apply(T.up(x), got, 42.0, True)
    assert np.array_equal(got.cpu().numpy(), y42)


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_csr_kernel_variants_agree(T, oracle, built_lib, variant):
    ptr, col, val = oracle.poisson3d(40)
    x = oracle.random_f64(5, 40 ** 3)
    want = oracle.spmv_csr(ptr, col, val, x)
    built_lib.spmv_csr_set_variant(variant)
    try:
        y = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), fmt="csr") @ T.up(x)
    finally:
        built_lib.spmv_csr_set_variant(-1)
    assert np.array_equal(y.cpu().numpy(), want)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
def test_hell_kernel_variants_agree(T, oracle, built_lib, variant):
    ptr, col, val = oracle.poisson3d(37)          # odd size: ragged last lanes
    x = oracle.random_f64(5, 37 ** 3)
    want = oracle.spmv_csr(ptr, col, val, x)
    A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), fmt="hell")
    assert A.hell.width == 7 and A.hell.tail_nnz == 0
    built_lib.spmv_hell_set_variant(variant)
    try:
        y = A @ T.up(x)
    finally:
        built_lib.spmv_hell_set_variant(-1)
    assert np.array_equal(y.cpu().numpy(), want)


def test_sell_layout_and_tail(T, oracle):
    # slice-major storage, one contiguous region per slice; same width rule / tail as HELL
    ptr, col, val = oracle.random_matrix(33, 1500, 1500, 16)
    h = oracle.hell_build(ptr, col, val)
    S = T.ops.SlicedELL(T.up(ptr), T.up(col), T.up(val))
    assert (S.width, S.tail_nnz) == (h["width"], h["tail"])
    w, n = S.width, 1500
    raw = S.sell.cpu().numpy().reshape(-1, w * 512 * 12)               # one region per slice: columns, then values
    sc = raw[:, :w * 512 * 4].copy().view(np.int32).reshape(-1, w, 512)
    sv = raw[:, w * 512 * 4:].copy().view(np.float64).reshape(-1, w, 512)
    # rows 2k and 2k+1 are placed together: entries merged by diagonal, each row in its CSR order, padding -1 where a
    # 16-byte load of the partner stays inside x and -2 where it would not (oracle.sell_pair_slots restates the rule)
    max_col = int(max(col[ptr[r]:ptr[r] + w].max() for r in range(n) if ptr[r + 1] > ptr[r]))
    for i in (0, 1, 510, 511, 512, 1022, 1023, 1498, 1499):
        slots = oracle.sell_pair_slots(ptr, col, i - (i % 2), w, max_col)[i % 2]
        want_c = [int(col[e]) if not isinstance(e, str) else (-1 if e == "safe" else -2) for e in slots]
        want_v = [float(val[e]) if not isinstance(e, str) else 0.0 for e in slots]
        assert sc[i // 512, :, i % 512].tolist() == want_c and sv[i // 512, :, i % 512].tolist() == want_v
        assert [c for c in want_c if c >= 0] == col[ptr[i]:min(ptr[i + 1], ptr[i] + w)].tolist()      # the row's ELL entries, in order
    assert np.all(sc[2, :, 1500 - 1024:] < 0)            # padding rows of the last slice
    # a 16-byte load that would leave x is ruled out by the padding code -2: the last row of an odd-sized matrix has no
    # partner, and its entry sits in the last column
    tp, tc, tv = np.array([0, 1, 3, 4], dtype=np.int32), np.array([0, 0, 2, 2], dtype=np.int32), np.array([1.0, 2.0, 3.0, 4.0])
    S3 = T.ops.SlicedELL(T.up(tp), T.up(tc), T.up(tv), codes=False)
    assert S3.width == 2
    raw3 = S3.sell.cpu().numpy()
    c3 = raw3[:2 * 512 * 4].copy().view(np.int32).reshape(2, 512)
    assert c3[:, :4].tolist() == [[0, 0, 2, -2], [-1, 2, -1, -1]]
    y3 = T.torch.empty(3, dtype=T.torch.float64, device=T.dev)
    x3 = T.up(np.array([1.0, np.inf, 0.5]))              # x[1] is never referenced: it must not reach any sum (inf * 0 = NaN)
    S3.mul(x3, y3)
    assert y3.cpu().numpy().tolist() == [1.0, 2.0 * 1.0 + 3.0 * 0.5, 4.0 * 0.5]
    assert np.array_equal(S.csr_ptr.cpu().numpy(), h["csr_ptr"]) and np.array_equal(S.csr_val.cpu().numpy(), h["csr_val"])


def test_hell_conversion_matches_oracle_layout(T, oracle):
    # sparse/ell.hpp:400-508 device conversion == hybrid_ell.inl:138-198 host fill
    ptr, col, val = oracle.random_matrix(31, 1024, 1024, 16)
    h = oracle.hell_build(ptr, col, val)
    A = T.ops.HybridELL(T.up(ptr), T.up(col), T.up(val))
    assert (A.width, A.pitch, A.tail_nnz) == (h["width"], h["pitch"], h["tail"])
    assert np.array_equal(A.ell_col.cpu().numpy(), h["ell_col"])
    assert np.array_equal(A.ell_val.cpu().numpy(), h["ell_val"])
    assert np.array_equal(A.csr_ptr.cpu().numpy(), h["csr_ptr"])
    assert np.array_equal(A.csr_col.cpu().numpy(), h["csr_col"])
    assert np.array_equal(A.csr_val.cpu().numpy(), h["csr_val"])


def test_index_and_value_types(T, oracle):
    # spmv.cpp:89-114 non-default index types; float matrices
    ptr, col, val = oracle.random_matrix(41, 1024, 1024, 16)
    x = oracle.random_f64(42, 1024)
    want = oracle.spmv_csr(ptr, col, val, x)
    A = T.ops.SpMat(T.up(ptr.astype(np.int64)), T.up(col.astype(np.int64)), T.up(val))
    assert A.fmt == "csr"
    assert np.array_equal((A @ T.up(x)).cpu().numpy(), want)
    v32, x32 = val.astype(np.float32), x.astype(np.float32)
    want32 = oracle.spmv_csr(ptr, col, v32, x32)
    for fmt in ("csr", "hell", "sell"):
        y = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), fmt=fmt) @ T.up(x32)
        assert np.array_equal(y.cpu().numpy(), want32)


def test_unaligned_views_take_the_scalar_path(T, oracle):
    ptr, col, val = oracle.random_matrix(51, 512, 512, 16)
    x = oracle.random_f64(52, 512)
    pad_c = T.up(np.concatenate([[0], col]).astype(np.int32))[1:]
    pad_v = T.up(np.concatenate([[0.0], val]))[1:]
    y = T.torch.empty(512, dtype=T.torch.float64, device=T.dev)
    T.ops.spmv_csr(T.up(ptr), pad_c, pad_v, T.up(x), y)
    assert np.array_equal(y.cpu().numpy(), oracle.spmv_csr(ptr, col, val, x))


@pytest.mark.parametrize("name", ["poisson5", "poisson12", "rect"])
def test_golden_fixtures(T, oracle, name):
    ptr, col, val, x, want = (G[name + s] for s in ("_ptr", "_col", "_val", "_x", "_y"))
    bound = oracle.spmv_abs_bound(ptr, col, val, x)
    for fmt in ("csr", "hell", "sell"):
        y = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), n_cols=len(x), fmt=fmt) @ T.up(x)
        _check(y.cpu().numpy(), want, bound)


def test_device_poisson_generator_is_the_reference_matrix(T, oracle):
    for n in (2, 3, 9, 32):
        ptr, col, val = oracle.poisson3d(n)
        dp, dc, dv = T.ops.poisson3d(n, T.dev)
        assert np.array_equal(dp.cpu().numpy(), ptr)
        assert np.array_equal(dc.cpu().numpy(), col)
        assert np.array_equal(dv.cpu().numpy(), val)
    # a strip: global columns, local ptr
    ptr, col, val = oracle.poisson3d(16)
    r0, r1 = 1024, 3072
    dp, dc, dv = T.ops.poisson3d(16, T.dev, rows=(r0, r1))
    assert np.array_equal(dp.cpu().numpy(), ptr[r0:r1 + 1] - ptr[r0])
    assert np.array_equal(dc.cpu().numpy(), col[ptr[r0]:ptr[r1]])


def test_poisson128_benchmark_size(T, oracle):
    # examples/benchmark.cpp:358-360: n = 128, N = 2 097 152, nnz = 14 099 408
    n = 128
    ptr, col, val = oracle.poisson3d(n)
    assert len(col) == 14099408
    x = oracle.random_f64(9, n ** 3)
    want = oracle.spmv_csr(ptr, col, val, x, omp=True)
    dp, dc, dv = T.ops.poisson3d(n, T.dev)
    for fmt in ("csr", "hell", "sell"):
        y = T.ops.SpMat(dp, dc, dv, fmt=fmt) @ T.up(x)
        assert np.array_equal(y.cpu().numpy(), want)


def test_variable_coefficient_128_every_storage(T, oracle):
    """The general banded matrix -- the 7-point pattern with a different coefficient on every face (~4 N distinct values)
    (what bench.py's variable-coefficient row runs at 512^3) -- at 128^3: the device generator equals its host
    restatement bit for bit, the default SpMat picks diagonal codes WITHOUT value codes, and every storage the matrix
    allows (sell8, sell32, csr, the reference's hybrid-ELL layout; `y = ` and `y += alpha *`) equals the oracle's CSR
    loop bit for bit.  The pair kernels must also run the per-entry kernels' arithmetic: A/B switch."""
    n = 128
    N = n ** 3
    ptr, col, val = oracle.diffusion3d(n)
    dp, dc, dv = T.ops.diffusion3d(n, T.dev)
    assert np.array_equal(dp.cpu().numpy(), ptr) and np.array_equal(dc.cpu().numpy(), col) and np.array_equal(dv.cpu().numpy(), val)
    assert len(np.unique(val)) > 3.9 * (n - 2) ** 3            # every face its own coefficient: ~4 distinct values per row, no value coding
    x = oracle.random_f64(11, N) - 0.5
    y0 = oracle.random_f64(12, N)
    want = oracle.spmv_csr(ptr, col, val, x, omp=True)
    want_app = y0.copy(); oracle.spmv_csr(ptr, col, val, x, want_app, -1.75, True)
    A = T.ops.SpMat(dp, dc, dv)
    assert A.storage == "sell8" and A.hell.ndeltas == 7 and A.hell.values is None
    for fmt in ("sell", "sell32", "csr", "hell"):
        for variant in ((0, 1) if fmt in ("sell", "sell32") else (0,)):
            T.L.spmv_sell8_set_variant(variant)
            try:
                B = T.ops.SpMat(dp, dc, dv, fmt=fmt)
                y = B @ T.up(x)
                assert np.array_equal(y.cpu().numpy(), want), (fmt, variant)
                ya = T.up(y0.copy()); B.apply(T.up(x), ya, -1.75, True)
                assert np.array_equal(ya.cpu().numpy(), want_app), (fmt, variant)
            finally:
                T.L.spmv_sell8_set_variant(0)
    # symmetric, and interior rows sum to ~0 (diagonal = minus the sum of the six couplings)
    ones = T.torch.ones(N, dtype=T.torch.float64, device=T.dev)
    r = (A @ ones).view(n, n, n)[1:-1, 1:-1, 1:-1]
    assert float(r.abs().max()) <= 1e-9 * float(np.abs(val).max())


def _band(n, offsets, seed, dtype=np.float64, constant=False):
    """banded matrix with the given diagonals (CSR, columns ascending); constant: one value per diagonal"""
    rng = np.random.default_rng(seed)
    rows = np.arange(n, dtype=np.int64)
    cols = np.stack([rows + o for o in offsets], axis=1)
    ok = (cols >= 0) & (cols < n)
    vals = np.stack([(np.full(n, 0.5 + k) if constant else rng.random(n) + 0.25) for k in range(len(offsets))], axis=1)
    ptr = np.zeros(n + 1, dtype=np.int32)
    ptr[1:] = np.cumsum(ok.sum(axis=1))
    return ptr, cols[ok].astype(np.int32), vals[ok].astype(dtype)


@pytest.mark.parametrize("run", [2, 5, 16, 64])
def test_march_product_is_bit_identical(T, oracle, built_lib, run):
    """The march product (round 3: x window of the near diagonals in an LDS ring carried along a run of slices, far
    diagonals gathered) against the pair product it replaces AND the CSR oracle, bit for bit: Poisson 128^3 (value codes:
    near -128..128, far +-16384), the variable-coefficient operator on the same pattern (stored values), a banded matrix
    whose diagonals are all near and whose last slice is ragged, single precision, '=' and '+= alpha', several run
    lengths (the ring wraps after 2 / 4 slices; run 64 > slices per plane)."""
    torch = T.torch
    os.environ["VEXHIP_MARCH_RUN"] = str(run)
    try:
        n = 128
        N = n ** 3
        x = oracle.random_f64(3, N); y0 = oracle.random_f64(4, N)
        for label, (ptr, col, val) in (("poisson", oracle.poisson3d(n)), ("diffusion", oracle.diffusion3d(n, 11))):
            A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val))
            B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), march=False)
            assert A.storage == ("sell8v" if label == "poisson" else "sell8") and A.dictionary_blocks > 0 and B.march is None
            if label == "poisson":
                assert A.march is not None and A.march["run"] == run, A.march
                assert (A.march["lo"], A.march["hi"]) == (-128, 128) and A.march["x_last"] == N - 1 and A.march["far"] == [-16384, 16384]
            else:
                assert A.march is None            # stored values: bound by the value stream, the pair product stays
            want = oracle.spmv_csr(ptr, col, val, x)
            for alpha, append in ((1.0, False), (-0.75, True)):
                ya, yb = T.up(y0.copy()), T.up(y0.copy())
                A.apply(T.up(x), ya, alpha, append); B.apply(T.up(x), yb, alpha, append)
                assert torch.equal(ya, yb), (label, alpha)
                assert np.array_equal(ya.cpu().numpy(), (y0 + alpha * want) if append else alpha * want), (label, alpha)
            # variant 2 = the pair kernels through the march entry points
            T.L.spmv_sell8_set_variant(2)
            try:
                yc = torch.empty(N, dtype=torch.float64, device=T.dev); A.apply(T.up(x), yc)
                assert np.array_equal(yc.cpu().numpy(), want)
            finally:
                T.L.spmv_sell8_set_variant(0)
            if label == "poisson":
                v32, x32 = val.astype(np.float32), x.astype(np.float32)
                F = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32)); Fp = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), march=False)
                assert F.march is not None and Fp.march is None
                yf = torch.empty(N, dtype=torch.float32, device=T.dev); yp = torch.empty_like(yf)
                F.apply(T.up(x32), yf); Fp.apply(T.up(x32), yp)
                assert torch.equal(yf, yp)
        # banded matrices with a ragged last slice and one value per diagonal (value codes): (a) every diagonal near, odd and
        # even offsets, one beyond the slice length; (b) two far diagonals, requested a slice ahead, whose windows leave x at
        # both ends of the matrix; (c) THREE far diagonals: no slot for the third -- the plan declines (round 4: every wave
        # would take the per-entry loop) and the pair product runs
        m = 200 * 512 + 77
        for offs, near, far in (((-700, -513, -2, -1, 0, 1, 3, 512), (-700, 512), []),
                                ((-5000, -700, -2, 0, 1, 512, 7001), (-700, 512), [-5000, 7001]),
                                ((-9000, -5000, -700, -2, 0, 1, 512, 7001), None, None)):
            for constant in (True, False):
                ptr, col, val = _band(m, offs, 5, constant=constant)
                A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val)); B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), march=False)
                assert A.storage == ("sell8v" if constant else "sell8")
                if constant and near is not None:
                    assert A.march is not None and (A.march["lo"], A.march["hi"]) == near and A.march["far"] == far, A.march
                else:
                    assert A.march is None
                xb = oracle.random_f64(8, m)
                ya = torch.empty(m, dtype=torch.float64, device=T.dev); yb = torch.empty_like(ya)
                A.apply(T.up(xb), ya); B.apply(T.up(xb), yb)
                assert torch.equal(ya, yb)
                assert np.array_equal(ya.cpu().numpy(), oracle.spmv_csr(ptr, col, val, xb))
    finally:
        os.environ.pop("VEXHIP_MARCH_RUN", None)


def test_march_with_one_far_diagonal_and_narrow_widths(T, oracle, built_lib):
    """The hot loop of the march product always carries two far slots (a matrix with ONE far diagonal: the second slot
    repeats the first) or none.  (a) the five-point pattern of a 1024 x 1024 grid: the window takes -1024 .. 1 (three slices
    of span), +1024 stays far; (b) ELL widths 1 .. 3 (diagonal only; two near; one near + one far); each against the pair
    product and the CSR oracle bit for bit, '=' and '+= alpha', fp64 and fp32."""
    torch = T.torch
    cases = (((-1024, -1, 0, 1, 1024), 1024 * 1024 + 300, (-1024, 1), [1024]),
             ((0,), 100 * 512 + 9, (0, 0), []),
             ((-1, 2), 64 * 512, (-1, 2), []),
             ((0, 40000), 150 * 512 + 100, (0, 0), [40000]))
    for offs, m, near, far in cases:
        ptr, col, val = _band(m, offs, 9, constant=True)
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val)); B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), march=False)
        assert A.storage == "sell8v" and A.march is not None and B.march is None, (offs, A.march)
        assert (A.march["lo"], A.march["hi"]) == near and A.march["far"] == far, (offs, A.march)
        xb = oracle.random_f64(12, m); y0 = oracle.random_f64(13, m)
        want = oracle.spmv_csr(ptr, col, val, xb)
        for alpha, append in ((1.0, False), (0.5, True)):
            ya, yb = T.up(y0.copy()), T.up(y0.copy())
            A.apply(T.up(xb), ya, alpha, append); B.apply(T.up(xb), yb, alpha, append)
            assert torch.equal(ya, yb), (offs, alpha)
            assert np.array_equal(ya.cpu().numpy(), (y0 + alpha * want) if append else alpha * want), (offs, alpha)
        v32, x32 = val.astype(np.float32), xb.astype(np.float32)
        F = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32)); Fp = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), march=False)
        assert F.march is not None
        yf = torch.empty(m, dtype=torch.float32, device=T.dev); yp = torch.empty_like(yf)
        F.apply(T.up(x32), yf); Fp.apply(T.up(x32), yp)
        assert torch.equal(yf, yp), offs
    # x that holds NaN / Inf where padding entries "cover" it: rows at the ends of the band must not see them (the masks
    # clear the high word only -- a product with +0.0 must still be +0.0)
    offs, m = (-1024, -1, 0, 1, 1024), 200 * 512
    ptr, col, val = _band(m, offs, 9, constant=True)
    A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val)); B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), march=False)
    xb = oracle.random_f64(14, m)
    xb[0] = np.inf; xb[1] = -np.inf; xb[m - 1] = np.nan          # rows 0 / 1 / m - 1 read them as real entries; padding must not
    ya = torch.empty(m, dtype=torch.float64, device=T.dev); yb = torch.empty_like(ya)
    A.apply(T.up(xb), ya); B.apply(T.up(xb), yb)
    a, b = ya.cpu().numpy(), yb.cpu().numpy()
    assert np.array_equal(a, b, equal_nan=True)
    assert np.array_equal(a, oracle.spmv_csr(ptr, col, val, xb), equal_nan=True)
    assert np.isfinite(a[2048:m - 2048]).all()


def _grid7(nx, ny, nz):
    """7-point operator on an nx x ny x nz grid in the benchmark's form (examples/benchmark.cpp:364-415: boundary rows identity,
    interior rows -h, -h, -h, 6h, -h, -h, -h in that order), CSR with int32 indices; three distinct values"""
    N = nx * ny * nz
    idx = np.arange(N, dtype=np.int64)
    i, j, k = idx % nx, (idx // nx) % ny, idx // (nx * ny)
    interior = (i > 0) & (i < nx - 1) & (j > 0) & (j < ny - 1) & (k > 0) & (k < nz - 1)
    cnt = np.where(interior, 7, 1)
    ptr = np.zeros(N + 1, dtype=np.int64); ptr[1:] = np.cumsum(cnt)
    col = np.empty(ptr[-1], dtype=np.int32); val = np.empty(ptr[-1], dtype=np.float64)
    b = ~interior
    col[ptr[:-1][b]] = idx[b]; val[ptr[:-1][b]] = 1.0
    h = float((nx - 1) ** 2)
    base = ptr[:-1][interior]
    for q, (off, v) in enumerate(((-nx * ny, -h), (-nx, -h), (-1, -h), (0, 6 * h), (1, -h), (nx, -h), (nx * ny, -h))):
        col[base + q] = idx[interior] + off; val[base + q] = v
    return ptr.astype(np.int32), col, val


def _grid7_natural(nx, ny, nz, zero_face=False):
    """7-point operator with natural boundaries: every row holds its diagonal (= the number of neighbours it has) and -1 for each
    neighbour inside the grid, columns ascending -- nine kinds of grid lines.  zero_face: the +-1 entries of plane 1 are stored
    as explicit 0.0 (an entry with value zero is not the same as no entry: 0 * Inf = NaN)."""
    N = nx * ny * nz
    idx = np.arange(N, dtype=np.int64)
    i, j, k = idx % nx, (idx // nx) % ny, idx // (nx * ny)
    has = np.stack([k > 0, j > 0, i > 0, np.ones(N, bool), i < nx - 1, j < ny - 1, k < nz - 1], axis=1)
    offs = np.array([-nx * ny, -nx, -1, 0, 1, nx, nx * ny], dtype=np.int64)
    cols = idx[:, None] + offs[None, :]
    vals = np.where(np.arange(7)[None, :] == 3, (has.sum(axis=1) - 1).astype(np.float64)[:, None], -1.0)
    if zero_face:
        vals[(k == 1)[:, None] & ((np.arange(7) == 2) | (np.arange(7) == 4))[None, :]] = 0.0
    ptr = np.zeros(N + 1, dtype=np.int64); ptr[1:] = np.cumsum(has.sum(axis=1))
    return ptr.astype(np.int32), cols[has].astype(np.int32), vals[has]


def test_grid_product_is_bit_identical(T, oracle, built_lib):
    """The grid product (round 4, grid.hip: the plane walk for grid lines of ANY length -- the matrix re-expressed by grid line,
    a class per line) against the pair / streamed products AND the CSR restatement, bit for bit: the benchmark's operator on
    96^3 and 125^3 (odd line length: 16-byte requests at 8-byte addresses; odd lines per plane: the last tile stores one line),
    lines longer than 512 (one segment up to 1024 points, two beyond), 384- and 500-point lines, natural boundaries (nine line classes, explicit zeros),
    full bands whose +-1 diagonal crosses the line ends, a ragged last plane, several walk depths, '=' and '+= alpha',
    Inf / NaN in x under absent entries; and what the plan declines."""
    torch = T.torch
    try:
        def check(ptr, col, val, shape, seed, depth=None, expect=True, classes=None):
            if depth is None:
                os.environ.pop("VEXHIP_PLANE_DEPTH", None)
            else:
                os.environ["VEXHIP_PLANE_DEPTH"] = str(depth)
            m = len(ptr) - 1
            B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), march=False)
            assert B.storage == "sell8v" and B.grid is None and B.plane is None and B.march is None
            if not expect:
                for direct in (True, False):
                    assert T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), direct=direct).grid is None, (shape, direct)
                return
            xb = oracle.random_f64(seed, m); y0 = oracle.random_f64(seed + 1, m)
            want = oracle.spmv_csr(ptr, col, val, xb)
            # direct: the matrix by grid line straight from the CSR arrays (grid_build, the default); not direct: the SELL-512
            # storage first, the grid plan from its slices' codes
            for direct in (True, False):
                A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), direct=direct)
                assert A.storage == "sell8v" and A.grid is not None and A.plane is None and A.direct == direct, (shape, direct, A.storage, A.dictionary_blocks)
                assert (A.grid["nx"], A.grid["lines_per_plane"]) == shape[:2] and A.grid["x_last"] == m - 1, (shape, A.grid)
                assert A.grid["segments"] == (shape[0] + 1023) // 1024 and (depth is None or A.grid["depth"] == min(depth, A.grid["planes"])), (shape, A.grid)
                if classes is not None:
                    assert A.grid["classes"] == classes, (shape, A.grid)
                for alpha, append in ((1.0, False), (-0.75, True)):
                    ya, yb = T.up(y0.copy()), T.up(y0.copy())
                    A.apply(T.up(xb), ya, alpha, append); B.apply(T.up(xb), yb, alpha, append)
                    assert torch.equal(ya, yb), (shape, alpha, direct)
                    assert np.array_equal(ya.cpu().numpy(), (y0 + alpha * want) if append else alpha * want), (shape, alpha, direct)
            return A

        # Small grids: the plan is forced (the library keeps the pair product below 2^23 rows -- x lives in the L2s there -- and
        # declines when more than a quarter of the lines use another class than the most frequent one)
        os.environ["VEXHIP_PLANE_FORCE"] = "1"
        # the benchmark's operator (examples/benchmark.cpp:364-415): two line classes
        for n in (96, 125):
            ptr, col, val = oracle.poisson3d(n)
            check(ptr, col, val, (n, n, n), 31, classes=2)
        for shape, depth in (((70, 33, 20), None), ((70, 33, 20), 3), ((1030, 6, 8), None), ((1030, 6, 8), 5), ((384, 10, 12), None),
                             ((500, 7, 11), 4), ((127, 17, 19), 7), ((514, 5, 13), None),
                             # (round 5: 513 .. 1024 points in ONE segment -- workgroups of 5 .. 8 waves)
                             ((640, 6, 9), None), ((700, 5, 8), 3), ((1000, 4, 7), 2), ((1024, 6, 6), None)):
            ptr, col, val = _grid7(*shape)
            check(ptr, col, val, shape, 33, depth, classes=2)
        # natural boundaries: nine line classes, the other class changes along every walk; explicit zeros in plane 1
        for shape, depth in (((96, 20, 21), None), ((125, 15, 18), 5), ((250, 9, 16), None)):
            ptr, col, val = _grid7_natural(*shape, zero_face=True)
            A = check(ptr, col, val, shape, 35, depth)
            assert A.grid["classes"] >= 9, A.grid
        # full bands: the +-1 diagonal crosses the line ends (lane 0 / the last lane of a line read the neighbouring line's
        # element), a ragged last plane (lines not a multiple of the lines per plane)
        for nx, ny, nz, extra_lines, depth in ((96, 12, 30, 0, None), (125, 9, 33, 4, 6), (300, 8, 14, 3, None)):
            P = nx * ny; m = P * nz + extra_lines * nx
            ptr, col, val = _band(m, (-P, -nx, -1, 0, 1, nx, P), 5, constant=True)
            check(ptr, col, val, (nx, ny, nz), 37, depth)
        os.environ.pop("VEXHIP_PLANE_DEPTH", None)
        # Inf / NaN in x: only the rows that reference them may see them; a stored 0.0 times Inf is NaN as in the CSR loop
        shape = (125, 15, 18)
        ptr, col, val = _grid7_natural(*shape, zero_face=True)
        m = len(ptr) - 1
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val))
        assert A.grid is not None
        xb = oracle.random_f64(39, m)
        xb[0] = np.inf; xb[1] = -np.inf; xb[m - 1] = np.nan; xb[125 * 15 + 7 * 125 + 60] = np.inf; xb[5 * 125 * 15 + 124] = np.nan
        ya = torch.empty(m, dtype=torch.float64, device=T.dev)
        A.apply(T.up(xb), ya)
        want = oracle.spmv_csr(ptr, col, val, xb)
        assert np.isnan(want).sum() >= 8                      # the neighbours of the NaN and of the Inf under a stored zero
        assert np.array_equal(ya.cpu().numpy(), want, equal_nan=True)
        # the Poisson matrix bordered by Inf: the boundary rows are identity rows, their neighbours never look at them... but
        # interior rows next to the boundary do: as the CSR loop
        ptr, col, val = oracle.poisson3d(96)
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val))
        xb = oracle.random_f64(40, 96 ** 3); xb[::97] = np.inf
        ya = torch.empty(96 ** 3, dtype=torch.float64, device=T.dev)
        A.apply(T.up(xb), ya)
        assert A.grid is not None and np.array_equal(ya.cpu().numpy(), oracle.spmv_csr(ptr, col, val, xb), equal_nan=True)
        os.environ.pop("VEXHIP_PLANE_FORCE")
        # the plan as the library chooses it: 208^3 (9.0e6 rows: above the threshold), checked against the pair product and,
        # on the first planes, against the CSR restatement
        n = 208
        ptr, col, val = oracle.poisson3d(n)
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val)); B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), march=False)
        assert A.direct and A.grid is not None and A.grid["nx"] == n and A.grid["classes"] == 2 and B.grid is None, A.grid
        C = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), direct=False)
        assert not C.direct and C.grid is not None and C.grid["classes"] == 2
        xb = oracle.random_f64(47, n ** 3)
        ya = torch.empty(n ** 3, dtype=torch.float64, device=T.dev); yb = torch.empty_like(ya)
        A.apply(T.up(xb), ya); B.apply(T.up(xb), yb)
        assert torch.equal(ya, yb) and torch.equal(C @ T.up(xb), yb)
        # four right-hand sides through the storage by grid line: one product per component
        xs = [T.up(oracle.random_f64(50 + k, n ** 3)) for k in range(3)]
        ys = [torch.empty_like(ya) for _ in range(3)]
        A.apply_multi(xs, ys)
        for k in range(3):
            assert torch.equal(ys[k], B @ xs[k])
        del C
        assert np.array_equal(ya.cpu().numpy(), oracle.spmv_csr(ptr, col, val, xb))
        assert T.ops.SpMat(*[T.up(a) for a in oracle.poisson3d(96)]).grid is None          # small: the pair product stays

        os.environ["VEXHIP_PLANE_FORCE"] = "1"           # the structural reasons to decline hold whatever the size
        # declined: an eighth diagonal; a 2-D operator whose rows are no multiple of 1024 points; rows reversed (storage order is not position order);
        # rows that do not fill whole lines; plane=False / dictionary=False keep the older products
        ptr, col, val = _grid7(96, 20, 21)
        assert T.ops.SpMat(T.up(ptr), T.up(col), T.up(val.astype(np.float32))).grid is not None       # (fp32: test_grid_product_fp32_is_bit_identical)
        assert T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), plane=False).grid is None
        assert T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), dictionary=False).grid is None
        P = 96 * 20; m = P * 21
        b8 = _band(m, (-P, -96, -2, -1, 0, 1, 96, P), 7, constant=True)
        check(*b8, (96, 20, 21), 41, expect=False)
        b5 = _band(m, (-96, -1, 0, 1, 96), 7, constant=True)          # (a 2-D operator is walked only where 512-point virtual lines fit: round 5, test below)
        check(*b5, (96, 20, 21), 41, expect=False)
        ptr, col, val = _band(m, (-P, -96, -1, 0, 1, 96, P), 7, constant=True)
        rcol, rval = col.copy(), val.copy()
        for r in range(3 * P, 3 * P + 200):                      # a few rows with their entries reversed
            rcol[ptr[r]:ptr[r + 1]] = col[ptr[r]:ptr[r + 1]][::-1]; rval[ptr[r]:ptr[r + 1]] = val[ptr[r]:ptr[r + 1]][::-1]
        R = T.ops.SpMat(T.up(ptr), T.up(rcol), T.up(rval))
        assert R.grid is None
        xb = oracle.random_f64(43, m)
        assert np.array_equal((R @ T.up(xb)).cpu().numpy(), oracle.spmv_csr(ptr, rcol, rval, xb))
        ptr, col, val = _band(m + 50, (-P, -96, -1, 0, 1, 96, P), 7, constant=True)
        check(ptr, col, val, (96, 20, 21), 45, expect=False)
        # grid matrices that the one-pass set-up starts on and gives up: more than 254 values (a coefficient per face: the device-wide
        # value table fills up) and more than 128 classes of lines (a value that depends on the line) -- the SELL-512 set-up takes
        # over and the products equal the CSR restatement
        ptr, col, val = oracle.diffusion3d(48, 3)
        D = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val))
        assert D.grid is None and not D.direct and D.storage == "sell8", (D.storage, D.grid)
        xb = oracle.random_f64(49, 48 ** 3)
        assert np.array_equal((D @ T.up(xb)).cpu().numpy(), oracle.spmv_csr(ptr, col, val, xb))
        ptr, col, val = _grid7(96, 20, 21)
        val = val.copy()
        line_of_row = np.arange(96 * 20 * 21) // 96
        centre = (np.arange(96 * 20 * 21) % 96) == 40
        rows_c = np.nonzero(centre & (np.diff(ptr) == 7))[0]
        val[ptr[rows_c] + 3] = 1000.0 + (line_of_row[rows_c] % 200)              # the diagonal of one row per interior line: 200 values, > 128 classes
        C = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val))
        assert C.grid is None and not C.direct and C.storage == "sell8v", (C.storage, C.grid)
        xb = oracle.random_f64(51, len(ptr) - 1)
        assert np.array_equal((C @ T.up(xb)).cpu().numpy(), oracle.spmv_csr(ptr, col, val, xb))
    finally:
        os.environ.pop("VEXHIP_PLANE_DEPTH", None)
        os.environ.pop("VEXHIP_PLANE_FORCE", None)


def test_storage_by_grid_line_holds_the_matrix(T, oracle, built_lib):
    """The storage by grid line read back and decoded on the host: line_class[rows / nx], one table of 7 x pitch value codes per
    class (255 = no entry), the value table and the diagonal table reproduce every row of the CSR matrix -- columns, values and
    their ORDER (position order is storage order) -- for a matrix with natural boundaries (nine line classes, explicit zeros) and
    for the benchmark's operator; the value table holds exactly the distinct values, the diagonal table is sorted."""
    import ctypes
    os.environ["VEXHIP_PLANE_FORCE"] = "1"
    try:
        for (ptr, col, val), shape in ((_grid7_natural(70, 11, 13, zero_face=True), (70, 11, 13)), (oracle.poisson3d(40), (40, 40, 40))):
            nx, ny, nz = shape
            m = len(ptr) - 1
            A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val))
            assert A.direct and A.grid is not None and A.grid["nx"] == nx and A.grid["lines_per_plane"] == ny, A.grid
            g = A.info.grid
            lines, classes, pitch = m // nx, int(g.classes), int(g.pitch)

            def read(ptr_, nbytes, dtype):
                host = np.empty(nbytes, dtype=np.uint8)
                built_lib.memcpy_d2h(0, ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(ptr_), nbytes, None, 1)
                return host.view(dtype)
            line_class = read(g.line_class, 4 * lines, np.int32)
            table = read(g.table, classes * 7 * pitch, np.uint8).reshape(classes, 7, pitch)
            values = read(A.info.values, 8 * 256, np.float64)
            deltas = read(A.info.deltas, 4 * 256, np.int32)
            nd, nv = int(A.info.ndeltas), int(A.info.nvalues)
            assert sorted(set(val.tolist())) == sorted(values[:nv].tolist()) and values[255] == 0.0
            by_pos = [-nx * ny, -nx, -1, 0, 1, nx, nx * ny]
            assert deltas[:nd].tolist() == sorted(deltas[:nd].tolist()) and set(deltas[:nd].tolist()) <= set(by_pos)
            assert set(deltas[:nd].tolist()) == set((col - np.repeat(np.arange(m), np.diff(ptr))).tolist())
            assert line_class.min() >= 0 and line_class.max() < classes and (table[:, :, nx:] == 255).all()
            rng = np.random.default_rng(5)
            rows = np.unique(np.concatenate([rng.integers(0, m, 4000), np.arange(0, 3 * nx), np.arange(m - 3 * nx, m)]))
            for i in rows:
                codes = table[line_class[i // nx], :, i % nx]
                got_cols = [i + by_pos[p] for p in range(7) if codes[p] != 255]
                got_vals = [values[codes[p]] for p in range(7) if codes[p] != 255]
                assert got_cols == col[ptr[i]:ptr[i + 1]].tolist() and got_vals == val[ptr[i]:ptr[i + 1]].tolist(), (shape, i)
    finally:
        os.environ.pop("VEXHIP_PLANE_FORCE", None)


def _stencil27_const(g, gz=None):
    """constant-coefficient 27-point operator on g x g x gz (g^3 by default), identity rows on the boundary, columns ascending"""
    gz = g if gz is None else gz
    N = g * g * gz
    idx = np.arange(N, dtype=np.int64)
    i, j, k = idx % g, (idx // g) % g, idx // (g * g)
    inner = (i > 0) & (i < g - 1) & (j > 0) & (j < g - 1) & (k > 0) & (k < gz - 1)
    ptr = np.zeros(N + 1, dtype=np.int64); ptr[1:] = np.cumsum(np.where(inner, 27, 1))
    col = np.empty(ptr[-1], dtype=np.int32); val = np.empty(ptr[-1], dtype=np.float64)
    b = ptr[:-1][inner]; e = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                col[b + e] = (idx[inner] + dz * g * g + dy * g + dx).astype(np.int32)
                val[b + e] = 26.0 if (dx, dy, dz) == (0, 0, 0) else -1.0 - 0.125 * (dx == 0) - 0.25 * (dy == 0)
                e += 1
    col[ptr[:-1][~inner]] = idx[~inner].astype(np.int32); val[ptr[:-1][~inner]] = 1.0
    return ptr.astype(np.int32), col, val


def test_wide_value_coded_slices_share_a_dictionary(T, oracle, built_lib):
    """Round 6: value-coded SELL-512 slices WIDER than eight columns (a constant-coefficient 27-point operator: 27 diagonals, a handful of
    values) are pooled in the slice dictionary too -- 28 KiB of codes per distinct slice instead of 54 bytes per row (320^3: 1.06 -> 0.52 ms with the eight-column trips of the any-width kernel,
    profiles/r06_widen_probe.json).  Product, '+=', the multi-vector product: bit for bit the CSR loop."""
    for g, gz in ((64, None), (40, None), (48, 47), (40, 83)):          # (48 x 48 x 47, 40 x 40 x 83: a ragged last slice)
        ptr, col, val = _stencil27_const(g, gz)
        m = len(ptr) - 1
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val))
        assert A.storage == "sell8v" and int(A.info.ell_width) == 27, (A.storage, A.reason)
        assert A.dictionary_blocks > 0 and A.product == "sell8v_runs_kernel", (A.dictionary_blocks, A.product, A.reason)
        x = oracle.random_f64(21, m); y0 = oracle.random_f64(22, m)
        y = T.up(np.full(m, np.nan)); A.apply(T.up(x), y)
        assert np.array_equal(y.cpu().numpy(), oracle.spmv_csr(ptr, col, val, x))
        y = T.up(y0); A.apply(T.up(x), y, -0.5, True)
        assert np.array_equal(y.cpu().numpy(), oracle.spmv_csr(ptr, col, val, x, y=y0.copy(), alpha=-0.5, append=True))
        xs = [oracle.random_f64(30 + k, m) for k in range(3)]
        ys = [T.up(np.full(m, np.nan)) for _ in range(3)]
        A.apply_multi([T.up(v) for v in xs], ys)
        for v, w in zip(xs, ys):
            assert np.array_equal(w.cpu().numpy(), oracle.spmv_csr(ptr, col, val, v))
        B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), dictionary=False)          # one code block per slice: the same bits
        assert B.dictionary_blocks == 0
        y = T.up(np.full(m, np.nan)); B.apply(T.up(x), y)
        assert np.array_equal(y.cpu().numpy(), oracle.spmv_csr(ptr, col, val, x))
        # Inf / NaN in x where no entry of a row looks: never multiplied
        xi = x.copy(); xi[0] = np.inf; xi[m - 1] = np.nan
        y = T.up(np.full(m, np.nan)); A.apply(T.up(xi), y)
        assert np.array_equal(y.cpu().numpy(), oracle.spmv_csr(ptr, col, val, xi), equal_nan=True)
        # round 6, the runs product (sell8.hip sell8v_runs_kernel: entries decoded once per distinct slice -- masks and a value per wave
        # and column --, a request per triple of consecutive diagonals): y = alpha A x + beta z in the same pass, and the float matrix
        xd, zd = T.up(x), T.up(y0.copy())
        want = oracle.spmv_csr(ptr, col, val, x)
        for z, zh in ((zd, y0), (xd, x)):
            ya = T.torch.empty_like(xd)
            A.apply_axpby(xd, ya, -0.75, z, 2.5)
            assert np.array_equal(ya.cpu().numpy(), 2.5 * zh + (-0.75) * want)
        f32 = np.float32
        F = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val.astype(f32)))
        assert F.product == "sell8v_runs_kernel", (F.product, F.reason)
        yf = T.up(np.full(m, np.nan, dtype=f32)); F.apply(T.up(x.astype(f32)), yf)
        assert np.array_equal(yf.cpu().numpy(), oracle.spmv_csr(ptr, col, val.astype(f32), x.astype(f32)))
        # a value that is not finite: its column keeps the codes' path (Inf times the +0.0 of an absent entry would be NaN)
        vi = val.copy(); vi[vi == 26.0] = np.inf
        I = T.ops.SpMat(T.up(ptr), T.up(col), T.up(vi))
        y = T.up(np.full(m, np.nan)); I.apply(T.up(x), y)
        assert np.array_equal(y.cpu().numpy(), oracle.spmv_csr(ptr, col, vi, x), equal_nan=True)
    # 19 points (no corners: triples and single columns mixed), natural boundaries (rows of 8 .. 19 entries: columns of mixed diagonals),
    # values that differ from row to row inside a column (two alternating coefficients: no single value per column)
    g = 48
    N = g ** 3
    idx = np.arange(N, dtype=np.int64)
    i, j, k = idx % g, (idx // g) % g, idx // (g * g)
    offs = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if abs(dx) + abs(dy) + abs(dz) <= 2]
    assert len(offs) == 19
    for alternating in (False, True):
        has = np.stack([(i + dx >= 0) & (i + dx < g) & (j + dy >= 0) & (j + dy < g) & (k + dz >= 0) & (k + dz < g) for dx, dy, dz in offs], axis=1)
        cols = idx[:, None] + np.array([dz * g * g + dy * g + dx for dx, dy, dz in offs], dtype=np.int64)[None, :]
        vals = np.where(np.array([o == (0, 0, 0) for o in offs])[None, :], 18.0, -1.0 - 0.5 * np.array([abs(dx) + abs(dy) + abs(dz) for dx, dy, dz in offs], dtype=np.float64)[None, :]) * np.ones((N, 1))
        if alternating:
            vals = vals * np.where((idx % 3 == 0)[:, None], 1.0, 0.5)
        ptr = np.zeros(N + 1, dtype=np.int64); ptr[1:] = np.cumsum(has.sum(axis=1))
        ptr, col, val = ptr.astype(np.int32), cols[has].astype(np.int32), vals[has]
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val))
        assert A.storage == "sell8v" and int(A.info.ell_width) == 19 and A.dictionary_blocks > 0, (A.storage, A.reason)
        x = oracle.random_f64(77, N); y0 = oracle.random_f64(78, N)
        y = T.up(np.full(N, np.nan)); A.apply(T.up(x), y)
        assert np.array_equal(y.cpu().numpy(), oracle.spmv_csr(ptr, col, val, x)), alternating
        y = T.up(y0.copy()); A.apply(T.up(x), y, 1.5, True)
        assert np.array_equal(y.cpu().numpy(), oracle.spmv_csr(ptr, col, val, x, y=y0.copy(), alpha=1.5, append=True)), alternating


def test_product_with_a_vector_added_is_bit_identical(T, oracle, built_lib):
    """Round 6, vexhip_spmat_apply_axpby_f64: y = alpha A x + beta z in one call.  The plane product adds the vector in its own pass
    (z an array of its own, z = y, z = x taken from the registers that hold the centre lines), every other storage runs y = beta z and
    then y += alpha A x inside the call; both are round(beta z_i) + round(alpha (A x)_i) with (A x)_i summed in CSR order: compared bit
    for bit with that evaluation on the host (the CSR restatement of spmat/csr.inl:163-170)."""
    torch = T.torch
    # (geometry, format, storage by grid line forced, keyword arguments, the product expected, one pass expected)
    cases = (((512, 6, 8), None, True, {}, "sell8_plane_kernel", True), ((70, 11, 13), None, True, {}, "sell8_grid_kernel", True), ((1030, 5, 9), None, True, {}, "sell8_grid_kernel", True),
             ((512, 6, 8), "csr", False, {}, "csr_stream2_kernel", True),                   # the CSR kernels add it in their own epilogue
             ((512, 6, 8), "sell32", False, {}, "sell_pair_kernel", True),                  # the SELL-family kernels add it in store_pair
             ((512, 6, 8), "sell8", False, {}, "sell8_pair_kernel", True),
             ((512, 6, 8), "sell8", False, {"dictionary": False}, "sell8_pair_kernel", True),
             ((64, 16, 16), None, False, {"march": False}, "sell8_pair_kernel", True),       # value codes from the slice dictionary, pair product
             ((64, 16, 16), None, False, {"dictionary": False}, "sell8_pair_kernel", True),  # value codes, one block per slice
             ((64, 64, 64), None, False, {"march": False}, "sell8_pair_kernel", True),       # the benchmark's operator: slice dictionary, pair product
             ((64, 64, 64), None, False, {}, "sell8_march_kernel", True))                   # the march product: in its hot loop's epilogue (z and beta from the cold arguments)
    for (nx, ny, nz), fmt, force, kw, product, one_pass in cases:
        ptr, col, val = oracle.poisson3d(nx) if (nx, ny, nz) == (64, 64, 64) else _grid7_natural(nx, ny, nz, zero_face=False)
        m = len(ptr) - 1
        x = oracle.random_f64(11, m); z = oracle.random_f64(12, m); y0 = oracle.random_f64(13, m)
        if force:
            os.environ["VEXHIP_PLANE_FORCE"] = "1"
        try:
            A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), **kw) if fmt is None else T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), fmt=fmt, **kw)
        finally:
            os.environ.pop("VEXHIP_PLANE_FORCE", None)
        assert A.product == product, (fmt, kw, A.product, A.reason)
        dx = T.up(x)
        fused = built_lib.spmat_axpby_fused(A.handle, T.ops._p(dx), T.ops._p(dx), T.ops._p(T.up(y0)))
        assert bool(fused) == one_pass, (fmt, kw, product, fused)
        for alpha, beta in ((1.0, 1.0), (-1.0, 1.0), (2.0, 1.0), (0.5, -0.25), (-3.0, 0.0)):
            ax = oracle.spmv_csr(ptr, col, val, x, alpha=alpha)
            # z an array of its own
            dz, dy = T.up(z), T.up(np.full(m, np.nan))
            A.apply_axpby(dx, dy, alpha, dz, beta)
            assert np.array_equal(dy.cpu().numpy(), beta * z + ax), (fmt, kw, alpha, beta, "z")
            # z = x
            dy = T.up(np.full(m, np.nan))
            A.apply_axpby(dx, dy, alpha, dx, beta)
            assert np.array_equal(dy.cpu().numpy(), beta * x + ax), (fmt, alpha, beta, "z = x")
            # z = y
            dy = T.up(y0)
            A.apply_axpby(dx, dy, alpha, dy, beta)
            assert np.array_equal(dy.cpu().numpy(), beta * y0 + ax), (fmt, alpha, beta, "z = y")
    with pytest.raises(Exception):
        A.apply_axpby(dx, dx, 1.0, dz, 1.0)          # y = x: refused
    # vectors that start at an odd element (views): the plane product wants 16-byte addresses -- x or y odd: the grid product takes the
    # addend; only z odd: two passes inside the call; the same bits every time
    ptr, col, val = _grid7_natural(512, 6, 8, zero_face=False)
    m = len(ptr) - 1
    os.environ["VEXHIP_PLANE_FORCE"] = "1"
    try:
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val))
    finally:
        os.environ.pop("VEXHIP_PLANE_FORCE", None)
    x = oracle.random_f64(31, m); z = oracle.random_f64(32, m)
    ax = oracle.spmv_csr(ptr, col, val, x, alpha=-1.0)
    for ox, oz, oy in ((0, 1, 0), (1, 0, 0), (0, 0, 1), (1, 1, 1)):
        bx = T.up(np.concatenate([np.zeros(ox), x])); bz = T.up(np.concatenate([np.zeros(oz), z])); by = T.up(np.full(m + oy, np.nan))
        A.apply_axpby(bx[ox:], by[oy:], -1.0, bz[oz:], 0.5)
        assert np.array_equal(by[oy:].cpu().numpy(), 0.5 * z + ax), (ox, oz, oy)
    # a matrix without entries: y = beta z
    E0 = T.ops.SpMat(T.up(np.zeros(m + 1, dtype=np.int32)), T.up(np.zeros(0, dtype=np.int32)), T.up(np.zeros(0)), n_cols=m)
    dy = T.up(np.full(m, np.nan)); E0.apply_axpby(T.up(x), dy, 3.0, T.up(z), -2.0)
    assert np.array_equal(dy.cpu().numpy(), -2.0 * z)
    # float matrices: the fp32 plane and grid products take the addend as well; every rounding is a float's
    f32 = np.float32
    for (nx, ny, nz), product in (((512, 6, 8), "sell8_plane_f32_kernel"), ((70, 11, 13), "sell8_grid_f32_kernel"), ((1030, 5, 9), "sell8_grid_f32_kernel")):
        ptr, col, val = _grid7_natural(nx, ny, nz, zero_face=False)
        m = len(ptr) - 1
        v32 = val.astype(f32)
        x = oracle.random_f64(11, m).astype(f32); z = oracle.random_f64(12, m).astype(f32); y0 = oracle.random_f64(13, m).astype(f32)
        os.environ["VEXHIP_PLANE_FORCE"] = "1"
        try:
            A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32))
        finally:
            os.environ.pop("VEXHIP_PLANE_FORCE", None)
        assert A.product == product, (A.product, A.reason)
        dx = T.up(x)
        assert built_lib.spmat_axpby_fused(A.handle, T.ops._p(dx), T.ops._p(dx), T.ops._p(T.up(y0)))
        for alpha, beta in ((1.0, 1.0), (-1.0, 1.0), (0.5, -0.25)):
            ax = oracle.spmv_csr(ptr, col, v32, x, alpha=alpha)
            dy = T.up(np.full(m, np.nan, dtype=f32)); A.apply_axpby(dx, dy, alpha, T.up(z), beta)
            assert np.array_equal(dy.cpu().numpy(), f32(beta) * z + ax), (product, alpha, beta, "z")
            dy = T.up(np.full(m, np.nan, dtype=f32)); A.apply_axpby(dx, dy, alpha, dx, beta)
            assert np.array_equal(dy.cpu().numpy(), f32(beta) * x + ax), (product, alpha, beta, "z = x")
            dy = T.up(y0); A.apply_axpby(dx, dy, alpha, dy, beta)
            assert np.array_equal(dy.cpu().numpy(), f32(beta) * y0 + ax), (product, alpha, beta, "z = y")


def test_grid_product_fp32_is_bit_identical(T, oracle, built_lib):
    """The fp32 grid product (round 5, grid32.hip: the walk of the fp64 grid product with FOUR rows per lane -- 16-byte requests at
    4-byte addresses, the lane at the end of a line stores 1 .. 3 rows) against the pair product and the fp32 CSR restatement,
    bit for bit: the benchmark's operator on 96^3 and 125^3, line lengths that are no multiple of four (70, 125, 127, 250, 514,
    700, 1030 -- two segments), odd lines per plane, natural boundaries (nine classes, explicit zeros), full bands whose +-1
    diagonal crosses the line ends, a ragged last plane, walk depths, '=' and '+= alpha', Inf / NaN in x, vectors that start at
    any element; both set-ups (one pass over the CSR arrays, from the SELL-512 storage); the plan as the library chooses it (208^3)."""
    torch = T.torch
    f32 = np.float32
    try:
        def check(ptr, col, val, shape, seed, depth=None, classes=None):
            if depth is None:
                os.environ.pop("VEXHIP_GRID32_DEPTH", None)
            else:
                os.environ["VEXHIP_GRID32_DEPTH"] = str(depth)
            m = len(ptr) - 1
            v32 = val.astype(f32)
            B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), march=False)
            assert B.grid is None and B.plane is None and B.march is None
            xb = oracle.random_f64(seed, m).astype(f32); y0 = oracle.random_f64(seed + 1, m).astype(f32)
            want = oracle.spmv_csr(ptr, col, v32, xb)
            for direct in (True, False):
                A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), direct=direct)
                assert A.storage == "sell8v" and A.grid is not None and A.plane is None and A.direct == direct, (shape, direct, A.storage)
                assert (A.grid["nx"], A.grid["lines_per_plane"]) == shape[:2] and A.grid["x_last"] == m - 1, (shape, A.grid)
                if classes is not None:
                    assert A.grid["classes"] == classes, (shape, A.grid)
                for alpha, append in ((1.0, False), (-0.75, True)):
                    ya, yb = T.up(y0.copy()), T.up(y0.copy())
                    A.apply(T.up(xb), ya, alpha, append); B.apply(T.up(xb), yb, alpha, append)
                    assert torch.equal(ya, yb), (shape, alpha, direct)
                    assert np.array_equal(ya.cpu().numpy(), (y0 + f32(alpha) * want) if append else f32(alpha) * want), (shape, alpha, direct)
            return A

        os.environ["VEXHIP_PLANE_FORCE"] = "1"
        for n in (96, 125):
            ptr, col, val = oracle.poisson3d(n)
            check(ptr, col, val, (n, n, n), 31, classes=2)
        for shape, depth in (((70, 33, 20), None), ((70, 33, 20), 3), ((1030, 6, 8), None), ((1030, 6, 8), 5), ((384, 10, 12), None),
                             ((500, 7, 11), 4), ((127, 17, 19), 7), ((514, 5, 13), None), ((640, 6, 9), None), ((700, 5, 8), 3),
                             ((1000, 4, 7), 2), ((1024, 6, 6), None), ((258, 6, 40), 16), ((9, 40, 41), None)):
            ptr, col, val = _grid7(*shape)
            check(ptr, col, val, shape, 33, depth, classes=2)
        for shape, depth in (((96, 20, 21), None), ((125, 15, 18), 5), ((250, 9, 16), None)):
            ptr, col, val = _grid7_natural(*shape, zero_face=True)
            A = check(ptr, col, val, shape, 35, depth)
            assert A.grid["classes"] >= 9, A.grid
        for nx, ny, nz, extra_lines, depth in ((96, 12, 30, 0, None), (125, 9, 33, 4, 6), (300, 8, 14, 3, None), (1021, 4, 9, 2, None)):
            P = nx * ny; m = P * nz + extra_lines * nx
            ptr, col, val = _band(m, (-P, -nx, -1, 0, 1, nx, P), 5, constant=True)
            check(ptr, col, val, (nx, ny, nz), 37, depth)
        os.environ.pop("VEXHIP_GRID32_DEPTH", None)
        # Inf / NaN in x: only the rows that reference them may see them; a stored 0.0 times Inf is NaN as in the CSR loop
        shape = (125, 15, 18)
        ptr, col, val = _grid7_natural(*shape, zero_face=True)
        v32 = val.astype(f32)
        m = len(ptr) - 1
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32))
        assert A.grid is not None
        xb = oracle.random_f64(39, m).astype(f32)
        xb[0] = np.inf; xb[1] = -np.inf; xb[m - 1] = np.nan; xb[125 * 15 + 7 * 125 + 60] = np.inf; xb[5 * 125 * 15 + 124] = np.nan
        ya = torch.empty(m, dtype=torch.float32, device=T.dev)
        A.apply(T.up(xb), ya)
        want = oracle.spmv_csr(ptr, col, v32, xb)
        assert np.isnan(want).sum() >= 8
        assert np.array_equal(ya.cpu().numpy(), want, equal_nan=True)
        # vectors that start at any element
        xb = oracle.random_f64(27, m).astype(f32); y0 = oracle.random_f64(28, m).astype(f32)
        want = oracle.spmv_csr(ptr, col, v32, xb)
        for alpha, append in ((1.0, False), (-0.75, True)):
            xbig = torch.zeros(m + 7, dtype=torch.float32, device=T.dev); ybig = torch.zeros(m + 7, dtype=torch.float32, device=T.dev)
            for xo, yo in ((1, 1), (2, 0), (0, 3)):
                xv, yv = xbig[xo:xo + m], ybig[yo:yo + m]
                xv.copy_(T.up(xb)); yv.copy_(T.up(y0))
                A.apply(xv, yv, alpha, append)
                assert np.array_equal(yv.cpu().numpy(), (y0 + f32(alpha) * want) if append else f32(alpha) * want), (alpha, xo, yo)
        os.environ.pop("VEXHIP_PLANE_FORCE")
        # the plan as the library chooses it: 208^3
        n = 208
        ptr, col, val = oracle.poisson3d(n)
        v32 = val.astype(f32)
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32)); B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), march=False)
        assert A.direct and A.grid is not None and A.grid["nx"] == n and A.grid["classes"] == 2 and B.grid is None, A.grid
        xb = oracle.random_f64(47, n ** 3).astype(f32)
        ya = torch.empty(n ** 3, dtype=torch.float32, device=T.dev); yb = torch.empty_like(ya)
        A.apply(T.up(xb), ya); B.apply(T.up(xb), yb)
        assert torch.equal(ya, yb)
        assert np.array_equal(ya.cpu().numpy(), oracle.spmv_csr(ptr, col, v32, xb))
    finally:
        for k in ("VEXHIP_PLANE_FORCE", "VEXHIP_GRID32_DEPTH"):
            os.environ.pop(k, None)


def test_structured_products_on_random_shapes(T, oracle, built_lib):
    """Round 5: the four products of matrices stored by grid line -- plane / grid in fp64, plane32 / grid32 in fp32 -- on sixty
    seeded random grids (line lengths 8 .. 1100 incl. 512 and lengths that are no multiple of four, 2 .. 13 lines per plane, 4 .. 26
    planes, a ragged last plane now and then, random walk depths, natural or Dirichlet boundaries or full bands), '=' and
    '+= alpha', against the CSR restatement bit for bit.  What the hand-picked shapes of the tests above may have missed: a chunk
    boundary next to a class change, the last lane of a line with 1 .. 3 rows, odd lines per plane with short walks."""
    torch = T.torch
    rng = np.random.default_rng(int(os.environ.get("VEXHIP_TEST_SEED", "20250925")))      # (another seed: another sixty grids)
    keys = ("VEXHIP_PLANE_DEPTH", "VEXHIP_PLANE32_DEPTH", "VEXHIP_GRID32_DEPTH")
    os.environ["VEXHIP_PLANE_FORCE"] = "1"
    seen = set()
    try:
        for case in range(60):
            nx = int(rng.choice([512, 512, int(rng.integers(8, 1101)), int(rng.integers(8, 300)), 4 * int(rng.integers(3, 260)) + int(rng.integers(1, 4))]))
            ny = int(rng.integers(2, 14)); nz = int(rng.integers(4, 27))
            kind = int(rng.integers(0, 3))
            extra = int(rng.integers(0, ny)) if kind == 2 and rng.random() < 0.5 else 0
            if kind == 0:
                ptr, col, val = _grid7(nx, ny, nz)
            elif kind == 1:
                ptr, col, val = _grid7_natural(nx, ny, nz, zero_face=bool(rng.integers(0, 2)))
            else:
                P = nx * ny
                ptr, col, val = _band(P * nz + extra * nx, (-P, -nx, -1, 0, 1, nx, P), int(rng.integers(1, 100)), constant=True)
            m = len(ptr) - 1
            depth = None if rng.random() < 0.4 else int(rng.integers(1, nz + 1))
            for k in keys:
                if depth is None: os.environ.pop(k, None)
                else: os.environ[k] = str(depth)
            for dt in (np.float64, np.float32):
                v = val.astype(dt)
                xb = oracle.random_f64(100 + case, m).astype(dt); y0 = oracle.random_f64(200 + case, m).astype(dt)
                want = oracle.spmv_csr(ptr, col, v, xb)
                direct = bool(rng.integers(0, 2))
                A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v), direct=direct)
                if A.grid is None and A.plane is None:       # (hybrid-ELL width 1 on grids that are mostly boundary: a CSR tail, no grid plan)
                    seen.add("none")
                else:
                    seen.add(("plane" if A.plane else "grid") + ("32" if dt == np.float32 else "64"))
                for alpha, append in ((1.0, False), (-1.25, True)):
                    ya = T.up(y0.copy())
                    A.apply(T.up(xb), ya, alpha, append)
                    ref = (y0 + dt(alpha) * want) if append else dt(alpha) * want
                    assert np.array_equal(ya.cpu().numpy(), ref), (case, nx, ny, nz, kind, extra, depth, dt.__name__, direct, alpha, A.grid, A.plane)
        assert {"plane64", "plane32", "grid64", "grid32"} <= seen, seen
    finally:
        for k in keys + ("VEXHIP_PLANE_FORCE",):
            os.environ.pop(k, None)


def _virtual_line(W):
    """grid.hip grid_diagonals: the virtual line a 2-D row of W points is cut into -- 512 where an even number (>= 4) of them fit (the
    plane product), else the longest even divisor of W in [128, min(1024, W / 10)] (a flat grid plan), else 0 (the pair product)"""
    if W % 512 == 0 and (W // 512) % 2 == 0 and W // 512 >= 4:
        return 512
    for d in range(min(1024, W // 10), 127, -1):
        if W % d == 0 and d % 2 == 0:
            return d
    return 0


def test_two_dimensional_five_point_operators(T, oracle, built_lib):
    """Round 5: 5-point operators on 2-D grids -- diagonals {0, +-1, +-W}, no line-above / line-below pair -- take the
    plane product along VIRTUAL 512-point lines where the rows are an even number (>= 4) of them (grid.hip grid_diagonals: +-W as
    the far pair; the reference's SpMatCCSR covers such operators without a notion of dimension, spmat/ccsr.hpp:55-113).  Round 6:
    rows of any other length are cut into virtual lines of their longest even divisor up to 1024 points (ten or more per row) and
    take the grid product with a FLAT plan -- the walk requests no lines above / below a tile, an XCD owns walks instead of tiles
    (12 000^2: 0.46 ms against 0.68 through the march product, profiles/r06_2d.json); rows without such a divisor keep the pair
    product.  Bit for bit against the pair product and the CSR restatement: natural boundaries (rows of 3, 4 and 5 entries), an odd
    number of lines per row, '=' and '+= alpha', y = alpha A x + beta z in one pass, through the one-pass set-up and the SELL-512
    storage, fp64 and fp32."""
    torch = T.torch
    os.environ["VEXHIP_PLANE_FORCE"] = "1"
    try:
        for W, H, nx, plane in ((1000, 300, 0, False), (96, 700, 0, False), (2048, 120, 512, True), (4096, 64, 512, True), (1536, 90, 128, False), (3072, 40, 512, True),
                                (2000, 150, 200, False), (12000, 24, 1000, False), (9000, 37, 900, False), (1400, 100, 140, False), (5632, 41, 512, False)):
            assert _virtual_line(W) == nx == T.ops.lib().sell8_grid_virtual_line(W), (W, nx, _virtual_line(W))
            ptr, col, val = _grid7_natural(W, H, 1)
            m = len(ptr) - 1
            assert set((col - np.repeat(np.arange(m), np.diff(ptr))).tolist()) == {-W, -1, 0, 1, W}
            xb = oracle.random_f64(W, m); y0 = oracle.random_f64(W + 1, m)
            want = oracle.spmv_csr(ptr, col, val, xb)
            B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), march=False)
            assert B.grid is None and B.plane is None
            for direct in (True, False):
                A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), direct=direct)
                assert A.storage == "sell8v", (W, H, A.storage)
                if plane:
                    assert A.plane is not None and A.plane["lines_per_plane"] == W // 512 and A.plane["planes"] == H and A.plane["flat"] == 1, (W, H, direct, A.plane, A.grid)
                elif nx:           # virtual lines of a divisor of W: the grid product, flat
                    assert A.plane is None and A.grid is not None and A.grid["nx"] == nx and A.grid["lines_per_plane"] == W // nx and A.grid["planes"] == H \
                        and A.grid["flat"] == 1 and A.product == "sell8_grid_kernel", (W, H, direct, A.grid, A.product)
                else:              # no such divisor: the pair product
                    assert A.grid is None and A.plane is None, (W, H, direct, A.grid)
                for alpha, append in ((1.0, False), (-0.75, True)):
                    ya, yb = T.up(y0.copy()), T.up(y0.copy())
                    A.apply(T.up(xb), ya, alpha, append); B.apply(T.up(xb), yb, alpha, append)
                    assert torch.equal(ya, yb), (W, H, alpha, direct)
                    assert np.array_equal(ya.cpu().numpy(), (y0 + alpha * want) if append else alpha * want), (W, H, alpha, direct)
                # y = alpha A x + beta z in one pass (z an array / x itself): round(beta z) + round(alpha (A x)_i), as the host forms it
                xd, zd = T.up(xb), T.up(y0.copy())
                for z, zh in ((zd, y0), (xd, xb)):
                    ya = torch.empty_like(xd)
                    A.apply_axpby(xd, ya, -0.75, z, 2.5)
                    assert np.array_equal(ya.cpu().numpy(), 2.5 * zh + (-0.75) * want), (W, H, direct, "axpby")
            # the same operator in fp32: the fp32 plane product where rows are virtual 512-point lines, the pair product elsewhere
            f32 = np.float32
            v32, x32, y32 = val.astype(f32), xb.astype(f32), y0.astype(f32)
            want32 = oracle.spmv_csr(ptr, col, v32, x32)
            for direct in (True, False):
                A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), direct=direct)
                assert (A.plane is not None) == plane, (W, H, direct, A.plane, A.grid)
                if plane:
                    assert A.direct == direct and A.plane["lines_per_plane"] == W // 512 and A.plane["planes"] == H, (W, H, direct, A.plane)
                elif nx:
                    assert A.grid is not None and A.grid["nx"] == nx and A.grid["flat"] == 1 and A.product == "sell8_grid_f32_kernel", (W, H, direct, A.grid, A.product)
                for alpha, append in ((1.0, False), (-0.75, True)):
                    ya = T.up(y32.copy())
                    A.apply(T.up(x32), ya, alpha, append)
                    assert np.array_equal(ya.cpu().numpy(), (y32 + f32(alpha) * want32) if append else f32(alpha) * want32), (W, H, alpha, direct, "fp32")
        # a full band {0, +-1, +-W} (no boundary rows: the +-1 diagonal crosses the ends of the rows and of the virtual lines, the first
        # and the last rows lack their far entries), a ragged number of rows per walk, Inf in x under a position without an entry
        W, m = 2000, 2000 * 37
        ptr, col, val = _band(m, (-W, -1, 0, 1, W), 11, constant=True)
        xb = oracle.random_f64(5, m); xb[0] = np.inf
        want = oracle.spmv_csr(ptr, col, val, xb)
        for direct in (True, False):
            A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), direct=direct)
            assert A.grid is not None and A.grid["flat"] == 1 and A.grid["nx"] == 200 and A.grid["planes"] == 37, (direct, A.grid)
            assert np.array_equal((A @ T.up(xb)).cpu().numpy(), want, equal_nan=True), direct
    finally:
        os.environ.pop("VEXHIP_PLANE_FORCE", None)


def test_plane_product_fp32_is_bit_identical(T, oracle, built_lib):
    """The fp32 plane product (round 5, plane32.hip: the storage and the walk of the fp64 plane product with FOUR rows per lane, a
    workgroup of two waves per pair of grid lines) against the pair product and the fp32 CSR restatement, bit for bit: small
    banded matrices through the forced plan (five dictionary blocks, lines that change their block from plane to plane, a
    ragged last plane, walks shorter than a group of four steps, several walk depths), '=' and '+= alpha'; the plan as the
    library chooses it (a 512 x 64 x 80 band, the benchmark's operator on 512 x 72 x 72 with identity rows on every face);
    Inf / NaN in x where positions without an entry 'cover' it; vectors at addresses that are not multiples of 16 bytes;
    both stored forms (dictionary blocks of the SELL-512 storage, class tables of the storage by grid line -- grid.hip's
    one-pass build, which takes float values where this product applies)."""
    torch = T.torch
    f32 = np.float32
    try:
        os.environ["VEXHIP_PLANE_FORCE"] = "1"
        for ny, nz, extra, depth in ((8, 12, 0, None), (4, 40, 0, None), (16, 9, 3 * 512, None), (12, 33, 5 * 512, 7), (8, 21, 0, 3), (6, 50, 512, 16)):
            if depth is None:
                os.environ.pop("VEXHIP_PLANE32_DEPTH", None)
            else:
                os.environ["VEXHIP_PLANE32_DEPTH"] = str(depth)
            P = 512 * ny
            m = P * nz + extra
            ptr, col, val = _band(m, (-P, -512, -1, 0, 1, 512, P), 5, constant=True)
            v32 = val.astype(f32)
            B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), march=False)
            assert B.plane is None and B.march is None
            xb = oracle.random_f64(21, m).astype(f32); y0 = oracle.random_f64(22, m).astype(f32)
            want = oracle.spmv_csr(ptr, col, v32, xb)
            assert want.dtype == f32
            for direct in (False, True):       # dictionary blocks of the SELL-512 storage / class tables of the storage by grid line
                A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), direct=direct)
                assert A.storage == "sell8v" and A.plane is not None and (A.plane["table_pitch"] > 0) == direct and A.direct == direct, (ny, nz, direct, A.plane)
                assert A.plane["lines_per_plane"] == ny
                for alpha, append in ((1.0, False), (-0.75, True)):
                    ya, yb = T.up(y0.copy()), T.up(y0.copy())
                    A.apply(T.up(xb), ya, alpha, append); B.apply(T.up(xb), yb, alpha, append)
                    assert torch.equal(ya, yb), (ny, nz, alpha, direct)
                    assert np.array_equal(ya.cpu().numpy(), (y0 + f32(alpha) * want) if append else f32(alpha) * want), (ny, nz, alpha, direct)
        os.environ.pop("VEXHIP_PLANE32_DEPTH", None)
        # Inf / NaN in x: only the rows that reference them may see them
        ny, nz = 8, 12
        P = 512 * ny; m = P * nz
        ptr, col, val = _band(m, (-P, -512, -1, 0, 1, 512, P), 5, constant=True)
        v32 = val.astype(f32)
        xb = oracle.random_f64(23, m).astype(f32)
        xb[0] = np.inf; xb[1] = -np.inf; xb[m - 1] = np.nan; xb[5 * P + 3 * 512 + 255] = np.nan; xb[7 * P + 2 * 512 + 256] = np.inf
        for direct in (False, True):
            A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), direct=direct)
            assert A.plane is not None and A.direct == direct
            ya = torch.empty(m, dtype=torch.float32, device=T.dev)
            A.apply(T.up(xb), ya)
            assert np.array_equal(ya.cpu().numpy(), oracle.spmv_csr(ptr, col, v32, xb), equal_nan=True), direct
        # vectors that start at any element (views into larger vectors): the fp32 plane product issues its 16-byte requests at
        # 4-byte addresses -- a matrix stored by grid line has no other product
        xb = oracle.random_f64(27, m).astype(f32); y0 = oracle.random_f64(28, m).astype(f32)
        want = oracle.spmv_csr(ptr, col, v32, xb)
        for direct in (False, True):
            A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), direct=direct)
            for alpha, append in ((1.0, False), (-0.75, True)):
                xbig = torch.zeros(m + 7, dtype=torch.float32, device=T.dev); ybig = torch.zeros(m + 7, dtype=torch.float32, device=T.dev)
                for xo, yo in ((1, 1), (2, 0), (0, 3), (4, 4)):
                    xv, yv = xbig[xo:xo + m], ybig[yo:yo + m]
                    xv.copy_(T.up(xb)); yv.copy_(T.up(y0))
                    A.apply(xv, yv, alpha, append)
                    assert np.array_equal(yv.cpu().numpy(), (y0 + f32(alpha) * want) if append else f32(alpha) * want), (direct, alpha, xo, yo)
        os.environ.pop("VEXHIP_PLANE_FORCE")

        # the plan as the library chooses it
        ny, nz = 64, 80
        P = 512 * ny; m = P * nz
        ptr, col, val = _band(m, (-P, -512, -1, 0, 1, 512, P), 6, constant=True)
        v32 = val.astype(f32)
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32)); B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), plane=False)
        assert A.plane is not None and A.plane["lines_per_plane"] == ny and A.plane["planes"] == nz and B.plane is None and B.march is not None
        xb = oracle.random_f64(24, m).astype(f32)
        ya = torch.empty(m, dtype=torch.float32, device=T.dev); yb = torch.empty_like(ya)
        A.apply(T.up(xb), ya); B.apply(T.up(xb), yb)
        assert torch.equal(ya, yb)
        assert np.array_equal(ya.cpu().numpy(), oracle.spmv_csr(ptr, col, v32, xb))
        ptr, col, val = _grid7(512, 72, 72)
        v32 = val.astype(f32)
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32)); B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), march=False)
        assert A.storage == "sell8v" and A.dictionary_blocks == 2 and A.plane is not None and A.plane["lines_per_plane"] == 72, (A.plane, A.dictionary_blocks)
        m = 512 * 72 * 72
        xb = oracle.random_f64(25, m).astype(f32); y0 = oracle.random_f64(26, m).astype(f32)
        want = oracle.spmv_csr(ptr, col, v32, xb)
        for alpha, append in ((1.0, False), (2.5, True)):
            ya, yb = T.up(y0.copy()), T.up(y0.copy())
            A.apply(T.up(xb), ya, alpha, append); B.apply(T.up(xb), yb, alpha, append)
            assert torch.equal(ya, yb), alpha
            assert np.array_equal(ya.cpu().numpy(), (y0 + f32(alpha) * want) if append else f32(alpha) * want), alpha
    finally:
        for k in ("VEXHIP_PLANE_FORCE", "VEXHIP_PLANE32_DEPTH"):
            os.environ.pop(k, None)


@pytest.mark.parametrize("tile", [2, 4])
def test_plane_product_is_bit_identical(T, oracle, built_lib, tile):
    """The plane product (round 4: a workgroup owns `tile` grid lines of 512 points and walks through the planes; the +-512 and
    +-P neighbours of a lane's rows stay in registers, the +-1 neighbours come by DPP wave shifts) against the pair product AND
    the CSR restatement, bit for bit.  (a) small banded matrices through the forced plan (VEXHIP_PLANE_FORCE: five dictionary
    blocks, lines that change their block from plane to plane, a ragged last plane, walks shorter than a group of four
    steps), '=' and '+= alpha'; (b) the plan as the library chooses it on a 512 x 64 x 80 band and on the benchmark's
    operator on a 512 x 72 x 72 grid (identity rows on every face: the hot block + one other block per line and plane);
    (c) x holding Inf / NaN where positions without an entry 'cover' it; (d) what the plan must decline."""
    torch = T.torch
    os.environ["VEXHIP_PLANE_TILE"] = str(tile)
    try:
        os.environ["VEXHIP_PLANE_FORCE"] = "1"
        for ny, nz, extra, depth in ((8, 12, 0, None), (4, 40, 0, None), (16, 9, 3 * 512, None), (12, 33, 5 * 512, 7), (8, 21, 0, 3)):
            if depth is None:
                os.environ.pop("VEXHIP_PLANE_DEPTH", None)
            else:
                os.environ["VEXHIP_PLANE_DEPTH"] = str(depth)
            P = 512 * ny
            m = P * nz + extra
            ptr, col, val = _band(m, (-P, -512, -1, 0, 1, 512, P), 5, constant=True)
            B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), march=False)
            xb = oracle.random_f64(21, m); y0 = oracle.random_f64(22, m)
            want = oracle.spmv_csr(ptr, col, val, xb)
            for direct in (False, True):       # the plane kernel on dictionary blocks of the SELL-512 storage / on the class tables of the storage by grid line
                A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), direct=direct)
                assert A.storage == "sell8v" and A.plane is not None and B.plane is None and B.march is None and A.direct == direct, (ny, nz, A.plane)
                assert A.plane["lines_per_plane"] == ny and A.plane["tile"] == (tile if ny % tile == 0 else 2), A.plane
                assert (A.plane["table_pitch"] > 0) == direct
                for alpha, append in ((1.0, False), (-0.75, True)):
                    ya, yb = T.up(y0.copy()), T.up(y0.copy())
                    A.apply(T.up(xb), ya, alpha, append); B.apply(T.up(xb), yb, alpha, append)
                    assert torch.equal(ya, yb), (ny, nz, alpha, direct)
                    assert np.array_equal(ya.cpu().numpy(), (y0 + alpha * want) if append else alpha * want), (ny, nz, alpha, direct)
        os.environ.pop("VEXHIP_PLANE_DEPTH", None)
        # (c) Inf / NaN in x: only the rows that reference them may see them
        ny, nz = 8, 12
        P = 512 * ny; m = P * nz
        ptr, col, val = _band(m, (-P, -512, -1, 0, 1, 512, P), 5, constant=True)
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val))
        assert A.plane is not None
        xb = oracle.random_f64(23, m)
        xb[0] = np.inf; xb[1] = -np.inf; xb[m - 1] = np.nan; xb[5 * P + 3 * 512 + 255] = np.nan
        ya = torch.empty(m, dtype=torch.float64, device=T.dev)
        A.apply(T.up(xb), ya)
        assert np.array_equal(ya.cpu().numpy(), oracle.spmv_csr(ptr, col, val, xb), equal_nan=True)
        # vectors that start at an odd element (views into larger vectors: 8-byte, not 16-byte addresses).  The plane product moves
        # 16-byte pieces at 16-byte addresses and is not launched on them: the grid product (matrix stored by grid line -- it has
        # no other storage) or the march / pair products take the call, same bits
        xb = oracle.random_f64(27, m); y0 = oracle.random_f64(28, m)
        want = oracle.spmv_csr(ptr, col, val, xb)
        for direct in (True, False):
            A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), direct=direct)
            assert A.plane is not None and A.direct == direct
            for alpha, append in ((1.0, False), (-0.75, True)):
                xbig = torch.zeros(m + 3, dtype=torch.float64, device=T.dev); ybig = torch.zeros(m + 3, dtype=torch.float64, device=T.dev)
                for xo, yo in ((1, 1), (1, 0), (0, 1)):
                    xv, yv = xbig[xo:xo + m], ybig[yo:yo + m]
                    assert (xv.data_ptr() % 16 != 0) == (xo == 1) and (yv.data_ptr() % 16 != 0) == (yo == 1)
                    xv.copy_(T.up(xb)); yv.copy_(T.up(y0))
                    A.apply(xv, yv, alpha, append)
                    assert np.array_equal(yv.cpu().numpy(), (y0 + alpha * want) if append else alpha * want), (direct, alpha, xo, yo)
        os.environ.pop("VEXHIP_PLANE_FORCE")

        # (b) the plan as the library chooses it
        ny, nz = 64, 80
        P = 512 * ny; m = P * nz
        ptr, col, val = _band(m, (-P, -512, -1, 0, 1, 512, P), 6, constant=True)
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val)); B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), plane=False)
        assert A.plane is not None and A.plane["lines_per_plane"] == ny and A.plane["planes"] == nz and B.plane is None and B.march is not None
        xb = oracle.random_f64(24, m)
        ya = torch.empty(m, dtype=torch.float64, device=T.dev); yb = torch.empty_like(ya)
        A.apply(T.up(xb), ya); B.apply(T.up(xb), yb)
        assert torch.equal(ya, yb)
        assert np.array_equal(ya.cpu().numpy(), oracle.spmv_csr(ptr, col, val, xb))
        ptr, col, val = _grid7(512, 72, 72)
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val)); B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), march=False)
        assert A.storage == "sell8v" and A.dictionary_blocks == 2 and A.plane is not None and A.plane["lines_per_plane"] == 72, (A.plane, A.dictionary_blocks)
        m = 512 * 72 * 72
        xb = oracle.random_f64(25, m); y0 = oracle.random_f64(26, m)
        want = oracle.spmv_csr(ptr, col, val, xb)
        for alpha, append in ((1.0, False), (2.5, True)):
            ya, yb = T.up(y0.copy()), T.up(y0.copy())
            A.apply(T.up(xb), ya, alpha, append); B.apply(T.up(xb), yb, alpha, append)
            assert torch.equal(ya, yb), alpha
            assert np.array_equal(ya.cpu().numpy(), (y0 + alpha * want) if append else alpha * want), alpha

        # (d) declined: an eighth diagonal; rows whose entries do not ascend by diagonal (storage order is not position order); a
        # matrix whose rows do not fill whole lines; lines per plane not even.  (fp32 has its own plane product:
        # test_plane_product_fp32_is_bit_identical)
        P = 512 * 64; m = P * 40
        ptr, col, val = _band(m, (-P, -512, -2, -1, 0, 1, 512, P), 7, constant=True)
        assert T.ops.SpMat(T.up(ptr), T.up(col), T.up(val)).plane is None
        ptr, col, val = _band(m, (-P, -512, -1, 0, 1, 512, P), 7, constant=True)
        assert T.ops.SpMat(T.up(ptr), T.up(col), T.up(val.astype(np.float32))).plane is not None
        rcol, rval = col.copy(), val.copy()
        for r in range(3 * P, 3 * P + 2048):                      # a few rows with their entries reversed
            rcol[ptr[r]:ptr[r + 1]] = col[ptr[r]:ptr[r + 1]][::-1]; rval[ptr[r]:ptr[r + 1]] = val[ptr[r]:ptr[r + 1]][::-1]
        R = T.ops.SpMat(T.up(ptr), T.up(rcol), T.up(rval))
        assert R.plane is None
        xb = oracle.random_f64(27, m)
        assert np.array_equal((R @ T.up(xb)).cpu().numpy(), oracle.spmv_csr(ptr, rcol, rval, xb))
        ptr, col, val = _band(m + 100, (-P, -512, -1, 0, 1, 512, P), 7, constant=True)
        assert T.ops.SpMat(T.up(ptr), T.up(col), T.up(val)).plane is None
        P = 512 * 63
        ptr, col, val = _band(P * 40, (-P, -512, -1, 0, 1, 512, P), 7, constant=True)
        assert T.ops.SpMat(T.up(ptr), T.up(col), T.up(val)).plane is None
    finally:
        for k in ("VEXHIP_PLANE_TILE", "VEXHIP_PLANE_FORCE", "VEXHIP_PLANE_DEPTH"):
            os.environ.pop(k, None)


def test_march_needs_slices_that_repeat_in_runs(T, built_lib):
    """384^3: a grid line is 384 rows, a slice 512 -- the code blocks of consecutive slices cycle with period 3, every
    slice would decode anew, so the plan declines and the pair product runs (checked against an independent stencil
    evaluation; the strip order with the march product is covered at 512^3 below)."""
    torch, ops = T.torch, T.ops
    n = 384
    N = n ** 3
    dp, dc, dv = ops.poisson3d(n, T.dev)
    A = ops.SpMat(dp, dc, dv, direct=False)
    assert A.march is None and A.dictionary_blocks > 0 and not A.direct
    # round 4: the SELL-512 storage gets the grid plan (the matrix re-expressed by grid line from the slices' codes); the default
    # set-up stores the matrix by grid line straight from the CSR arrays -- same classes, same product
    assert A.grid is not None and A.grid["nx"] == 384 and A.grid["classes"] == 2, A.grid
    D = ops.SpMat(dp, dc, dv)
    assert D.direct and D.grid is not None and D.plane is None and D.dictionary_blocks == 0, (D.grid, D.plane)
    assert {k: v for k, v in D.grid.items() if k != "hot_class"} == {k: v for k, v in A.grid.items() if k != "hot_class"}, (D.grid, A.grid)
    x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=T.dev), 7)
    assert torch.equal(D @ x, A @ x) and torch.equal(A @ x, ops.SpMat(dp, dc, dv, march=False) @ x)
    del D
    ya = torch.full((N,), 3.0, dtype=torch.float64, device=T.dev)
    A.apply(x, ya, 0.5, True)
    h2i = float((n - 1) ** 2)
    X = x.view(n, n, n)
    ref = X.clone()
    c = X[1:-1, 1:-1, 1:-1]
    nb = (X[:-2, 1:-1, 1:-1] + X[2:, 1:-1, 1:-1] + X[1:-1, :-2, 1:-1] + X[1:-1, 2:, 1:-1] + X[1:-1, 1:-1, :-2] + X[1:-1, 1:-1, 2:])
    ref[1:-1, 1:-1, 1:-1] = h2i * (6 * c - nb)
    assert float((ya.view(n, n, n) - (3.0 + 0.5 * ref)).abs().max()) <= TOL * 12 * h2i


def test_row_pointers_64bit(T, oracle, built_lib):
    """64-bit row pointers through vexhip_spmat_create_*_p64 (round 3): every storage built from int64 pointers equals the
    one built from int32 pointers and the CSR oracle bit for bit (Poisson 48^3, the variable-coefficient operator, a matrix
    with a CSR tail, single precision)."""
    torch = T.torch
    n = 48
    N = n ** 3
    x = oracle.random_f64(3, N)
    for label, (ptr, col, val) in (("poisson", oracle.poisson3d(n)), ("diffusion", oracle.diffusion3d(n, 5))):
        want = oracle.spmv_csr(ptr, col, val, x)
        for fmt in ("auto", "sell8", "sell32", "csr"):
            A = T.ops.SpMat(T.up(ptr.astype(np.int64)), T.up(col), T.up(val), fmt=fmt)
            B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), fmt=fmt)
            assert A.handle and A.storage == B.storage, (label, fmt, A.storage, B.storage)
            y = torch.full((N,), 2.0, dtype=torch.float64, device=T.dev)
            A.apply(T.up(x), y, -0.5, True)
            assert np.array_equal(y.cpu().numpy(), 2.0 - 0.5 * want), (label, fmt)
    # rows wider than the ELL width keep a CSR tail (32-bit pointers inside the library)
    ptr, col, val = oracle.random_matrix(105, 700, 5000, 900)
    xr = oracle.random_f64(9, 5000)
    A = T.ops.SpMat(T.up(ptr.astype(np.int64)), T.up(col), T.up(val), n_cols=5000)
    y = torch.empty(700, dtype=torch.float64, device=T.dev)
    A.apply(T.up(xr), y)
    _check(y.cpu().numpy(), oracle.spmv_csr(ptr, col, val, xr), oracle.spmv_abs_bound(ptr, col, val, xr))
    ptr, col, val = oracle.poisson3d(n)
    F = T.ops.SpMat(T.up(ptr.astype(np.int64)), T.up(col), T.up(val.astype(np.float32)))
    G = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val.astype(np.float32)))
    yf = torch.empty(N, dtype=torch.float32, device=T.dev); yg = torch.empty_like(yf)
    F.apply(T.up(x.astype(np.float32)), yf); G.apply(T.up(x.astype(np.float32)), yg)
    assert torch.equal(yf, yg)
    # the one-pass build by grid line from 64-bit row pointers, both value types (forced: the grids are small)
    os.environ["VEXHIP_PLANE_FORCE"] = "1"
    try:
        for shape in ((96, 20, 21), (512, 6, 9), (125, 9, 12)):
            ptr, col, val = _grid7(*shape)
            m = len(ptr) - 1
            for dt in (np.float64, np.float32):
                v = val.astype(dt); xb = oracle.random_f64(11, m).astype(dt)
                A = T.ops.SpMat(T.up(ptr.astype(np.int64)), T.up(col), T.up(v))
                assert A.direct and A.grid is not None and (A.plane is not None) == (shape[0] == 512), (shape, dt.__name__, A.grid, A.plane)
                y = torch.empty(m, dtype=torch.float64 if dt == np.float64 else torch.float32, device=T.dev)
                A.apply(T.up(xb), y)
                assert np.array_equal(y.cpu().numpy(), oracle.spmv_csr(ptr, col, v, xb)), (shape, dt.__name__)
    finally:
        os.environ.pop("VEXHIP_PLANE_FORCE", None)


def test_csr_row_pointers_beyond_2_31_on_a_small_matrix(T, oracle, built_lib):
    """The staged CSR kernel with 64-bit row bounds (round 4: re-enabled).  Round 3 saw it fault on a small matrix whose row
    pointer VALUES cross 2^31 and take 120 s at 700^3; the cause was the fold of lanes WITHOUT a row in the ragged last
    workgroup ((int)(0 - tile_start) wraps to a large positive trip count once tile_start >= 2^31).  Here: 1000 and 1500 rows
    (not multiples of 256), pointers offset so that they run from below 2^31 to above it / start above 2^32 -- the column and
    value arrays are passed as base addresses that many entries BELOW the real arrays (the kernel only ever touches entries
    [ptr[0] & ~3, ptr[n])) -- bit for bit against the CSR restatement, '=' and '+= alpha', and the plain loop (variant 8)."""
    import ctypes
    from vexcl_amd import lib, _capi
    torch = T.torch
    L = lib()
    dev_index = T.dev.index or 0
    stream = ctypes.c_void_p(torch.cuda.current_stream(T.dev).cuda_stream)
    for n, m, off in ((1000, 3000, (1 << 31) - 2000), (1500, 1500, (1 << 32) + 4096), (777, 9000, (1 << 31) - 4)):
        ptr, col, val = oracle.random_matrix(200 + n, n, m, 12)
        x = oracle.random_f64(5, m); y0 = oracle.random_f64(6, n)
        want = oracle.spmv_csr(ptr, col, val, x)
        dptr = T.up(ptr.astype(np.int64) + off); dcol = T.up(col); dval = T.up(val); dx = T.up(x)
        assert dcol.data_ptr() % 16 == 0 and dval.data_ptr() % 16 == 0 and off % 4 == 0
        h = ctypes.c_void_p()
        L.spmat_create_f64_p64(dev_index, stream, n, ctypes.c_void_p(dptr.data_ptr()), ctypes.c_void_p(dcol.data_ptr() - 4 * off),
                               ctypes.c_void_p(dval.data_ptr() - 8 * off), _capi.SPMAT_CSR, _capi.SPMAT_BORROW_CSR, ctypes.byref(h))
        try:
            for variant in (-1, 8):
                L.spmv_csr_set_variant(variant)
                for alpha, append in ((1.0, False), (-0.75, True)):
                    y = T.up(y0.copy())
                    L.spmat_apply_f64(h, stream, alpha, int(append), ctypes.c_void_p(dx.data_ptr()), ctypes.c_void_p(y.data_ptr()))
                    torch.cuda.synchronize()
                    assert np.array_equal(y.cpu().numpy(), (y0 + alpha * want) if append else alpha * want), (n, off, variant, alpha)
        finally:
            L.spmv_csr_set_variant(-1)
            L.spmat_destroy(h)


def test_more_than_2_31_nonzeros(T, built_lib):
    """700^3 Poisson: 343 000 000 rows, 2 383 410 352 entries -- more than 2^31 on one device (the reference's default index
    type is size_t, vexcl/spmat.hpp:56-57).  Built in HBM with 64-bit row pointers, stored with diagonal and value codes,
    checked against an evaluation of the stencil that never touches the matrix; the CSR arrays themselves (row bounds
    read as 64-bit values) must give the same bits."""
    torch, ops = T.torch, T.ops
    n = 700
    N = n ** 3
    dp, dc, dv = ops.poisson3d(n, T.dev)
    assert dp.dtype == torch.int64 and int(dp[-1]) == 2383410352 and dc.numel() == 2383410352
    A = ops.SpMat(dp, dc, dv)
    assert A.storage == "sell8v" and A.info.nnz == 2383410352
    x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=T.dev), 11)
    y = torch.empty(N, dtype=torch.float64, device=T.dev)
    A.apply(x, y)
    C = ops.SpMat(dp, dc, dv, fmt="csr")
    yc = torch.empty_like(y)
    C.apply(x, yc)
    assert torch.equal(y, yc)
    del C, yc, dp, dc, dv
    A.ptr = A.col = A.val = None
    torch.cuda.empty_cache()
    h2i = float((n - 1) ** 2)
    X = x.view(n, n, n)
    ref = X.clone()
    c = X[1:-1, 1:-1, 1:-1]
    nb = (X[:-2, 1:-1, 1:-1] + X[2:, 1:-1, 1:-1] + X[1:-1, :-2, 1:-1] + X[1:-1, 2:, 1:-1] + X[1:-1, 1:-1, :-2] + X[1:-1, 1:-1, 2:])
    ref[1:-1, 1:-1, 1:-1] = h2i * (6 * c - nb)
    assert float((y.view(n, n, n) - ref).abs().max()) <= TOL * 12 * h2i


def test_poisson512_properties(T):
    """BASELINE.json's full size (134 217 728 rows, 930 123 728 nnz): no CPU
    oracle in seconds at this size, so size-independent properties:
    (1) CSR and HELL kernels agree bit for bit; (2) A*1 = indicator of the
    boundary (interior rows cancel exactly: 6h - 6h); (3) against an independent
    fp64 stencil evaluation written with torch slicing; (4) linearity."""
    torch, ops = T.torch, T.ops
    n = 512
    N = n ** 3
    dp, dc, dv = ops.poisson3d(n, T.dev)
    assert dc.numel() == 930123728 and int(dp[-1]) == 930123728
    A_csr = ops.SpMat(dp, dc, dv, fmt="csr")
    A_ell = ops.SpMat(dp, dc, dv, direct=False)         # the SELL-512 storage (slices, dictionary, march / plane plans)
    assert A_ell.fmt == "sell" and A_ell.hell.width == 7 and A_ell.hell.tail_nnz == 0 and not A_ell.direct
    A_hell = ops.SpMat(dp, dc, dv, fmt="hell")          # reference layout + L2-tiled traversal order
    assert A_hell.hell.order_grid >= N // 512 and A_ell.hell.order_grid >= N // 512

    x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=T.dev), 42)
    y1 = A_csr @ x
    y2 = A_ell @ x
    assert torch.equal(y1, y2)
    # the default product here is the march product on the XCD strip order (64 slices per strip, runs of 16); the pair product
    # of the same storage must give the same bits
    assert A_ell.march is not None and A_ell.march["far"] == [-262144, 262144] and A_ell.info.traversal.chunk % A_ell.march["run"] == 0
    # round 4: the default product is the plane product (512 lines per plane); the march product of the same storage must give
    # the same bits, also for '+= alpha'
    assert A_ell.plane is not None and A_ell.plane["lines_per_plane"] == 512 and A_ell.plane["planes"] == 512, A_ell.plane
    A_march = ops.SpMat(dp, dc, dv, plane=False)
    assert A_march.plane is None and A_march.march is not None and torch.equal(A_march @ x, y2)
    ya = y1.clone(); yb = y1.clone()
    A_ell.apply(x, ya, -0.5, True); A_march.apply(x, yb, -0.5, True)
    assert torch.equal(ya, yb)
    del A_march
    # round 4: the default set-up stores the matrix by grid line straight from the CSR arrays (two classes: interior lines,
    # boundary lines); the plane kernel reads those tables -- same bits, '=' and '+= alpha'
    A_line = ops.SpMat(dp, dc, dv)
    assert A_line.direct and A_line.storage == "sell8v" and A_line.dictionary_blocks == 0 and A_line.march is None
    assert A_line.grid is not None and A_line.grid["nx"] == 512 and A_line.grid["classes"] == 2, A_line.grid
    assert A_line.plane is not None and A_line.plane["table_pitch"] >= 514 and A_ell.plane["table_pitch"] == 0, (A_line.plane, A_ell.plane)
    assert A_line.plane["lines_per_plane"] == 512 and A_line.plane["planes"] == 512 and A_line.matrix_bytes() < (1 << 21)
    assert torch.equal(A_line @ x, y2)
    yc = y1.clone()
    A_line.apply(x, yc, -0.5, True)
    assert torch.equal(yc, ya)
    del A_line, ya, yb, yc
    A_pair = ops.SpMat(dp, dc, dv, march=False)
    assert A_pair.march is None and A_pair.plane is None and torch.equal(A_pair @ x, y2)
    del A_pair
    assert torch.equal(y1, A_hell @ x)
    y0 = torch.empty_like(y1)
    A_hell.hell.mul(x, y0, tiled=False)
    assert torch.equal(y1, y0)
    del A_hell, y0

    ones = torch.ones(N, dtype=torch.float64, device=T.dev)
    y = A_ell @ ones
    g = y.view(n, n, n)
    assert float(g[1:-1, 1:-1, 1:-1].abs().max()) == 0.0
    assert float(y.sum()) == float(N - (n - 2) ** 3)
    del ones, y, g

    # independent stencil evaluation
    h2i = float((n - 1) ** 2)
    X = x.view(n, n, n)
    ref = X.clone()
    c = X[1:-1, 1:-1, 1:-1]
    nb = (X[:-2, 1:-1, 1:-1] + X[2:, 1:-1, 1:-1] + X[1:-1, :-2, 1:-1] + X[1:-1, 2:, 1:-1]
          + X[1:-1, 1:-1, :-2] + X[1:-1, 1:-1, 2:])
    ref[1:-1, 1:-1, 1:-1] = h2i * (6 * c - nb)
    bound = 12 * h2i          # sum |a_ij x_j| <= 12 h2i for x in [0,1)
    assert float((y1.view(n, n, n) - ref).abs().max()) <= TOL * bound
    del ref, nb, c

    # linearity: A(2x) == 2 A x exactly (power-of-two scale)
    y3 = A_ell @ (x * 2)
    assert torch.equal(y3, y1 * 2)


def _structured(kind, rng):
    """Adversarial sparsity structures (beyond tests/random_matrix.hpp's 0..15 per row)."""
    if kind == "one_row":
        n, m = 1, 3000
        widths = np.array([2999])
    elif kind == "long_rows":                     # rows far longer than an LDS tile (2048) and than any ELL width
        n, m = 300, 20000
        widths = rng.integers(0, 4, size=n); widths[[7, 150, 299]] = [6000, 2048, 4097]
    elif kind == "power_law":
        n, m = 5000, 5000
        widths = np.minimum((rng.pareto(1.2, size=n) * 3).astype(np.int64), 3000)
    elif kind == "wide_regular":                  # ELL width 27 > 8: run-time width kernels
        n, m = 4099, 4099
        widths = np.full(n, 27)
    elif kind == "tile_edges":                    # block nnz exactly at / around the 2048-entry LDS tile
        n, m = 1024, 4096
        widths = np.full(n, 8); widths[255] = 9; widths[511] = 7
    elif kind == "all_empty":
        n, m = 777, 100
        widths = np.zeros(n, dtype=np.int64)
    else:                                         # "slice_tail": n just past a 512-row slice
        n, m = 513, 513
        widths = rng.integers(1, 8, size=n)
    widths = np.minimum(widths, m)
    ptr = np.concatenate([[0], np.cumsum(widths)]).astype(np.int32)
    col = np.concatenate([np.sort(rng.choice(m, size=int(w), replace=False)) for w in widths] or [np.zeros(0)]).astype(np.int32)
    val = rng.random(len(col)) - 0.5
    return ptr, col, val, m


@pytest.mark.parametrize("fmt", ["csr", "hell", "sell"])
@pytest.mark.parametrize("kind", ["one_row", "long_rows", "power_law", "wide_regular", "tile_edges", "all_empty", "slice_tail"])
def test_adversarial_structures(T, oracle, kind, fmt):
    import zlib
    rng = np.random.default_rng(zlib.crc32(kind.encode()))
    ptr, col, val, m = _structured(kind, rng)
    n = len(ptr) - 1
    x = rng.random(m) - 0.5
    y0 = rng.random(n)
    A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), n_cols=m, fmt=fmt)
    if kind == "all_empty":
        assert A.fmt == ("csr" if fmt == "sell" else fmt)       # no ELL part: SELL degrades to CSR
    want = y0.copy()
    oracle.spmv_csr(ptr, col, val, x, want, -0.75, True)
    y = T.up(y0.copy())
    A.apply(T.up(x), y, -0.75, True)
    assert np.array_equal(y.cpu().numpy(), want)                # same order, unfused: bit-exact
    y = T.up(y0.copy())
    A.apply(T.up(x), y, 2.0, False)
    assert np.array_equal(y.cpu().numpy(), oracle.spmv_csr(ptr, col, val, x, alpha=2.0))


def _banded(rng, n, offsets, density=0.8):
    """Random banded matrix: each row keeps each in-range diagonal with probability `density`."""
    rows, cols = [], []
    for off in offsets:
        i = np.arange(max(0, -off), min(n, n - off))
        keep = rng.random(len(i)) < density
        rows.append(i[keep]); cols.append(i[keep] + off)
    r = np.concatenate(rows); c = np.concatenate(cols)
    o = np.lexsort((c, r)); r, c = r[o], c[o]
    ptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n))]).astype(np.int32)
    return ptr, c.astype(np.int32), rng.random(len(c)) - 0.5


def test_sell8_diagonal_codes(T, oracle, built_lib):
    """<= 254 distinct diagonals => 1-byte diagonal codes (254 / 255 are the two padding codes); layout, pair
    alignment, fallback and bit-exact products."""
    rng = np.random.default_rng(8)
    # (a) Poisson: 7 diagonals {-n^2, -n, -1, 0, 1, n, n^2}
    n = 20
    ptr, col, val = oracle.poisson3d(n)
    S = T.ops.SlicedELL(T.up(ptr), T.up(col), T.up(val), value_codes=False)      # the values as they are (SELL8 proper)
    assert S.ndeltas == 7 and S.width == 7 and S.values is None
    assert S.deltas.cpu().numpy()[:7].tolist() == [-n * n, -n, -1, 0, 1, n, n * n]
    N = n ** 3
    raw = S.sell.cpu().numpy().reshape(-1, 4 * 1024 + 7 * 512 * 8)
    codes = raw[:, :4096].copy().view(np.uint32).reshape(-1, 4, 256)      # word [jp][t]
    vals = raw[:, 4096:].copy().view(np.float64).reshape(-1, 7, 512)
    table = S.deltas.cpu().numpy()
    for i in (0, 1, n * n + n + 1, 4321, N - 1):
        s, t, q = i // 512, (i % 512) // 2, i % 2
        got_cols, got_vals = [], []
        for j in range(7):
            code = (int(codes[s, j // 2, t]) >> (8 * ((j % 2) * 2 + q))) & 255
            if code < 254:
                got_cols.append(i + int(table[code])); got_vals.append(vals[s, j, 2 * t + q])
            else:
                assert vals[s, j, 2 * t + q] == 0.0
        assert got_cols == col[ptr[i]:ptr[i + 1]].tolist() and got_vals == val[ptr[i]:ptr[i + 1]].tolist()
    # pair alignment (oracle.sell_pair_slots): rows 2k, 2k+1 share an ELL column where they share a diagonal -- the
    # identity row of a grid boundary sits where its interior neighbour has its diagonal entry -- and the empty half is
    # code 255 (the partner's 16-byte load may cover it) or 254 (that load would leave x: first / last column)
    max_col = N - 1
    for i0 in (0, n * n + n, n * n + 2 * n - 2, 4320, N - 2):
        for q, slots in enumerate(oracle.sell_pair_slots(ptr, col, i0, 7, max_col)):
            s, t = i0 // 512, (i0 % 512) // 2
            for j, e in enumerate(slots):
                code = (int(codes[s, j // 2, t]) >> (8 * ((j % 2) * 2 + q))) & 255
                want = {"safe": 255, "unsafe": 254}[e] if isinstance(e, str) else [-n * n, -n, -1, 0, 1, n, n * n].index(int(col[e]) - (i0 + q))
                assert code == want, (i0, q, j, code, want)
    assert oracle.sell_pair_slots(ptr, col, n * n + n, 7, max_col)[0] == ["safe"] * 3 + [int(ptr[n * n + n])] + ["safe"] * 3
    # (b) banded matrices with gaps, odd sizes, more diagonals than the unrolled widths, a CSR tail
    for nn, offs in ((5000, [-700, -3, -1, 0, 2, 9, 1234]), (1537, list(range(-20, 21, 2))), (3000, [0]),
                     (2048, [-1024, -512, -64, -8, -1, 0, 1, 8, 64, 512, 1024, 1500])):
        ptr, col, val = _banded(rng, nn, offs)
        x = rng.random(nn) - 0.5
        y0 = rng.random(nn)
        S = T.ops.SlicedELL(T.up(ptr), T.up(col), T.up(val))
        assert 1 <= S.ndeltas <= len(offs)
        want = y0.copy(); oracle.spmv_csr(ptr, col, val, x, want, 1.25, True)
        y = T.up(y0.copy()); S.mul(T.up(x), y, 1.25, True)
        assert np.array_equal(y.cpu().numpy(), want)
        S32 = T.ops.SlicedELL(T.up(ptr), T.up(col), T.up(val), codes=False)      # 32-bit columns: same bits
        y2 = T.up(y0.copy()); S32.mul(T.up(x), y2, 1.25, True)
        assert T.torch.equal(y, y2)
    # (b2) up to 254 diagonals are coded, 255 are not
    for nd, coded in ((254, True), (255, False)):
        offs = list(range(-(nd // 2), nd - nd // 2))
        ptr, col, val = _banded(rng, 3000, offs, density=1.0)
        S = T.ops.SlicedELL(T.up(ptr), T.up(col), T.up(val))
        assert (S.ndeltas == nd) if coded else (S.ndeltas == -1)
        x = rng.random(3000)
        y = T.torch.empty(3000, dtype=T.torch.float64, device=T.dev)
        S.mul(T.up(x), y)
        assert np.array_equal(y.cpu().numpy(), oracle.spmv_csr(ptr, col, val, x))
    # (c) not banded: more than 254 diagonals -> 32-bit columns are kept
    ptr, col, val = oracle.random_matrix(9, 4000, 4000, 16)
    S = T.ops.SlicedELL(T.up(ptr), T.up(col), T.up(val))
    assert S.ndeltas == -1 and S.deltas is None
    x = rng.random(4000)
    y = T.torch.empty(4000, dtype=T.torch.float64, device=T.dev)
    S.mul(T.up(x), y)
    assert np.array_equal(y.cpu().numpy(), oracle.spmv_csr(ptr, col, val, x))
    # float matrices
    ptr, col, val = _banded(rng, 3001, [-5, 0, 5, 77])
    v32, x32 = val.astype(np.float32), (rng.random(3001) - 0.5).astype(np.float32)
    S = T.ops.SlicedELL(T.up(ptr), T.up(col), T.up(v32))
    assert S.ndeltas == 4
    y = T.torch.empty(3001, dtype=T.torch.float32, device=T.dev)
    S.mul(T.up(x32), y)
    assert np.array_equal(y.cpu().numpy(), oracle.spmv_csr(ptr, col, v32, x32))


@pytest.mark.parametrize("nrhs", [1, 2, 3, 4, 5, 7])
def test_multi_rhs_products_are_bit_identical(T, oracle, built_lib, nrhs):
    """`SpMat * multivector` (spmat.hpp:388-398): one pass over the matrix for up to four
    right-hand sides; every result equals the single-vector product bit for bit."""
    rng = np.random.default_rng(100 + nrhs)
    cases = []
    n = 24
    cases.append(oracle.poisson3d(n))                                      # SELL8, unrolled width 7
    cases.append(_banded(rng, 5000, [-700, -3, -1, 0, 2, 9]))              # SELL8, run-time width
    cases.append(oracle.random_matrix(11, 3000, 3000, 16))                 # 32-bit columns + CSR tail
    p, c, v = oracle.random_matrix(12, 2000, 2000, 6)
    cases.append((p, c, v))
    for ptr, col, val in cases:
        rows, cols = len(ptr) - 1, len(ptr) - 1
        A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val))
        xs = [T.up(rng.random(cols) - 0.5) for _ in range(nrhs)]
        y0 = [rng.random(rows) for _ in range(nrhs)]
        for alpha, append in ((1.0, False), (-2.5, True)):
            want = [T.up(y.copy()) for y in y0]
            for x, y in zip(xs, want):
                A.apply(x, y, alpha, append)
            got = [T.up(y.copy()) for y in y0]
            A.apply_multi(xs, got, alpha, append)
            for g, w in zip(got, want):
                assert T.torch.equal(g, w)
        # and against the oracle for the first component
        y = oracle.spmv_csr(ptr, col, val, xs[0].cpu().numpy())
        out = [T.torch.empty(rows, dtype=T.torch.float64, device=T.dev) for _ in range(nrhs)]
        A.apply_multi(xs, out)
        assert np.array_equal(out[0].cpu().numpy(), y)
    # float32, CSR-only format falls back to one product per component
    ptr, col, val = _banded(rng, 3001, [-5, 0, 5, 77])
    v32 = val.astype(np.float32)
    A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32))
    xs = [T.up((rng.random(3001) - 0.5).astype(np.float32)) for _ in range(nrhs)]
    out = [T.torch.empty(3001, dtype=T.torch.float32, device=T.dev) for _ in range(nrhs)]
    A.apply_multi(xs, out)
    for x, o in zip(xs, out):
        assert np.array_equal(o.cpu().numpy(), oracle.spmv_csr(ptr, col, v32, x.cpu().numpy()))
    C = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), fmt="csr")
    out2 = [T.torch.empty(3001, dtype=T.torch.float32, device=T.dev) for _ in range(nrhs)]
    C.apply_multi(xs, out2)
    for a, b in zip(out, out2):
        assert T.torch.equal(a, b)


def test_sell8v_value_codes(T, oracle, built_lib):
    """<= 255 distinct values in the ELL part => 1-byte value codes next to the diagonal codes; layout,
    fallbacks, bit-exact products (the same arithmetic in the same order as every other format)."""
    rng = np.random.default_rng(21)
    # (a) Poisson: 7 diagonals, 3 values
    n = 20
    ptr, col, val = oracle.poisson3d(n)
    N = n ** 3
    S = T.ops.SlicedELL(T.up(ptr), T.up(col), T.up(val))
    assert S.ndeltas == 7 and S.nvalues == 3 and S.width == 7
    vt = S.values.cpu().numpy()[:3]
    assert sorted(vt.tolist()) == sorted(set(val.tolist()))
    raw = S.sell.cpu().numpy().reshape(-1, 2 * 4 * 1024)                   # per slice: 4 KiB diagonal codes, 4 KiB value codes
    ccodes = raw[:, :4096].copy().view(np.uint32).reshape(-1, 4, 256)
    vcodes = raw[:, 4096:].copy().view(np.uint32).reshape(-1, 4, 256)
    dt = S.deltas.cpu().numpy()
    for i in (0, 1, n * n + n + 1, 4321, N - 1):
        s, t, q = i // 512, (i % 512) // 2, i % 2
        cols, vals = [], []
        for j in range(7):
            sh = 8 * ((j % 2) * 2 + q)
            code = (int(ccodes[s, j // 2, t]) >> sh) & 255
            if code < 254:                        # 254 / 255: padding
                cols.append(i + int(dt[code])); vals.append(float(S.values[(int(vcodes[s, j // 2, t]) >> sh) & 255]))
        assert cols == col[ptr[i]:ptr[i + 1]].tolist() and vals == val[ptr[i]:ptr[i + 1]].tolist()
    x = rng.random(N) - 0.5
    y0 = rng.random(N)
    want = y0.copy(); oracle.spmv_csr(ptr, col, val, x, want, -0.75, True)
    y = T.up(y0.copy()); S.mul(T.up(x), y, -0.75, True)
    assert np.array_equal(y.cpu().numpy(), want)
    # (b) banded matrices with few distinct values, signed zeros, odd sizes, widths beyond the unrolled ones, a CSR tail
    for nn, offs, nvals in ((5000, [-700, -3, -1, 0, 2, 9, 1234], 5), (1537, list(range(-20, 21, 2)), 200), (2048, [-1024, -8, 0, 8, 1024, 1500], 1)):
        ptr, col, val = _banded(rng, nn, offs)
        pool = np.concatenate([rng.random(nvals) - 0.5, [0.0, -0.0]])[:max(nvals, 1)]
        val = pool[rng.integers(0, len(pool), size=len(val))]
        x = rng.random(nn) - 0.5
        S = T.ops.SlicedELL(T.up(ptr), T.up(col), T.up(val))
        assert 1 <= S.nvalues <= len(pool)
        S8 = T.ops.SlicedELL(T.up(ptr), T.up(col), T.up(val), value_codes=False)
        assert S8.values is None and S8.deltas is not None
        ya = T.torch.empty(nn, dtype=T.torch.float64, device=T.dev); yb = T.torch.empty_like(ya)
        S.mul(T.up(x), ya, 1.5, False); S8.mul(T.up(x), yb, 1.5, False)
        assert T.torch.equal(ya, yb)
        assert np.array_equal(ya.cpu().numpy(), 1.5 * oracle.spmv_csr(ptr, col, val, x))
    # (c) more than 255 distinct values: diagonal codes only
    ptr, col, val = _banded(rng, 4000, [-5, 0, 5, 77])
    S = T.ops.SlicedELL(T.up(ptr), T.up(col), T.up(val))
    assert S.nvalues == -1 and S.values is None and S.ndeltas == 4
    # float matrices, and the SpMat front end
    ptr, col, val = _banded(rng, 3001, [-5, 0, 5, 77])
    v32 = np.float32(rng.integers(1, 9, size=len(val)) * 0.125)
    x32 = (rng.random(3001) - 0.5).astype(np.float32)
    A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32))
    assert A.hell.nvalues == 8
    y = T.torch.empty(3001, dtype=T.torch.float32, device=T.dev)
    A.apply(T.up(x32), y)
    assert np.array_equal(y.cpu().numpy(), oracle.spmv_csr(ptr, col, v32, x32))
    for fmt, has_codes, has_values in (("sell8", True, False), ("sell32", False, False), ("sell", True, True)):
        B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), fmt=fmt)
        assert (B.hell.deltas is not None) == has_codes and (B.hell.values is not None) == has_values
        y2 = T.torch.empty_like(y); B.apply(T.up(x32), y2)
        assert T.torch.equal(y, y2)


def test_randomized_structures_all_storages(T, oracle, built_lib):
    """Fuzz: random sizes, diagonal sets, value pools, densities, alpha, SET / += -- every storage the matrix allows
    (value codes, diagonal codes, 32-bit columns, reference ELL layout, CSR) must equal the oracle bit for bit."""
    rng = np.random.default_rng(2024)
    seen = {"sell8v": 0, "sell8": 0, "sell32": 0}
    for trial in range(40):
        n = int(rng.integers(1, 6000))
        nd = int(rng.integers(1, 12))
        offs = sorted(set(int(o) for o in rng.integers(-min(n, 2000), min(n, 2000) + 1, size=nd)) | {0})
        ptr, col, val = _banded(rng, n, offs, density=float(rng.uniform(0.3, 1.0)))
        if len(col) == 0:
            continue
        kind = trial % 3
        if kind == 0:                                    # few distinct values
            pool = rng.standard_normal(int(rng.integers(1, 40)))
            val = pool[rng.integers(0, len(pool), size=len(val))]
        elif kind == 1:                                  # a few rows much wider than the rest (CSR tail)
            extra_rows = rng.integers(0, n, size=max(1, n // 50))
            r = np.repeat(np.arange(n), np.diff(ptr)); c = col.astype(np.int64)
            er = np.repeat(extra_rows, 20); ec = rng.integers(0, n, size=len(er))
            r = np.concatenate([r, er]); c = np.concatenate([c, ec])
            key = np.unique(r * n + c); r, c = key // n, key % n
            ptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n))]).astype(np.int32)
            col = c.astype(np.int32); val = rng.standard_normal(len(col))
        x = rng.standard_normal(n)
        y0 = rng.standard_normal(n)
        alpha = float(rng.choice([1.0, -1.0, 0.5, 3.25]))
        append = bool(rng.integers(0, 2))
        want = y0.copy() if append else np.zeros(n)
        oracle.spmv_csr(ptr, col, val, x, want, alpha, append)
        for fmt in ("sell", "sell8", "sell32", "hell", "csr"):
            try:
                A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), fmt=fmt)
            except T.ops.Error:
                continue
            if fmt == "sell" and A.hell is not None:
                seen["sell8v" if A.hell.values is not None else ("sell8" if A.hell.deltas is not None else "sell32")] += 1
            y = T.up(y0.copy())
            A.apply(T.up(x), y, alpha, append)
            assert np.array_equal(y.cpu().numpy(), want), (trial, fmt, n, offs[:5], alpha, append)
    assert seen["sell8v"] >= 5 and seen["sell8"] >= 5, seen


def test_slice_dictionary_function(T):
    """vexhip_slice_dictionary: numbering in order of first appearance, pool = the representatives, -1 past the capacity."""
    import ctypes
    torch, L = T.torch, T.L
    rng = np.random.default_rng(5)
    # (slices, words compared per slice, words between slices, distinct patterns, capacity)
    for ns, words, stride, kinds, cap in ((1, 8, 8, 1, 4), (200, 2048, 2048, 3, 8), (777, 96, 96, 8, 8), (300, 64, 64, 9, 8),
                                          (64, 512, 512, 64, 128), (500, 256, 2048, 5, 8)):
        patterns = rng.integers(0, 2 ** 32, size=(kinds, words), dtype=np.uint64).astype(np.uint32)
        if kinds > 1:
            patterns[1] = patterns[0]; patterns[1, -1] ^= 1                       # two patterns that differ in ONE bit of the last word
        which = rng.integers(0, kinds, size=ns); which[:min(ns, kinds)] = np.arange(min(ns, kinds))
        full = rng.integers(0, 2 ** 32, size=(ns, stride), dtype=np.uint64).astype(np.uint32)   # what follows the compared part differs everywhere
        full[:, :words] = patterns[which]
        buf = T.up(full.view(np.int32))
        blocks = torch.full((ns,), -7, dtype=torch.int32, device=T.dev)
        pool = torch.zeros((cap, words), dtype=torch.int32, device=T.dev)
        nb = ctypes.c_int64(-5)
        L.slice_dictionary(0, None, ns, stride * 4, words * 4, ctypes.c_void_p(buf.data_ptr()), cap, ctypes.c_void_p(blocks.data_ptr()),
                           ctypes.c_void_p(pool.data_ptr()), ctypes.byref(nb))
        distinct = len(np.unique(which))
        if distinct > cap:
            assert nb.value == -1
            continue
        assert nb.value == distinct
        first = {}
        want = np.array([first.setdefault(int(k), len(first)) for k in which], dtype=np.int32)
        assert np.array_equal(blocks.cpu().numpy(), want)
        reps = [int(np.nonzero(want == b)[0][0]) for b in range(distinct)]
        assert np.array_equal(pool[:distinct].cpu().numpy().view(np.uint32), patterns[which[reps]])


@pytest.mark.parametrize("n", [40, 64])
def test_spmat_slice_dictionary_is_bit_identical(T, oracle, built_lib, n):
    """The value-coded Poisson matrix repeats its slices: vex::SpMat stores the distinct ones once (vexhip_spmat_info.
    dictionary_blocks) and every product -- single, multi-vector, '=' and '+=' -- equals the streamed layout's and the CSR
    oracle's bit for bit.  Matrices whose slices differ, and small ones, keep one block per slice."""
    torch = T.torch
    ptr, col, val = oracle.poisson3d(n)
    N = n ** 3
    A = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val))
    B = T.ops.SpMat(T.up(ptr), T.up(col), T.up(val), dictionary=False)
    assert A.storage == "sell8v" and B.storage == "sell8v" and B.dictionary_blocks == 0
    ns = (N + 511) // 512
    if n == 64:          # 8 grid lines per slice, 8 slices per plane: a handful of distinct slices out of 512
        assert 1 <= A.dictionary_blocks <= 16 and A.matrix_bytes() < B.matrix_bytes() // 8
    if A.dictionary_blocks:      # n = 40: slices and grid lines do not line up; whatever the count, the products must agree
        assert A.dictionary_blocks <= 128 and A.matrix_bytes() == A.dictionary_blocks * (B.matrix_bytes() // ns) + 4 * ns
    x = oracle.random_f64(3, N); y0 = oracle.random_f64(4, N)
    want = oracle.spmv_csr(ptr, col, val, x)
    for alpha, append in ((1.0, False), (-0.75, True)):
        ya, yb = T.up(y0.copy()), T.up(y0.copy())
        A.apply(T.up(x), ya, alpha, append); B.apply(T.up(x), yb, alpha, append)
        assert torch.equal(ya, yb)
        ref = (y0 + alpha * want) if append else alpha * want
        assert np.array_equal(ya.cpu().numpy(), ref)
    xs = [T.up(oracle.random_f64(10 + k, N)) for k in range(3)]
    ya = [torch.empty(N, dtype=torch.float64, device=T.dev) for _ in range(3)]; yb = [torch.empty_like(v) for v in ya]
    A.apply_multi(xs, ya); B.apply_multi(xs, yb)
    for k in range(3):
        assert torch.equal(ya[k], yb[k])
        assert np.array_equal(ya[k].cpu().numpy(), oracle.spmv_csr(ptr, col, val, xs[k].cpu().numpy()))
    # the per-entry kernels (A/B switch) read through the dictionary too
    T.L.spmv_sell8_set_variant(1)
    try:
        yc = torch.empty(N, dtype=torch.float64, device=T.dev); A.apply(T.up(x), yc)
        assert np.array_equal(yc.cpu().numpy(), want)
    finally:
        T.L.spmv_sell8_set_variant(0)
    # stored values (variable coefficients): the CODE part of the slices is pooled, the values stay in the slices
    p2, c2, v2 = oracle.diffusion3d(n, 7)
    C = T.ops.SpMat(T.up(p2), T.up(c2), T.up(v2)); D = T.ops.SpMat(T.up(p2), T.up(c2), T.up(v2), dictionary=False)
    assert C.storage == "sell8" and D.storage == "sell8" and D.dictionary_blocks == 0
    if n == 64:
        assert 1 <= C.dictionary_blocks <= 16 and C.matrix_bytes() < D.matrix_bytes()
    want2 = oracle.spmv_csr(p2, c2, v2, x)
    for alpha, append in ((1.0, False), (0.5, True)):
        ya, yb = T.up(y0.copy()), T.up(y0.copy())
        C.apply(T.up(x), ya, alpha, append); D.apply(T.up(x), yb, alpha, append)
        assert torch.equal(ya, yb)
        assert np.array_equal(ya.cpu().numpy(), (y0 + alpha * want2) if append else alpha * want2)
    C.apply_multi(xs, ya_ := [torch.empty(N, dtype=torch.float64, device=T.dev) for _ in range(3)])
    for k in range(3):
        assert np.array_equal(ya_[k].cpu().numpy(), oracle.spmv_csr(p2, c2, v2, xs[k].cpu().numpy()))
    T.L.spmv_sell8_set_variant(1)
    try:
        yc = torch.empty(N, dtype=torch.float64, device=T.dev); C.apply(T.up(x), yc)
        assert np.array_equal(yc.cpu().numpy(), want2)
    finally:
        T.L.spmv_sell8_set_variant(0)
    # single precision, both storages, and a matrix with a CSR tail (a few rows wider than the ELL width)
    if n == 64:
        v32 = val.astype(np.float32); x32 = x.astype(np.float32)
        F = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32)); Fn = T.ops.SpMat(T.up(ptr), T.up(col), T.up(v32), dictionary=False)
        assert F.storage == "sell8v" and F.dictionary_blocks > 0
        yf = torch.empty(N, dtype=torch.float32, device=T.dev); yg = torch.empty_like(yf)
        F.apply(T.up(x32), yf); Fn.apply(T.up(x32), yg)
        assert torch.equal(yf, yg) and np.array_equal(yf.cpu().numpy(), oracle.spmv_csr(ptr, col, v32, x32))
        w32 = v2.astype(np.float32)
        Gm = T.ops.SpMat(T.up(p2), T.up(c2), T.up(w32))
        assert Gm.storage == "sell8" and Gm.dictionary_blocks > 0
        Gm.apply(T.up(x32), yf)
        assert np.array_equal(yf.cpu().numpy(), oracle.spmv_csr(p2, c2, w32, x32))
        # tail: 40 rows get 3 extra entries each -> hybrid ELL keeps width 7 and a CSR tail; the ELL slices still repeat
        rng = np.random.default_rng(9)
        rows_extra = np.sort(rng.choice(np.arange(N // 4, N // 2), size=40, replace=False))
        cnt = np.diff(ptr).astype(np.int64); add = np.zeros(N, dtype=np.int64); add[rows_extra] = 3
        ptr_t = np.concatenate([[0], np.cumsum(cnt + add)]).astype(np.int32)
        col_t = np.empty(ptr_t[-1], dtype=np.int32); val_t = np.empty(ptr_t[-1], dtype=np.float64)
        for i in range(N):
            b, e = ptr[i], ptr[i + 1]; bt = ptr_t[i]
            col_t[bt:bt + e - b] = col[b:e]; val_t[bt:bt + e - b] = val[b:e]
        for i in rows_extra:
            bt = ptr_t[i] + cnt[i]
            col_t[bt:bt + 3] = [i - 7 * n, i + 5, i + 9 * n]; val_t[bt:bt + 3] = val[ptr[i]]      # values from the table: still value-coded
        Tm = T.ops.SpMat(T.up(ptr_t), T.up(col_t), T.up(val_t)); Tn = T.ops.SpMat(T.up(ptr_t), T.up(col_t), T.up(val_t), dictionary=False)
        assert Tm.hell.tail_nnz > 0 and Tm.storage == Tn.storage
        yt = torch.empty(N, dtype=torch.float64, device=T.dev); yu = torch.empty_like(yt)
        Tm.apply(T.up(x), yt); Tn.apply(T.up(x), yu)
        assert torch.equal(yt, yu) and np.array_equal(yt.cpu().numpy(), oracle.spmv_csr(ptr_t, col_t, val_t, x))
    # fewer than 64 slices: not tried
    p3, c3, v3 = oracle.poisson3d(16)
    assert T.ops.SpMat(T.up(p3), T.up(c3), T.up(v3)).dictionary_blocks == 0


def test_ccsr_to_csr_expansion(T):
    """vexhip_ccsr_to_csr_*: row i of the CSR matrix = table row idx[i] with absolute columns, entries in table order; a
    column outside [0, n) is reported."""
    import ctypes
    torch, L = T.torch, T.L
    rng = np.random.default_rng(11)
    n = 5000
    row = np.array([0, 1, 4, 4, 9], dtype=np.uint32)                     # 4 unique rows, one of them empty
    col = np.array([0, -1, 0, 1, -3, -1, 0, 2, 3], dtype=np.int32)
    val = rng.random(9)
    idx = rng.integers(0, 4, size=n).astype(np.uint32)
    idx[:3] = 0; idx[-3:] = 0                                            # rows whose offsets would leave [0, n) use the 1-entry row
    d = lambda a: T.up(a.view(np.int32) if a.dtype == np.uint32 else a)
    didx, drow, dcol, dval = d(idx), d(row), d(col), d(val)
    ptr = torch.empty(n + 1, dtype=torch.int32, device=T.dev)
    nnz = ctypes.c_int64(-1)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    L.ccsr_to_csr_f64_i32(0, None, n, p(didx), p(drow), p(dcol), p(dval), p(ptr), None, None, ctypes.byref(nnz))
    lens = (row[idx + 1] - row[idx]).astype(np.int64)
    want_ptr = np.concatenate([[0], np.cumsum(lens)])
    assert nnz.value == want_ptr[-1] and np.array_equal(ptr.cpu().numpy(), want_ptr)
    oc = torch.empty(nnz.value, dtype=torch.int32, device=T.dev); ov = torch.empty(nnz.value, dtype=torch.float64, device=T.dev)
    bad = ctypes.c_int64(-5)
    L.ccsr_to_csr_f64_i32(0, None, n, p(didx), p(drow), p(dcol), p(dval), p(ptr), p(oc), p(ov), ctypes.byref(bad))
    assert bad.value == 0
    wc = np.concatenate([np.arange(n)[i] + col[row[idx[i]]:row[idx[i] + 1]] for i in range(n)])
    wv = np.concatenate([val[row[idx[i]]:row[idx[i] + 1]] for i in range(n)])
    assert np.array_equal(oc.cpu().numpy(), wc) and np.array_equal(ov.cpu().numpy(), wv)
    idx[0] = 3                                                           # row 0 with offset -3: column -3
    L.ccsr_to_csr_f64_i32(0, None, n, p(d(idx)), p(drow), p(dcol), p(dval), p(ptr), None, None, ctypes.byref(nnz))
    oc = torch.empty(nnz.value, dtype=torch.int32, device=T.dev); ov = torch.empty(nnz.value, dtype=torch.float64, device=T.dev)
    L.ccsr_to_csr_f64_i32(0, None, n, p(d(idx)), p(drow), p(dcol), p(dval), p(ptr), p(oc), p(ov), ctypes.byref(bad))
    assert bad.value == -1
