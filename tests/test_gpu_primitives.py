"""GPU parity tests for Reductor / scan / sort through the C ABI.
Reference tests mirrored: tests/vector_arithmetics.cpp:66-99 (reductions),
tests/scan.cpp:9-41, tests/sort.cpp:9-45.  Integers and orderings bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden.npz"))


@pytest.fixture(scope="module")
def T():
    import torch
    from vexcl_amd import ops

    class NS:
        pass
    ns = NS()
    ns.torch, ns.ops, ns.dev = torch, ops, torch.device("cuda:0")
    ns.up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ns.dev)
    return ns


def _u32(T, a):          # torch has no arithmetic on uint32: carry bits in int32
    return T.up(a.view(np.int32))


@pytest.mark.parametrize("n", [1, 63, 1000, 1 << 20, (1 << 22) + 3])
def test_reductor_sum_kahan_minmax(T, oracle, n):
    # vector_arithmetics.cpp:72-96: values (U-0.5)*1e8; SUM to 1e-8 %, MIN/MAX exact
    x = (oracle.random_f64(n, n) - 0.5) * 1e8
    d = T.up(x)
    exact = oracle.sum_kahan(x)
    scale = np.abs(x).sum()
    assert abs(T.ops.Reductor("SUM")(d) - exact) <= 1e-10 * scale
    assert abs(T.ops.Reductor("SUM_Kahan")(d) - exact) <= 1e-13 * scale
    assert T.ops.Reductor("MIN")(d) == x.min()
    assert T.ops.Reductor("MAX")(d) == x.max()
    assert T.ops.Reductor("MIN_MAX")(d) == (x.min(), x.max())


def test_reductor_golden_and_types(T, oracle):
    v = G["sum_x"]
    assert abs(T.ops.Reductor("SUM")(T.up(v)) - float(G["sum_exact"])) <= 1e-10 * np.abs(v).sum()
    i = oracle.random_i32(3, 100001, -1000, 1000)
    assert T.ops.Reductor("SUM")(T.up(i)) == int(i.sum(dtype=np.int32))
    assert T.ops.Reductor("MAX")(T.up(i)) == i.max() and T.ops.Reductor("MIN")(T.up(i)) == i.min()
    f = oracle.random_f64(4, 77777).astype(np.float32)
    assert abs(T.ops.Reductor("SUM")(T.up(f)) - float(f.astype(np.float64).sum())) <= 1e-4 * 77777
    u = oracle.random_u32(5, 12345)
    assert T.ops.Reductor("MAX")(_u32(T, u), unsigned=True) & 0xffffffff == int(u.max())
    l = oracle.random_i32(6, 5000, -50, 50).astype(np.int64) * (1 << 33)
    assert T.ops.Reductor("SUM")(T.up(l)) == int(l.sum())


def test_reductor_dot_and_empty(T, oracle):
    # examples/benchmark.cpp:224-246: sum(a*b), N = 2^24
    n = 1 << 24
    a, b = oracle.random_f64(1, n), oracle.random_f64(2, n)
    got = T.ops.Reductor("SUM").dot(T.up(a), T.up(b))
    assert abs(got - oracle.dot_kahan(a, b)) <= 1e-10 * float(np.dot(a, b))
    e = T.torch.empty(0, dtype=T.torch.float64, device=T.dev)
    assert T.ops.Reductor("SUM")(e) == 0.0
    assert T.ops.Reductor("MAX")(e) == np.finfo(np.float64).min       # reductor.hpp:86-88 initial()


@pytest.mark.parametrize("n", [1, 2, 255, 4096, 4097, 1 << 20, (1 << 24) + 11])
def test_scan_u32_exact(T, oracle, n):
    # scan.cpp:9-24: in-place inclusive on ints, exact vs std::partial_sum
    x = oracle.random_u32(n, n)
    d = _u32(T, x)
    T.ops.inclusive_scan(d, d, unsigned=True)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), oracle.inclusive_scan(x))
    d = _u32(T, x)
    out = T.ops.exclusive_scan(d, None, init=7, unsigned=True)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), oracle.exclusive_scan(x, 7))


def test_scan_golden_and_types(T, oracle):
    out = T.ops.inclusive_scan(_u32(T, G["scan_in"]), unsigned=True)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), G["scan_inclusive"])
    i = oracle.random_i32(1, 100003, 0, 100)
    assert np.array_equal(T.ops.inclusive_scan(T.up(i)).cpu().numpy(), np.cumsum(i, dtype=np.int32))
    l = oracle.random_i32(2, 50001, -5, 100).astype(np.int64) * 100000
    assert np.array_equal(T.ops.exclusive_scan(T.up(l), init=-3).cpu().numpy(), oracle.exclusive_scan(l, -3))
    # scan.cpp:26-41: exclusive on doubles to 1e-8 %
    x = oracle.random_f64(3, 1 << 20)
    got = T.ops.exclusive_scan(T.up(x)).cpu().numpy()
    want = oracle.exclusive_scan(x, 0.0)
    assert np.all(np.abs(got - want) <= 1e-10 * np.maximum(want, 1.0))
    f = x[:10000].astype(np.float32)
    got = T.ops.inclusive_scan(T.up(f)).cpu().numpy()
    assert np.allclose(got, np.cumsum(f.astype(np.float64)), rtol=1e-5)


@pytest.mark.parametrize("n", [2, 100, 4096, 4097, 12288, 12289, 3 * 12288 - 1, 1 << 20, (1 << 22) + 5])
def test_sort_u32_exact(T, oracle, n):
    x = oracle.random_u32(n + 1, n)
    d = _u32(T, x)
    T.ops.sort(d, unsigned=True)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), np.sort(x))


def test_sort_floats_sorted(T, oracle):
    # sort.cpp:9-20: 1M floats, is_sorted
    x = (oracle.random_f64(1, 1 << 20) - 0.5).astype(np.float32)
    x[:3] = [0.0, -0.0, np.float32(-1e30)]
    d = T.up(x)
    T.ops.sort(d)
    got = d.cpu().numpy()
    assert np.array_equal(got, np.sort(x, kind="stable")) or np.all(np.diff(got) >= 0)
    assert np.array_equal(np.sort(got), np.sort(x))
    dd = (oracle.random_f64(2, 100001) - 0.5) * 1e6
    d = T.up(dd); T.ops.sort(d)
    assert np.array_equal(d.cpu().numpy(), np.sort(dd))
    d = T.up(dd); T.ops.sort(d, descending=True)
    assert np.array_equal(d.cpu().numpy(), np.sort(dd)[::-1])


@pytest.mark.parametrize("n", [5000, 1 << 20])
def test_sort_by_key_is_stable(T, oracle, n):
    # sort.cpp:22-45: int keys U[0,100], float values; must equal std::stable_sort's permutation
    k = oracle.random_i32(1, n, 0, 100)
    v = oracle.random_f64(2, n).astype(np.float32)
    dk, dv = T.up(k), T.up(v)
    T.ops.sort_by_key(dk, dv)
    wk, wv = oracle.stable_sort_by_key(k, v)
    assert np.array_equal(dk.cpu().numpy(), wk) and np.array_equal(dv.cpu().numpy(), wv)
    # 8-byte payload = original position: the permutation itself
    idx = np.arange(n, dtype=np.int64)
    dk, di = T.up(k), T.up(idx)
    T.ops.sort_by_key(dk, di)
    assert np.array_equal(di.cpu().numpy(), np.argsort(k, kind="stable"))


@pytest.mark.parametrize("mode", [0, 6, 7])
def test_sort_rank_schemes_give_the_stable_permutation(T, oracle, mode):
    """Every way the scatter ranks a tile -- match words (0: ordered by construction); the default (6): half-wave units whose
    64-bit counter words return the rank AND the lanes served before, one returning atomic per key; and (7) the path the default
    takes when a lane WAS served out of order: the tile ranked a second time by ballots, forced here for every tile (round 6: until
    then the kernel trapped) -- must produce std::stable_sort's permutation (sort.cpp:22-45), for few and for many distinct keys,
    keys whose upper digits are constant (tiles copied as blocks), ragged sizes, 4- and 8-byte keys and payloads; the status of
    the sort reports the re-ranked tiles (none by default: this part serves the lanes of an LDS atomic in lane order) and no
    dropped tile."""
    from vexcl_amd import lib
    L = lib()
    L.sort_set_rank(mode)
    try:
        for n, hi in ((12288 * 3 + 17, 3), (1 << 20, 100), (777777, 1 << 30), (1 << 22, 255), (12288 * 40 + 5, 0), (12288 * 9, 70000)):
            k = oracle.random_i32(n, n, 0, hi)
            idx = np.arange(n, dtype=np.int64)
            dk, di = T.up(k), T.up(idx)
            T.ops.sort_by_key(dk, di)
            assert np.array_equal(di.cpu().numpy(), np.argsort(k, kind="stable")), (mode, n, hi)
            assert np.array_equal(dk.cpu().numpy(), np.sort(k, kind="stable"))
            reranked, dropped = T.ops.sort_status()
            assert dropped == 0 and (reranked > 0 if (mode == 7 and hi > 0 and n >= 4096) else reranked == 0), (mode, n, hi, reranked, dropped)
        l = oracle.random_i32(5, 300001, -50, 50).astype(np.int64) * 3000000007
        v = np.arange(300001, dtype=np.int32)
        dk, dv = T.up(l), T.up(v)
        T.ops.sort_by_key(dk, dv)
        assert np.array_equal(dv.cpu().numpy(), np.argsort(l, kind="stable").astype(np.int32))
    finally:
        L.sort_set_rank(-1)


def test_sort_golden_signed_and_64bit(T, oracle):
    dk, dv = T.up(G["sort_keys"]), T.up(G["sort_vals"])
    T.ops.sort_by_key(dk, dv)
    assert np.array_equal(dk.cpu().numpy(), G["sort_keys_sorted"])
    assert np.array_equal(dv.cpu().numpy(), G["sort_vals_sorted"])
    i = oracle.random_i32(3, 100000, -1000000, 1000000)
    d = T.up(i); T.ops.sort(d)
    assert np.array_equal(d.cpu().numpy(), np.sort(i))
    l = i.astype(np.int64) * 3000000007
    d = T.up(l); T.ops.sort(d)
    assert np.array_equal(d.cpu().numpy(), np.sort(l))


def test_fill_hash_matches_host_restatement(T):
    # the device generator of SURVEY 8(d), restated on the host
    n, seed = 10007, 42
    i = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    t = T.ops.fill_hash(T.torch.empty(n, dtype=T.torch.float64, device=T.dev), seed)
    assert np.array_equal(t.cpu().numpy(), (z >> np.uint64(11)).astype(np.float64) / 2.0 ** 53)
    t = T.ops.fill_hash(T.torch.empty(n, dtype=T.torch.int32, device=T.dev), seed)
    assert np.array_equal(t.cpu().numpy().view(np.uint32), (z >> np.uint64(32)).astype(np.uint32))


def test_scan_lookback_equals_reduce_then_scan_and_is_repeatable(T, oracle, built_lib):
    # single-pass decoupled look-back (integers) vs the deterministic 3-kernel path; repeated to shake out races
    n = (1 << 24) + 4099
    x = oracle.random_u32(77, n)
    want = oracle.inclusive_scan(x)
    d = _u32(T, x)
    built_lib.scan_set_lookback(0)
    try:
        ref = T.ops.inclusive_scan(d, unsigned=True).cpu().numpy().view(np.uint32)
    finally:
        built_lib.scan_set_lookback(1)
    assert np.array_equal(ref, want)
    for rep in range(20):
        out = T.ops.inclusive_scan(d, unsigned=True)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want), rep
    l = (x.astype(np.int64) << 20) - 12345
    dl = T.up(l)
    for rep in range(5):                                  # 64-bit: two status words per tile
        out = T.ops.exclusive_scan(dl, init=5)
        assert np.array_equal(out.cpu().numpy(), oracle.exclusive_scan(l, 5)), rep
    # in place, under load from another stream
    s2 = T.torch.cuda.Stream()
    junk = T.torch.empty(1 << 26, dtype=T.torch.float32, device=T.dev)
    with T.torch.cuda.stream(s2):
        for _ in range(10):
            junk.normal_()
    dd = _u32(T, x)
    T.ops.inclusive_scan(dd, dd, unsigned=True)
    T.torch.cuda.synchronize()
    assert np.array_equal(dd.cpu().numpy().view(np.uint32), want)
