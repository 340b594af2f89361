"""GPU parity tests for vex::FFT through the C ABI (vexhip_fft_*), against the oracle (numpy's pocketfft per axis,
itself pinned by the DFT definition in tests/test_oracle.py).  Cases follow the reference's tests/fft.cpp:
round trips fft -> ifft (:26-52, :54-103), random dimensions / batches / awkward lengths (:112-153), real input.
Tolerance: fp64 |delta| <= 1e-12 * n-scaled norm (rms relative 1e-12; the reference accepts 1e-8), fp32 1e-5."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


def _rand(seed, n):
    r = np.random.default_rng(seed)
    return r.standard_normal(n) + 1j * r.standard_normal(n)


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


# lengths: trivial, every radix, mixed radix, one LDS row (2048), four-step (4096 ... 2^20), 11 * 1365 (four-step with
# an odd split), primes and composites with a large prime factor (Bluestein), Bluestein over a four-step convolution
LENGTHS = [1, 2, 3, 4, 5, 7, 8, 11, 13, 16, 64, 100, 243, 1000, 1024, 2048, 2187, 4096, 5000, 15015, 1 << 16, 1 << 20,
           17, 1009, 2018, 2039, 4099, 65537]


@pytest.mark.parametrize("n", LENGTHS)
def test_fft_1d_fp64(T, oracle, n):
    x = _rand(n, n)
    f = T.ops.FFT([n], T.ops.FORWARD)
    y = f(T.up(x)).cpu().numpy()
    assert _rel(y, oracle.fft_nd(x, [n], [oracle.FFT_FORWARD])) < 1e-12
    g = T.ops.FFT([n], T.ops.INVERSE)
    back = g(T.up(y)).cpu().numpy()
    assert _rel(back, x) < 1e-12                          # tests/fft.cpp:98-102 round trip
    yi = g(T.up(x)).cpu().numpy()
    assert _rel(yi, oracle.fft_nd(x, [n], [oracle.FFT_INVERSE])) < 1e-12


@pytest.mark.parametrize("n", [1, 8, 100, 1024, 4096, 8192, 15015, 1009, 2039, 4093, 1 << 18])
def test_fft_1d_fp32(T, oracle, n):
    x = _rand(n + 1, n).astype(np.complex64)
    f = T.ops.FFT([n], T.ops.FORWARD, dtype=T.torch.complex64)
    y = f(T.up(x)).cpu().numpy()
    assert _rel(y.astype(np.complex128), oracle.fft_nd(x, [n], [oracle.FFT_FORWARD])) < 2e-6


CASES = [
    ([7, 12], [0, 0]),                      # 2-D
    ([64, 64], [0, 0]),
    ([30, 40], [1, 1]),                     # 2-D inverse
    ([5, 6, 7], [0, 0, 0]),                 # 3-D
    ([16, 3000], [0, 0]),                   # a long row next to a short one
    ([3000, 16], [0, 0]),
    ([33, 1024], [2, 0]),                   # batch of rows (tests/fft.cpp:70-74)
    ([100, 17], [2, 0]),                    # batch of Bluestein rows
    ([4, 9, 10], [2, 0, 0]),                # batch of 2-D transforms
    ([12, 5, 8], [0, 2, 0]),                # a `none` dimension in the middle
    ([12, 40], [0, 2]),                     # transform along the slow dimension only
    ([6, 1, 9], [0, 0, 1]),                 # unit dimension, mixed directions
    ([3, 4100], [2, 0]),                    # batch of four-step rows
    ([2, 2053], [2, 1]),                    # batch, Bluestein, inverse
]


@pytest.mark.parametrize("sizes,dirs", CASES)
def test_fft_nd(T, oracle, sizes, dirs):
    n = int(np.prod(sizes))
    x = _rand(n, n)
    f = T.ops.FFT(sizes, dirs)
    y = f(T.up(x)).cpu().numpy()
    assert _rel(y, oracle.fft_nd(x, sizes, dirs)) < 1e-12


def test_fft_random_dimensions(T, oracle):
    """tests/fft.cpp:112-153: 100 random shapes (mostly 1-D, sizes mostly of the planner's best sizes), batches,
    forward then inverse returns the input."""
    r = np.random.default_rng(7)
    done = 0
    for it in range(300):
        dims = 1 + int(5 * r.random() ** 3)
        batch = 1 + int(100 * r.random() ** 5)
        d_max = int(4096 ** (1.0 / dims))
        ns = []
        for _ in range(dims):
            sz = 1 + int(d_max * r.random() ** (3 if dims == 1 else 1))
            if r.integers(3) != 0:
                sz = T.ops.fft_best_size(sz)
            ns.append(sz)
        total = batch * int(np.prod(ns))
        if total > 4096:
            continue
        sizes, fd, idr = list(ns), [0] * dims, [1] * dims
        if batch != 1:
            sizes, fd, idr = [batch] + sizes, [2] + fd, [2] + idr
        x = _rand(it, total)
        y = T.ops.FFT(sizes, fd)(T.up(x))
        assert _rel(y.cpu().numpy(), oracle.fft_nd(x, sizes, fd)) < 1e-12
        back = T.ops.FFT(sizes, idr)(y).cpu().numpy()
        assert _rel(back, x) < 1e-12
        done += 1
        if done == 100:
            break
    assert done >= 50


def test_fft_best_size(T, oracle):
    for n in list(range(1, 300)) + [1000, 1025, 4097, 65537, 10 ** 6 + 3]:
        b = T.ops.fft_best_size(n)
        assert b == oracle.fft_best_size(n)
        m = b
        for p in (2, 3, 5, 7):
            while m % p == 0:
                m //= p
        assert m == 1 and b >= n


def test_fft_linearity_and_parseval_large(T):
    """Size-independent properties at a size the CPU oracle would take long on (2^24 fp64 = 256 MiB per buffer):
    Parseval, and the transform of a shifted delta is a pure phase."""
    n = 1 << 24
    torch = T.torch
    g = torch.Generator(device=T.dev); g.manual_seed(5)
    x = torch.randn(n, dtype=torch.float64, device=T.dev, generator=g) + 1j * torch.randn(n, dtype=torch.float64, device=T.dev, generator=g)
    f = T.ops.FFT([n], T.ops.FORWARD)
    y = f(x)
    ex, ey = float((x.abs() ** 2).sum()), float((y.abs() ** 2).sum())
    assert abs(ey / n - ex) <= 1e-12 * ex
    d = torch.zeros(n, dtype=torch.complex128, device=T.dev)
    d[12345] = 1.0
    yd = f(d)
    k = torch.tensor([0, 1, 77, n // 2, n - 1], device=T.dev)
    want = torch.exp(-2j * np.pi * ((k.double() * 12345) % n) / n)
    assert float((yd[k] - want).abs().max()) < 1e-12
    assert float((yd.abs() - 1).abs().max()) < 1e-12
    back = T.ops.FFT([n], T.ops.INVERSE)(y)
    assert float((back - x).abs().max()) < 1e-12 * float(x.abs().max()) * 10
