"""Unstructured matrices at size, built on the device (round 5; reference: tests/random_matrix.hpp -- sorted, uniformly random
columns per row -- and the irregular matrices amgcl feeds vex::SpMat):
  random16(n)  -- 16 entries per row, columns uniform in [0, n), sorted within the row;
  powerlaw(n)  -- row lengths floor(6 / sqrt(u)), u uniform in (0, 1], capped at 4096 (mean ~12, a few rows in the thousands),
                  columns uniform, sorted within the row.
Values are a hash of the entry number (ops.fill_hash).  reference_product() evaluates y = A x without any matrix kernel
(gather + segmented sum in torch) together with sum |terms| per row for the tolerance of SURVEY 8(c)."""
import torch


def random16(n, dev, seed=1, per_row=16):
    from vexcl_amd import ops
    g = torch.Generator(device=dev); g.manual_seed(seed)
    col = torch.randint(0, n, (n, per_row), device=dev, dtype=torch.int32, generator=g)
    col = torch.sort(col, dim=1).values.contiguous().view(-1)
    ptr = (torch.arange(n + 1, device=dev, dtype=torch.int64) * per_row).to(torch.int32)
    val = ops.fill_hash(torch.empty(n * per_row, dtype=torch.float64, device=dev), seed + 100)
    return ptr, col, val


def powerlaw(n, dev, seed=2, cap=4096):
    from vexcl_amd import ops
    g = torch.Generator(device=dev); g.manual_seed(seed)
    u = torch.rand(n, device=dev, dtype=torch.float64, generator=g).clamp_(min=1e-12)
    length = (6.0 / torch.sqrt(u)).to(torch.int64).clamp_(1, cap)
    ptr64 = torch.zeros(n + 1, device=dev, dtype=torch.int64)
    torch.cumsum(length, 0, out=ptr64[1:])
    nnz = int(ptr64[-1])
    assert nnz < 2 ** 31, nnz
    rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), length)
    col = torch.randint(0, n, (nnz,), device=dev, dtype=torch.int64, generator=g)
    key = torch.sort(rows * n + col).values           # sorted by (row, column)
    col = (key % n).to(torch.int32)
    del key, rows
    val = ops.fill_hash(torch.empty(nnz, dtype=torch.float64, device=dev), seed + 100)
    return ptr64.to(torch.int32), col, val


def banded16(n, dev, seed=3, per_row=16, half_band=100000):
    """round 6: an unstructured matrix WITH locality -- 16 entries per row, columns uniform within +-half_band of the diagonal (clamped
    to the matrix), sorted within the row: what a mesh-ordered operator looks like to the memory system (x is re-used from the caches)."""
    from vexcl_amd import ops
    g = torch.Generator(device=dev); g.manual_seed(seed)
    off = torch.randint(-half_band, half_band + 1, (n, per_row), device=dev, dtype=torch.int64, generator=g)
    col = (torch.arange(n, device=dev, dtype=torch.int64).view(-1, 1) + off).clamp_(0, n - 1)
    del off
    col = torch.sort(col, dim=1).values.to(torch.int32).contiguous().view(-1)
    ptr = (torch.arange(n + 1, device=dev, dtype=torch.int64) * per_row).to(torch.int32)
    val = ops.fill_hash(torch.empty(n * per_row, dtype=torch.float64, device=dev), seed + 100)
    return ptr, col, val


def stencil27(n, dev, seed=4):
    """round 6: a 27-point operator with a DIFFERENT value in every entry (a hash of the entry number) on a g^3 grid, g = round(n^(1/3)),
    identity rows on the boundary: the widest stencil an AMG hierarchy over a structured mesh produces; nothing to compress."""
    from vexcl_amd import ops
    g = int(round(n ** (1.0 / 3.0)))
    N = g ** 3
    assert N == n, "stencil27 wants a cube number of rows"
    r = torch.arange(N, device=dev, dtype=torch.int32)
    ix, iy, iz = r % g, (r // g) % g, r // (g * g)
    inner = (ix > 0) & (ix < g - 1) & (iy > 0) & (iy < g - 1) & (iz > 0) & (iz < g - 1)
    del ix, iy, iz
    ptr64 = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.where(inner, 27, 1), 0, out=ptr64[1:])
    nnz = int(ptr64[-1])
    assert nnz < 2 ** 31
    col = torch.empty(nnz, dtype=torch.int32, device=dev)
    b = ptr64[:-1]
    bi, ri = b[inner], r[inner]
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                col[bi + k] = ri + (dz * g * g + dy * g + dx)
                k += 1
    del bi, ri
    col[b[~inner]] = r[~inner]
    val = ops.fill_hash(torch.empty(nnz, dtype=torch.float64, device=dev), seed + 100)
    return ptr64.to(torch.int32), col, val


def stencil27_const(g, dev):
    """round 6: a CONSTANT-coefficient 27-point operator on g^3 (centre 26, the 26 neighbours -1; identity rows on the boundary): 27 diagonals,
    three values -- the slices repeat, so the value-coded SELL-512 storage pools them in its dictionary."""
    N = g ** 3
    r = torch.arange(N, device=dev, dtype=torch.int32)
    ix, iy, iz = r % g, (r // g) % g, r // (g * g)
    inner = (ix > 0) & (ix < g - 1) & (iy > 0) & (iy < g - 1) & (iz > 0) & (iz < g - 1)
    del ix, iy, iz
    ptr64 = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.where(inner, 27, 1), 0, out=ptr64[1:])
    nnz = int(ptr64[-1])
    assert nnz < 2 ** 31
    col = torch.empty(nnz, dtype=torch.int32, device=dev); val = torch.empty(nnz, dtype=torch.float64, device=dev)
    b = ptr64[:-1]
    bi, ri = b[inner], r[inner]
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                col[bi + k] = ri + (dz * g * g + dy * g + dx)
                val[bi + k] = 26.0 if (dx, dy, dz) == (0, 0, 0) else -1.0
                k += 1
    del bi, ri
    col[b[~inner]] = r[~inner]; val[b[~inner]] = 1.0
    return ptr64.to(torch.int32), col, val


def reference_product(ptr, col, val, x):
    """(y, sum |terms| per row) without a matrix kernel."""
    n = ptr.numel() - 1
    length = (ptr[1:] - ptr[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(n, device=x.device, dtype=torch.int64), length)
    t = val * x[col.to(torch.int64)]
    y = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(0, rows, t)
    mag = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(0, rows, t.abs())
    return y, mag


MAKERS = {"random16": random16, "powerlaw": powerlaw, "banded16": banded16, "stencil27": stencil27}
ROWS_OF = {"stencil27": 256 ** 3}          # rows of a maker that does not take the bench's UNSTRUCTURED_ROWS
# kernels a product of such a matrix may launch (the ELL part, the CSR arrays); names as rocprofv3 prints them
PRODUCT_KERNELS = ("sell_kernel", "sell_pair_kernel", "sell8_pair_kernel", "hell_kernel", "csr_stream2_kernel", "csr_stream_kernel",
                   "csr_scalar_kernel", "csr_rows_kernel", "sellu_kernel", "sell8_march_kernel", "sell8_grid_kernel", "sell8_plane_kernel", "sell8_kernel", "sell8v_kernel")


def stencil2d(W, H, dev):
    """The benchmark's operator (examples/benchmark.cpp:364-415) in TWO dimensions: 5-point Laplacian on a W x H grid, identity rows
    on the boundary; int32 CSR on the device.  -> (ptr, col, val, h2i)"""
    N = W * H
    r = torch.arange(N, device=dev, dtype=torch.int32)
    ix, iy = r % W, r // W
    inner = (ix > 0) & (ix < W - 1) & (iy > 0) & (iy < H - 1)
    del ix, iy
    ptr64 = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.where(inner, 5, 1), 0, out=ptr64[1:])
    nnz = int(ptr64[-1])
    assert nnz < 2 ** 31
    col = torch.empty(nnz, dtype=torch.int32, device=dev)
    val = torch.empty(nnz, dtype=torch.float64, device=dev)
    h2i = float((W - 1) ** 2)
    b = ptr64[:-1]
    bi, ri = b[inner], r[inner]
    for k, (d, v) in enumerate(((-W, -h2i), (-1, -h2i), (0, 4 * h2i), (1, -h2i), (W, -h2i))):
        col[bi + k] = ri + d
        val[bi + k] = v
    del bi, ri
    bo, ro = b[~inner], r[~inner]
    col[bo] = ro
    val[bo] = 1.0
    return ptr64.to(torch.int32), col, val, h2i


def stencil2d_reference(x, W, H, h2i):
    """(y, sum |terms| per row) of that operator without a matrix: torch slicing on the grid."""
    X = x.view(H, W)
    y = X.clone(); mag = X.abs().clone()
    c = X[1:-1, 1:-1]
    nb = X[:-2, 1:-1], X[1:-1, :-2], X[1:-1, 2:], X[2:, 1:-1]
    y[1:-1, 1:-1] = h2i * (4 * c) - h2i * (nb[0] + nb[1] + nb[2] + nb[3])
    mag[1:-1, 1:-1] = h2i * (4 * c.abs() + nb[0].abs() + nb[1].abs() + nb[2].abs() + nb[3].abs())
    return y.view(-1), mag.view(-1)
