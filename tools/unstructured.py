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


def reference_product(ptr, col, val, x):
    """(y, sum |terms| per row) without a matrix kernel."""
    n = ptr.numel() - 1
    length = (ptr[1:] - ptr[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(n, device=x.device, dtype=torch.int64), length)
    t = val * x[col.to(torch.int64)]
    y = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(0, rows, t)
    mag = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(0, rows, t.abs())
    return y, mag


MAKERS = {"random16": random16, "powerlaw": powerlaw}
# kernels a product of such a matrix may launch (the ELL part, the CSR arrays); names as rocprofv3 prints them
PRODUCT_KERNELS = ("sell_kernel", "sell_pair_kernel", "sell8_pair_kernel", "hell_kernel", "csr_stream2_kernel", "csr_stream_kernel",
                   "csr_scalar_kernel", "csr_rows_kernel", "sellu_kernel")
