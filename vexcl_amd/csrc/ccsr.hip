// Compressed-stencil SpMV (vex::SpMatCCSR, vexcl/spmat/ccsr.hpp:40-53,184-200):
//     y[i] (=|+=) alpha * sum_{j in [row[idx[i]], row[idx[i]+1])} val[j] * x[i + col[j]]
// for matrices with a handful of UNIQUE rows (stencil operators).  No (col,val)
// stream at all: per matrix row the kernel reads 4 B of idx, gathers x and writes y.
// The unique-row tables live in LDS (a few hundred bytes); a lane handles four rows
// 256 apart (all wave accesses contiguous and fully used); workgroups follow the
// same strip traversal as the SELL product so that x stays in one XCD's L2.
// Same summation order as the reference loop; -ffp-contract=off like spmv.hip.
#include "common.hpp"

#include <algorithm>
#include <vector>

namespace vexhip {
namespace {

constexpr int KB = 256;
constexpr int KTABLE = 1024;           // max entries / unique rows staged in LDS

struct strip { int chunk, planes, plane_blocks; };

__device__ __forceinline__ long long strip_block(const strip &t, long long nblocks) {
    const long long b = blockIdx.x;
    if (t.chunk > 0) {
        const long long k = b & 7, q = b >> 3;
        const long long i = q % t.chunk, r = q / t.chunk;
        const long long p = r % t.planes, tile = r / t.planes;
        const long long l = tile * 8 * t.chunk + k * t.chunk + i;
        const long long lb = p * t.plane_blocks + l;
        return (l < t.plane_blocks && lb < nblocks) ? lb : -1;
    }
    return b < nblocks ? b : -1;
}

// A lane handles RPL rows that are 256 apart (row = block + q*256 + lane): every load of x,
// idx and every store of y is a fully used, contiguous 512-byte (256 for idx) wave access, and
// the RPL idx loads are in flight together.  Measured at 512^3 (HBM traffic is the minimum,
// 2.71 GB, in every variant): 2 consecutive rows per lane 0.98 ms, RPL = 2: 1.07, 4: 0.91, 8: 0.89.
template <typename V, bool LDS, int RPL>
__global__ __launch_bounds__(KB)
void ccsr_kernel(long long n, long long nblocks, V alpha, int append,
        const unsigned *__restrict__ idx, int m, const unsigned *__restrict__ row,
        const int *__restrict__ col, const V *__restrict__ val, int entries,
        const V *__restrict__ x, V *__restrict__ y, strip tr)
{
    __shared__ unsigned s_row[LDS ? KTABLE + 1 : 1];
    __shared__ int s_col[LDS ? KTABLE : 1];
    __shared__ V s_val[LDS ? KTABLE : 1];
    if constexpr (LDS) {
        for (int j = threadIdx.x; j <= m; j += KB) s_row[j] = row[j];
        for (int j = threadIdx.x; j < entries; j += KB) { s_col[j] = col[j]; s_val[j] = val[j]; }
        __syncthreads();
    }
    const long long lb = strip_block(tr, nblocks);
    if (lb < 0) return;
    const long long i0 = lb * (KB * RPL) + threadIdx.x;
    unsigned p[RPL];
#pragma unroll
    for (int q = 0; q < RPL; ++q) p[q] = i0 + q * KB < n ? idx[i0 + q * KB] : 0u;

#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        const long long i = i0 + q * KB;
        if (i >= n) continue;
        const unsigned pos = p[q];
        unsigned j = LDS ? s_row[pos] : row[pos];
        const unsigned end = LDS ? s_row[pos + 1] : row[pos + 1];
        // 8 entries at a time: table reads, then 8 independent gathers, then the fold in row
        // order (padding entries gather x[i] and are not added)
        V s = 0;
        for (; j < end; j += 8) {
            int c[8]; V v[8], xv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool in = j + k < end;
                c[k] = in ? (LDS ? s_col[j + k] : col[j + k]) : 0;
                v[k] = in ? (LDS ? s_val[j + k] : val[j + k]) : V(0);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) xv[k] = x[i + c[k]];
#pragma unroll
            for (int k = 0; k < 8; ++k) if (j + k < end) s += v[k] * xv[k];
        }
        V o = alpha * s;
        if (append) o = y[i] + o;
        y[i] = o;
    }
}

int g_ccsr_rpl = 4;

template <typename V, int RPL>
int launch_ccsr(hipStream_t st, int64_t n, V alpha, int append, const uint32_t *idx, int64_t m,
        const uint32_t *row, const int32_t *col, const V *val, int64_t entries, int64_t s_big, const V *x, V *y)
{
    constexpr long long ROWS = (long long)KB * RPL;
    const long long nb = (n + ROWS - 1) / ROWS;
    // strip traversal when the farthest offset is a "plane" too long for 3 planes of x in L2
    strip tr = {0, 0, 0};
    long long grid = nb;
    if (s_big >= 2 * 65536 && s_big % ROWS == 0 && n >= 8 * 65536) {
        const long long plane_blocks = s_big / ROWS;
        const long long chunk = std::max<long long>(1, std::min<long long>(64 * 512 / ROWS, plane_blocks / 8));
        const long long planes = (nb + plane_blocks - 1) / plane_blocks;
        const long long tiles = (plane_blocks + 8 * chunk - 1) / (8 * chunk);
        tr = strip{(int)chunk, (int)planes, (int)plane_blocks};
        grid = tiles * planes * 8 * chunk;
    }
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    if (m <= KTABLE && entries <= KTABLE)
        ccsr_kernel<V, true, RPL><<<(unsigned)grid, KB, 0, st>>>(n, nb, alpha, append, idx, (int)m, row, col, val, (int)entries, x, y, tr);
    else
        ccsr_kernel<V, false, RPL><<<(unsigned)grid, KB, 0, st>>>(n, nb, alpha, append, idx, (int)m, row, col, val, (int)entries, x, y, tr);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

template <typename V>
int spmv_ccsr(int dev, void *stream, int64_t n, V alpha, int append, const uint32_t *idx, int64_t m,
        const uint32_t *row, const int32_t *col, const V *val, int64_t entries, int64_t s_big,
        const V *x, V *y)
{
    VEXHIP_REQUIRE(n >= 0 && m >= 0 && entries >= 0, "negative size");
    if (n == 0) return 0;
    VEXHIP_REQUIRE(idx && row && x && y && (entries == 0 || (col && val)), "NULL argument");
        VEXHIP_SET_DEVICE(dev);
    hipStream_t st = as_stream(stream);
    if (g_ccsr_rpl == 4)
        return launch_ccsr<V, 4>(st, n, alpha, append, idx, m, row, col, val, entries, s_big, x, y);
    if (g_ccsr_rpl == 8)
        return launch_ccsr<V, 8>(st, n, alpha, append, idx, m, row, col, val, entries, s_big, x, y);
    if (g_ccsr_rpl == 1)
        return launch_ccsr<V, 1>(st, n, alpha, append, idx, m, row, col, val, entries, s_big, x, y);
    return launch_ccsr<V, 2>(st, n, alpha, append, idx, m, row, col, val, entries, s_big, x, y);
}

} // namespace
} // namespace vexhip

using namespace vexhip;

extern "C" {

int vexhip_spmv_ccsr_f64(int dev, void *stream, int64_t n, double alpha, int append, const uint32_t *idx, int64_t m,
        const uint32_t *row, const int32_t *col, const double *val, int64_t entries, int64_t far_offset,
        const double *x, double *y)
{ return spmv_ccsr<double>(dev, stream, n, alpha, append, idx, m, row, col, val, entries, far_offset, x, y); }

int vexhip_spmv_ccsr_f32(int dev, void *stream, int64_t n, float alpha, int append, const uint32_t *idx, int64_t m,
        const uint32_t *row, const int32_t *col, const float *val, int64_t entries, int64_t far_offset,
        const float *x, float *y)
{ return spmv_ccsr<float>(dev, stream, n, alpha, append, idx, m, row, col, val, entries, far_offset, x, y); }

int vexhip_spmv_ccsr_set_rows_per_lane(int rpl) { g_ccsr_rpl = rpl; return 0; }

} // extern "C"
