// Compressed-stencil SpMV (vex::SpMatCCSR, vexcl/spmat/ccsr.hpp:40-53,184-200):
//     y[i] (=|+=) alpha * sum_{j in [row[idx[i]], row[idx[i]+1])} val[j] * x[i + col[j]]
// for matrices with a handful of UNIQUE rows (stencil operators).  No (col,val)
// stream at all: per matrix row the kernel reads 4 B of idx, gathers x and writes y.
// The unique-row tables live in LDS (a few hundred bytes); a lane handles two
// consecutive rows (8-byte idx load, 16-byte y store); workgroups follow the
// same strip traversal as the SELL product so that x stays in one XCD's L2.
// Same summation order as the reference loop; -ffp-contract=off like spmv.hip.
#include "common.hpp"

#include <algorithm>
#include <vector>

namespace vexhip {
namespace {

constexpr int KB = 256;
constexpr int KROWS = 512;             // rows per workgroup (2 per lane)
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

template <typename V, bool LDS>
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
    const long long i0 = lb * KROWS + 2 * threadIdx.x;
    if (i0 >= n) return;
    const bool two = i0 + 1 < n;
    unsigned p0, p1 = 0;
    if (two) { uint2 pp = *reinterpret_cast<const uint2 *>(idx + i0); p0 = pp.x; p1 = pp.y; }
    else p0 = idx[i0];

    V sum[2] = {V(0), V(0)};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (q == 1 && !two) break;
        const unsigned pos = q ? p1 : p0;
        const long long i = i0 + q;
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
        sum[q] = s;
    }
    if (two) {
        typedef V v2 __attribute__((ext_vector_type(2)));
        v2 o; o.x = alpha * sum[0]; o.y = alpha * sum[1];
        v2 *yp = reinterpret_cast<v2 *>(y + i0);
        if (append) { v2 old = *yp; o.x = old.x + o.x; o.y = old.y + o.y; }
        *yp = o;
    } else {
        V o = alpha * sum[0];
        if (append) o = y[i0] + o;
        y[i0] = o;
    }
}

template <typename V>
int spmv_ccsr(int dev, void *stream, int64_t n, V alpha, int append, const uint32_t *idx, int64_t m,
        const uint32_t *row, const int32_t *col, const V *val, int64_t entries, int64_t s_big,
        const V *x, V *y)
{
    VEXHIP_REQUIRE(n >= 0 && m >= 0 && entries >= 0, "negative size");
    if (n == 0) return 0;
    VEXHIP_REQUIRE(idx && row && x && y && (entries == 0 || (col && val)), "NULL argument");
    VEXHIP_REQUIRE((reinterpret_cast<uintptr_t>(idx) & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0, "idx / y must be 8 / 16-byte aligned");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t st = as_stream(stream);
    const long long nb = (n + KROWS - 1) / KROWS;
    // strip traversal when the farthest offset is a "plane" too long for 3 planes of x in L2
    strip tr = {0, 0, 0};
    long long grid = nb;
    if (s_big >= 2 * 65536 && s_big % KROWS == 0 && n >= 8 * 65536) {
        const long long plane_blocks = s_big / KROWS;
        const long long chunk = std::max<long long>(1, std::min<long long>(64, plane_blocks / 8));
        const long long planes = (nb + plane_blocks - 1) / plane_blocks;
        const long long tiles = (plane_blocks + 8 * chunk - 1) / (8 * chunk);
        tr = strip{(int)chunk, (int)planes, (int)plane_blocks};
        grid = tiles * planes * 8 * chunk;
    }
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    if (m <= KTABLE && entries <= KTABLE)
        ccsr_kernel<V, true><<<(unsigned)grid, KB, 0, st>>>(n, nb, alpha, append, idx, (int)m, row, col, val, (int)entries, x, y, tr);
    else
        ccsr_kernel<V, false><<<(unsigned)grid, KB, 0, st>>>(n, nb, alpha, append, idx, (int)m, row, col, val, (int)entries, x, y, tr);
    VEXHIP_LAUNCH_CHECK();
    return 0;
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

} // extern "C"
