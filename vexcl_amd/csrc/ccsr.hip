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

// 32-bit arithmetic: launch grids are < 2^31 (checked on the host) and a 64-bit division costs ~100 instructions,
// four of them per workgroup in the 64-bit version of this function
__device__ __forceinline__ long long strip_block(const strip &t, long long nblocks) {
    const unsigned b = blockIdx.x;
    if (t.chunk > 0) {
        const unsigned k = b & 7u, q = b >> 3, chunk = (unsigned)t.chunk, planes = (unsigned)t.planes;
        const unsigned r = q / chunk, i = q - r * chunk;
        const unsigned tile = r / planes, p = r - tile * planes;
        const long long l = (long long)tile * (8 * chunk) + k * chunk + i;
        const long long lb = (long long)p * t.plane_blocks + l;
        return (l < t.plane_blocks && lb < nblocks) ? lb : -1;
    }
    return b < nblocks ? (long long)b : -1;
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

// Pair form (round 2, the default): a lane owns rows 2t and 2t+1 of a 512-row block and reads x for both with ONE
// 16-byte load per stencil entry when the two rows use the same unique row (idx equal: everywhere but next to a
// boundary); otherwise each row walks its own entries with 8-byte loads.  Half the vector-memory instructions of the
// kernel above (4.5 instead of 9 per row).  Measured at 512^3 (tools/r02_spmv_ab.py): 0.89 -> 0.87 ms with the gathers
// under per-lane conditions, 0.83 ms with unconditional gathers (see below).  With 20 bytes of HBM traffic per row this
// product is not bound by HBM; the ablation of the value-coded SELL product (profiles/r02_sell8v_ablation.json) prices
// its ingredients: x and y streamed once 0.33 ms, seven 16-byte gathers +0.27 ms, a per-row code stream the gathers
// depend on +0.16 ms -- the position word here plays the role of the codes there.  Loading the positions BEFORE the
// tables are staged, so that both are in flight together, changes nothing (0.864 against 0.867 ms).
// Nor does removing the position stream: with the slice dictionary of sell8.hip applied to idx (two distinct 512-row blocks
// for this operator, read from L1 instead of 0.54 GB from HBM) the product takes 0.8135 against 0.8165 ms -- dropped.
// Entries are taken eight at a time: table reads, gathers, then the fold in row order (same order as the reference's
// loop, ccsr.hpp:184-200): bit-identical to the kernel above.
template <typename V, bool LDS>
__global__ __launch_bounds__(KB)
void ccsr_pair_kernel(long long n, long long nblocks, V alpha, int append,
        const unsigned *__restrict__ idx, int m, const unsigned *__restrict__ row,
        const int *__restrict__ col, const V *__restrict__ val, int entries,
        const V *__restrict__ x, V *__restrict__ y, strip tr)
{
    typedef V V2 __attribute__((ext_vector_type(2)));
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
    const long long i = lb * (2 * KB) + 2 * threadIdx.x;
    if (i >= n) return;
    const bool two = i + 1 < n;
    unsigned p0, p1;
    if (two && (reinterpret_cast<unsigned long long>(idx) & 7) == 0) {
        const uint2 pp = *reinterpret_cast<const uint2 *>(idx + i);
        p0 = pp.x; p1 = pp.y;
    } else { p0 = idx[i]; p1 = two ? idx[i + 1] : p0; }
    // Gathers are UNCONDITIONAL inside a wave-uniform branch: a load under a per-lane condition (`if (j + k < end)`) is
    // compiled as an exec-masked block with a full vmcnt(0) wait behind it, which serialises the gathers of a row (one
    // memory round trip each; that was the 0.87 ms of the first version of this kernel).  Lanes without an entry k read
    // x[i] (a valid address: the operator is square) and discard it; their table value is 0 and the sum is kept by a
    // select, so NaN / Inf in x cannot leak into rows that do not reference them.
    V s0 = 0, s1 = 0;
    if (p0 == p1 && two) {
        unsigned j = LDS ? s_row[p0] : row[p0];
        const unsigned end = LDS ? s_row[p0 + 1] : row[p0 + 1];
        for (; j < end; j += 8) {
            int c[8]; V v[8]; V2 xv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool in = j + k < end;
                const unsigned e = in ? j + k : j;
                c[k] = LDS ? s_col[e] : col[e];
                v[k] = LDS ? s_val[e] : val[e];
                c[k] = in ? c[k] : 0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {          // a load no lane of the wave needs is not issued
                xv[k].x = V(0); xv[k].y = V(0);
                if (__builtin_amdgcn_ballot_w64(j + k < end) != 0) __builtin_memcpy(&xv[k], x + (i + c[k]), sizeof(V2));
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const V t0 = s0 + v[k] * xv[k].x, t1 = s1 + v[k] * xv[k].y;
                const bool in = j + k < end;
                s0 = in ? t0 : s0; s1 = in ? t1 : s1;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (q == 1 && !two) break;
            const unsigned pos = q ? p1 : p0;
            unsigned j = LDS ? s_row[pos] : row[pos];
            const unsigned end = LDS ? s_row[pos + 1] : row[pos + 1];
            V s = 0;
            for (; j < end; j += 8) {
                int c[8]; V v[8], xv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool in = j + k < end;
                    const unsigned e = in ? j + k : j;
                    c[k] = LDS ? s_col[e] : col[e];
                    v[k] = LDS ? s_val[e] : val[e];
                    c[k] = in ? c[k] : 0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) { xv[k] = V(0); if (__builtin_amdgcn_ballot_w64(j + k < end) != 0) xv[k] = x[i + q + c[k]]; }
#pragma unroll
                for (int k = 0; k < 8; ++k) { const V t = s + v[k] * xv[k]; s = (j + k < end) ? t : s; }
            }
            if (q) s1 = s; else s0 = s;
        }
    }
    if (two && (reinterpret_cast<unsigned long long>(y) & (2 * sizeof(V) - 1)) == 0) {
        V2 *yp = reinterpret_cast<V2 *>(y + i);
        V2 o; o.x = alpha * s0; o.y = alpha * s1;
        if (append) { const V2 old = *yp; o.x = old.x + o.x; o.y = old.y + o.y; *yp = o; }
        else __builtin_nontemporal_store(o, yp);
    } else {
        V o = alpha * s0; if (append) o = y[i] + o; y[i] = o;
        if (two) { V o1 = alpha * s1; if (append) o1 = y[i + 1] + o1; y[i + 1] = o1; }
    }
}

int g_ccsr_rpl = 0;      // 0: pair form (default); 1, 2, 4, 8: rows per lane of the first form (A/B)

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
    if (g_ccsr_rpl == 0) {
        constexpr long long ROWS = 2 * KB;
        const long long nb = (n + ROWS - 1) / ROWS;
        strip tr = {0, 0, 0};
        long long grid = nb;
        if (s_big >= 2 * 65536 && s_big % ROWS == 0 && n >= 8 * 65536) {       // strip traversal, as launch_ccsr
            const long long plane_blocks = s_big / ROWS;
            const long long chunk = std::max<long long>(1, std::min<long long>(64, plane_blocks / 8));
            const long long planes = (nb + plane_blocks - 1) / plane_blocks;
            const long long tiles = (plane_blocks + 8 * chunk - 1) / (8 * chunk);
            tr = strip{(int)chunk, (int)planes, (int)plane_blocks};
            grid = tiles * planes * 8 * chunk;
        }
        VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
        if (m <= KTABLE && entries <= KTABLE)
            ccsr_pair_kernel<V, true><<<(unsigned)grid, KB, 0, st>>>(n, nb, alpha, append, idx, (int)m, row, col, val, (int)entries, x, y, tr);
        else
            ccsr_pair_kernel<V, false><<<(unsigned)grid, KB, 0, st>>>(n, nb, alpha, append, idx, (int)m, row, col, val, (int)entries, x, y, tr);
        VEXHIP_LAUNCH_CHECK();
        return 0;
    }
    if (g_ccsr_rpl == 4)
        return launch_ccsr<V, 4>(st, n, alpha, append, idx, m, row, col, val, entries, s_big, x, y);
    if (g_ccsr_rpl == 8)
        return launch_ccsr<V, 8>(st, n, alpha, append, idx, m, row, col, val, entries, s_big, x, y);
    if (g_ccsr_rpl == 1)
        return launch_ccsr<V, 1>(st, n, alpha, append, idx, m, row, col, val, entries, s_big, x, y);
    return launch_ccsr<V, 2>(st, n, alpha, append, idx, m, row, col, val, entries, s_big, x, y);
}

// ---- CCSR -> CSR on the device: row i becomes the table row idx[i] with its offsets made absolute.  vex::SpMatCCSR uses it
//      to hand its operator to vexhip_spmat (diagonal / value codes + slice dictionary: the faster product for such matrices).
__global__ __launch_bounds__(256)
void ccsr_len_kernel(long long n, const unsigned *__restrict__ idx, const unsigned *__restrict__ row, int *__restrict__ ptr)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (long long)gridDim.x * blockDim.x)
        ptr[i] = i < n ? (int)(row[idx[i] + 1] - row[idx[i]]) : 0;
}

template <typename V>
__global__ __launch_bounds__(256)
void ccsr_expand_kernel(long long n, const unsigned *__restrict__ idx, const unsigned *__restrict__ row,
        const int *__restrict__ col, const V *__restrict__ val, const int *__restrict__ ptr, int *__restrict__ out_col, V *__restrict__ out_val,
        int *__restrict__ out_of_range)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned p = idx[i];
        long long o = ptr[i];
        for (unsigned j = row[p], e = row[p + 1]; j < e; ++j, ++o) {
            const long long c = i + col[j];
            if (c < 0 || c >= n) atomicExch(out_of_range, 1);
            out_col[o] = (int)c; out_val[o] = val[j];
        }
    }
}

template <typename V>
int ccsr_to_csr(int dev, void *stream, int64_t n, const uint32_t *idx, const uint32_t *row, const int32_t *col, const V *val,
        int32_t *ptr, int32_t *out_col, V *out_val, int64_t *nnz)
{
    VEXHIP_REQUIRE(n >= 0 && nnz, "bad argument");
    *nnz = 0;
    if (n == 0) return 0;
    VEXHIP_REQUIRE(idx && row && ptr, "NULL argument");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    const int grid = (int)std::min<int64_t>((n + 256) / 256, (int64_t)info(dev).cus * 32);
    if (!out_col) {                       // phase 1: row pointers and the number of entries
        ccsr_len_kernel<<<grid, 256, 0, s>>>(n, idx, row, ptr);
        VEXHIP_LAUNCH_CHECK();
        void *tmp = nullptr;
        const size_t tb = vexhip_scan_tmp_bytes(VEXHIP_I32, n + 1);
        if (tb) VEXHIP_TRY(hipMalloc(&tmp, tb));
        const int zero = 0; int last = 0;
        int rc = vexhip_scan(dev, stream, VEXHIP_I32, 1, &zero, ptr, ptr, n + 1, tmp);
        hipError_t e = rc ? hipSuccess : hipMemcpyAsync(&last, ptr + n, sizeof(int), hipMemcpyDeviceToHost, s);
        if (!rc && e == hipSuccess) e = hipStreamSynchronize(s);
        if (tmp) (void)hipFree(tmp);
        if (rc) return rc;
        VEXHIP_TRY(e);
        VEXHIP_REQUIRE(last >= 0, "CCSR operator has 2^31 or more entries");
        *nnz = last;
        return 0;
    }
    VEXHIP_REQUIRE(col && val && out_val, "NULL argument");
    int *flag = nullptr, h = 0;
    VEXHIP_TRY(hipMalloc(&flag, sizeof(int)));
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), s);
    if (e == hipSuccess) { ccsr_expand_kernel<V><<<grid, 256, 0, s>>>(n, idx, row, col, val, ptr, out_col, out_val, flag); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(flag);
    VEXHIP_TRY(e);
    *nnz = h ? -1 : 0;                    // -1: an entry refers to a column outside [0, n)
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

int vexhip_ccsr_to_csr_f64_i32(int dev, void *stream, int64_t n, const uint32_t *idx, const uint32_t *row, const int32_t *col, const double *val,
        int32_t *ptr, int32_t *out_col, double *out_val, int64_t *nnz)
{ return ccsr_to_csr<double>(dev, stream, n, idx, row, col, val, ptr, out_col, out_val, nnz); }
int vexhip_ccsr_to_csr_f32_i32(int dev, void *stream, int64_t n, const uint32_t *idx, const uint32_t *row, const int32_t *col, const float *val,
        int32_t *ptr, int32_t *out_col, float *out_val, int64_t *nnz)
{ return ccsr_to_csr<float>(dev, stream, n, idx, row, col, val, ptr, out_col, out_val, nnz); }

int vexhip_spmv_ccsr_set_rows_per_lane(int rpl) { g_ccsr_rpl = rpl; return 0; }

} // extern "C"

VEXHIP_WARM_TU(ccsr)
