// Sparse matrix-vector products for gfx950 (MI355X): the headline path.
//
//   csr_stream_kernel   restates `csr_spmv` (vexcl/spmat/csr.inl:153-171)
//   hell_kernel         restates `hybrid_ell_spmv` (vexcl/spmat/hybrid_ell.inl:238-269)
//
// Both are HBM-bandwidth bound (fp64 AI ~ 0.13 flop/B): no MFMA.  The design
// levers are 16-byte coalesced streams of (col,val), gathers of x that stay in
// L2 / Infinity Cache, full wave occupancy and >> 256 workgroups per launch.
//
// This file is compiled with -ffp-contract=off: every product val*x is rounded
// before it is added, and each row is folded sequentially in storage order --
// the same arithmetic as the reference's host check loop (tests/spmv.cpp:28-32)
// built without FMA, so results are bit-identical to the oracle, not merely
// within the stated 1e-10 tolerance.
#include "common.hpp"

namespace vexhip {
namespace {

typedef int    int2v  __attribute__((ext_vector_type(2)));
typedef int    int4v  __attribute__((ext_vector_type(4)));
typedef long long long2v __attribute__((ext_vector_type(2)));
typedef double double2v __attribute__((ext_vector_type(2)));
typedef float  float2v  __attribute__((ext_vector_type(2)));
typedef float  float4v  __attribute__((ext_vector_type(4)));

template <bool NT, typename T>
__device__ __forceinline__ T ld(const T *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// load 4 consecutive elements starting at a 16-byte aligned address
template <bool NT> __device__ __forceinline__ void load4(const int *p, int (&o)[4]) {
    int4v v = ld<NT>(reinterpret_cast<const int4v *>(p));
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <bool NT> __device__ __forceinline__ void load4(const long long *p, long long (&o)[4]) {
    long2v a = ld<NT>(reinterpret_cast<const long2v *>(p));
    long2v b = ld<NT>(reinterpret_cast<const long2v *>(p) + 1);
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
}
template <bool NT> __device__ __forceinline__ void load4(const double *p, double (&o)[4]) {
    double2v a = ld<NT>(reinterpret_cast<const double2v *>(p));
    double2v b = ld<NT>(reinterpret_cast<const double2v *>(p) + 1);
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
}
template <bool NT> __device__ __forceinline__ void load4(const float *p, float (&o)[4]) {
    float4v v = ld<NT>(reinterpret_cast<const float4v *>(p));
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}

// XCD-aware remap (cdna_hip_programming.md T1): hardware places block b on
// XCD b % 8.  With SWZ the 8 XCDs each walk one contiguous eighth of the row
// range, so the x-planes a stencil row touches are re-used inside ONE 4 MiB L2
// instead of being fetched by all eight.  Speed only; any placement is correct.
template <bool SWZ>
__device__ __forceinline__ long long logical_block(long long nblocks) {
    long long b = blockIdx.x;
    if constexpr (SWZ) {
        long long per = (nblocks + 7) / 8;
        long long lb = (b % 8) * per + b / 8;
        return lb;     // may be >= nblocks for the ragged tail: caller checks
    }
    return b;
}

// ---------------------------------------------------------------------------
// CSR, LDS row staging.  One workgroup = 256 consecutive rows.  The workgroup
// streams the rows' contiguous (col,val) range with 16-byte loads (4 entries
// per lane per step, start aligned down to a multiple of 4 entries), multiplies
// by the gathered x and parks the products in LDS; then lane t folds row t's
// products in CSR order.  Rows longer than a tile are folded across tiles.
// ---------------------------------------------------------------------------
constexpr int CSR_BLOCK = 256;
constexpr int CSR_TILE  = 2048;          // entries per LDS tile (16 KiB fp64)

template <typename V, typename I, bool NT, bool SWZ>
__global__ __launch_bounds__(CSR_BLOCK)
void csr_stream_kernel(long long n, long long nblocks, V alpha, int append,
        const I *__restrict__ ptr, const I *__restrict__ col, const V *__restrict__ val,
        const V *__restrict__ x, V *__restrict__ y)
{
    __shared__ V s_prod[CSR_TILE];
    __shared__ I s_ptr[CSR_BLOCK + 1];

    const long long lb = logical_block<SWZ>(nblocks);
    if (lb >= nblocks) return;
    const int t = threadIdx.x;
    const long long r0 = lb * CSR_BLOCK;
    const long long rows_here = (n - r0 < CSR_BLOCK) ? (n - r0) : CSR_BLOCK;

    // row pointers of this workgroup: coalesced, shared through LDS
    {
        long long r = r0 + t; if (r > n) r = n;
        s_ptr[t] = ptr[r];
        if (t == 0) { long long re = r0 + CSR_BLOCK; if (re > n) re = n; s_ptr[CSR_BLOCK] = ptr[re]; }
    }
    __syncthreads();
    const long long base = s_ptr[0];
    const long long end  = s_ptr[CSR_BLOCK];
    const long long my_lo = s_ptr[t];
    const long long my_hi = (t < rows_here) ? (long long)s_ptr[t + 1] : my_lo;

    V sum = 0;
    for (long long tb = base & ~3ll; tb < end; tb += CSR_TILE) {
        const long long te = (end - tb > CSR_TILE) ? tb + CSR_TILE : end;
#pragma unroll 2
        for (long long k = tb + 4 * t; k < te; k += 4 * CSR_BLOCK) {
            if (te - k >= 4) {
                I c[4]; V v[4];
                load4<NT>(col + k, c);
                load4<NT>(val + k, v);
                V p0 = v[0] * x[c[0]];
                V p1 = v[1] * x[c[1]];
                V p2 = v[2] * x[c[2]];
                V p3 = v[3] * x[c[3]];
                V *d = s_prod + (int)(k - tb);
                d[0] = p0; d[1] = p1; d[2] = p2; d[3] = p3;
            } else {
                for (long long q = k; q < te; ++q) s_prod[(int)(q - tb)] = val[q] * x[col[q]];
            }
        }
        __syncthreads();
        const int lo = (int)((my_lo > tb ? my_lo : tb) - tb);
        const int hi = (int)((my_hi < te ? my_hi : te) - tb);
        for (int j = lo; j < hi; ++j) sum += s_prod[j];
        if (te < end) __syncthreads();
    }

    // "+=" on an empty row leaves y untouched: no read-modify-write traffic for
    // the mostly-empty remote part of a partitioned matrix
    if (t < rows_here && !(append && my_lo == my_hi)) {
        V r = alpha * sum;
        if (append) r = y[r0 + t] + r;
        y[r0 + t] = r;
    }
}

// Fallback for CSR arrays that are not 16-byte aligned (sub-views): the
// reference's one-row-per-work-item loop, unchanged.
template <typename V, typename I>
__global__ __launch_bounds__(256)
void csr_scalar_kernel(long long n, V alpha, int append,
        const I *__restrict__ ptr, const I *__restrict__ col, const V *__restrict__ val,
        const V *__restrict__ x, V *__restrict__ y)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        V sum = 0;
        for (I j = ptr[i], e = ptr[i + 1]; j < e; ++j) sum += val[j] * x[col[j]];
        V r = alpha * sum;
        if (append) r = y[i] + r;
        y[i] = r;
    }
}

// ---------------------------------------------------------------------------
// Hybrid ELL.  ELL part column-major (element (i,j) at i + j*pitch, pitch a
// multiple of 16, padding column -1), CSR tail for the rows wider than the ELL
// width.  RPT consecutive rows per lane: RPT = 2 gives 8-byte column and
// 16-byte value loads; the gathers of x stay (nearly) contiguous across the
// wave because adjacent rows of a banded matrix read adjacent columns.
// W > 0: compile-time ELL width (fully unrolled, all loads issued before the
// first dependent gather); W == 0: run-time width.
// ---------------------------------------------------------------------------
template <int RPT> struct colvec;
template <> struct colvec<1> { typedef int   type; };
template <> struct colvec<2> { typedef int2v type; };
template <> struct colvec<4> { typedef int4v type; };

template <typename V, int RPT, bool NT>
__device__ __forceinline__ void ell_load(const int *cp, const V *vp, int (&c)[RPT], V (&v)[RPT]) {
    if constexpr (RPT == 1) {
        c[0] = ld<NT>(cp); v[0] = ld<NT>(vp);
    } else if constexpr (RPT == 2) {
        int2v cc = ld<NT>(reinterpret_cast<const int2v *>(cp));
        c[0] = cc.x; c[1] = cc.y;
        if constexpr (sizeof(V) == 8) {
            double2v vv = ld<NT>(reinterpret_cast<const double2v *>(vp));
            v[0] = vv.x; v[1] = vv.y;
        } else {
            float2v vv = ld<NT>(reinterpret_cast<const float2v *>(vp));
            v[0] = vv.x; v[1] = vv.y;
        }
    } else {
        load4<NT>(cp, c);
        load4<NT>(vp, v);
    }
}

template <typename V, int RPT, int W, bool NT, bool SWZ>
__global__ __launch_bounds__(256)
void hell_kernel(long long n, long long nblocks, V alpha, int append,
        int ell_w, long long pitch,
        const int *__restrict__ ell_col, const V *__restrict__ ell_val,
        const int *__restrict__ csr_ptr, const int *__restrict__ csr_col, const V *__restrict__ csr_val,
        const V *__restrict__ x, V *__restrict__ y)
{
    const long long lb = logical_block<SWZ>(nblocks);
    if (lb >= nblocks) return;
    const long long i = (lb * 256 + threadIdx.x) * RPT;
    if (i >= n) return;

    V sum[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) sum[r] = 0;

    if constexpr (W > 0) {
        int c[W][RPT]; V v[W][RPT];
#pragma unroll
        for (int j = 0; j < W; ++j) ell_load<V, RPT, NT>(ell_col + i + j * pitch, ell_val + i + j * pitch, c[j], v[j]);
        V xv[W][RPT];
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
            for (int r = 0; r < RPT; ++r) xv[j][r] = (c[j][r] != -1) ? x[c[j][r]] : V(0);
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
            for (int r = 0; r < RPT; ++r) if (c[j][r] != -1) sum[r] += v[j][r] * xv[j][r];
    } else {
        for (int j = 0; j < ell_w; ++j) {
            int c[RPT]; V v[RPT];
            ell_load<V, RPT, NT>(ell_col + i + j * pitch, ell_val + i + j * pitch, c, v);
#pragma unroll
            for (int r = 0; r < RPT; ++r) if (c[r] != -1) sum[r] += v[r] * x[c[r]];
        }
    }

    if (csr_ptr) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if (i + r < n)
                for (int j = csr_ptr[i + r], e = csr_ptr[i + r + 1]; j < e; ++j)
                    sum[r] += csr_val[j] * x[csr_col[j]];
        }
    }

#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        if (i + r < n) {
            V o = alpha * sum[r];
            if (append) o = y[i + r] + o;
            y[i + r] = o;
        }
    }
}

template <typename V>
__global__ __launch_bounds__(256)
void zero_kernel(long long n, V *y) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) y[i] = V(0);
}

template <typename V, typename I>
__global__ __launch_bounds__(256)
void gather_kernel(long long n, const I *__restrict__ idx, const V *__restrict__ src, V *__restrict__ dst) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}

// variant word (sweep tool / bench only): bit0 = nontemporal matrix streams,
// bit1 = XCD-contiguous block order, bits 2..3 = log2(rows per lane) for HELL.
int g_csr_variant = -1;     // -1: built-in default
int g_hell_variant = -1;
constexpr int kCsrDefault  = 0;
constexpr int kHellDefault = 7;   // RPT = 2, nontemporal streams, XCD-contiguous rows (sweep: profiles/)

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename V, typename I>
int spmv_csr(int dev, void *stream, int64_t n, V alpha, int append,
        const I *ptr, const I *col, const V *val, const V *x, V *y)
{
    VEXHIP_REQUIRE(n >= 0, "negative row count");
    if (n == 0) return 0;
    VEXHIP_REQUIRE(ptr && x && y, "NULL argument");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    if (!aligned16(col) || !aligned16(val)) {
        int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)info(dev).cus * 32);
        csr_scalar_kernel<V, I><<<grid, 256, 0, s>>>(n, alpha, append, ptr, col, val, x, y);
        VEXHIP_LAUNCH_CHECK();
        return 0;
    }
    long long nb = (n + CSR_BLOCK - 1) / CSR_BLOCK;
    int variant = g_csr_variant < 0 ? kCsrDefault : g_csr_variant;
    bool nt = variant & 1, swz = variant & 2;
    long long grid = swz ? ((nb + 7) / 8) * 8 : nb;
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
#define LAUNCH(NT, SWZ) csr_stream_kernel<V, I, NT, SWZ><<<(unsigned)grid, CSR_BLOCK, 0, s>>>( \
        n, nb, alpha, append, ptr, col, val, x, y)
    if (nt) { if (swz) LAUNCH(true, true); else LAUNCH(true, false); }
    else    { if (swz) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

template <typename V, int RPT, bool NT, bool SWZ>
int launch_hell_w(hipStream_t s, long long grid, long long nb, int64_t n, V alpha, int append,
        int w, int64_t pitch, const int *ec, const V *ev,
        const int *cp, const int *cc, const V *cv, const V *x, V *y)
{
#define CASE(W) case W: hell_kernel<V, RPT, W, NT, SWZ><<<(unsigned)grid, 256, 0, s>>>( \
        n, nb, alpha, append, w, pitch, ec, ev, cp, cc, cv, x, y); break;
    switch (w) {
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
        default: hell_kernel<V, RPT, 0, NT, SWZ><<<(unsigned)grid, 256, 0, s>>>(
                n, nb, alpha, append, w, pitch, ec, ev, cp, cc, cv, x, y);
    }
#undef CASE
    return 0;
}

template <typename V>
int spmv_hell(int dev, void *stream, int64_t n, V alpha, int append,
        int64_t w, int64_t pitch, const int *ec, const V *ev,
        const int *cp, const int *cc, const V *cv, const V *x, V *y)
{
    VEXHIP_REQUIRE(n >= 0 && w >= 0, "negative size");
    if (n == 0) return 0;
    VEXHIP_REQUIRE(x && y, "NULL argument");
    VEXHIP_REQUIRE(w == 0 || (ec && ev), "ELL arrays are NULL");
    VEXHIP_REQUIRE(w == 0 || pitch >= n, "ELL pitch smaller than row count");
    VEXHIP_REQUIRE(w < (1 << 30), "ELL width too large");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);

    if (w == 0 && !cp) {                 // empty matrix part: csr.inl:196-199
        if (!append) {
            int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)info(dev).cus * 16);
            zero_kernel<V><<<grid, 256, 0, s>>>(n, y);
            VEXHIP_LAUNCH_CHECK();
        }
        return 0;
    }

    int variant = g_hell_variant < 0 ? kHellDefault : g_hell_variant;
    bool nt = variant & 1, swz = variant & 2;
    int rpt = 1 << ((variant >> 2) & 3);
    // vector loads need pitch % 16 == 0 (the reference's alignup(n,16)) and
    // 16-byte aligned arrays; otherwise one row per lane.
    if (rpt > 1 && w > 0 && ((pitch % 16) != 0 || !aligned16(ec) || !aligned16(ev))) rpt = 1;
    if (rpt == 3 || rpt > 4) rpt = 4;

    long long nb = (n + (long long)256 * rpt - 1) / ((long long)256 * rpt);
    long long grid = swz ? ((nb + 7) / 8) * 8 : nb;
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");

#define GO(RPT, NT, SWZ) launch_hell_w<V, RPT, NT, SWZ>(s, grid, nb, n, alpha, append, (int)w, pitch, ec, ev, cp, cc, cv, x, y)
#define GO_RPT(RPT) do { \
        if (nt) { if (swz) GO(RPT, true, true); else GO(RPT, true, false); } \
        else    { if (swz) GO(RPT, false, true); else GO(RPT, false, false); } } while (0)
    if (rpt == 1) GO_RPT(1); else if (rpt == 2) GO_RPT(2); else GO_RPT(4);
#undef GO_RPT
#undef GO
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

} // namespace
} // namespace vexhip

using namespace vexhip;

extern "C" {

int vexhip_spmv_csr_set_variant(int variant) { g_csr_variant = variant; return 0; }
int vexhip_spmv_hell_set_variant(int variant) { g_hell_variant = variant; return 0; }

int vexhip_spmv_csr_f64_i32(int dev, void *stream, int64_t n, double alpha, int append,
        const int32_t *ptr, const int32_t *col, const double *val, const double *x, double *y)
{ return spmv_csr<double, int>(dev, stream, n, alpha, append, ptr, col, val, x, y); }

int vexhip_spmv_csr_f32_i32(int dev, void *stream, int64_t n, float alpha, int append,
        const int32_t *ptr, const int32_t *col, const float *val, const float *x, float *y)
{ return spmv_csr<float, int>(dev, stream, n, alpha, append, ptr, col, val, x, y); }

int vexhip_spmv_csr_f64_i64(int dev, void *stream, int64_t n, double alpha, int append,
        const int64_t *ptr, const int64_t *col, const double *val, const double *x, double *y)
{
    return spmv_csr<double, long long>(dev, stream, n, alpha, append,
            reinterpret_cast<const long long *>(ptr), reinterpret_cast<const long long *>(col), val, x, y);
}

int vexhip_spmv_hell_f64_i32(int dev, void *stream, int64_t n, double alpha, int append,
        int64_t w, int64_t pitch, const int32_t *ec, const double *ev,
        const int32_t *cp, const int32_t *cc, const double *cv, const double *x, double *y)
{ return spmv_hell<double>(dev, stream, n, alpha, append, w, pitch, ec, ev, cp, cc, cv, x, y); }

int vexhip_spmv_hell_f32_i32(int dev, void *stream, int64_t n, float alpha, int append,
        int64_t w, int64_t pitch, const int32_t *ec, const float *ev,
        const int32_t *cp, const int32_t *cc, const float *cv, const float *x, float *y)
{ return spmv_hell<float>(dev, stream, n, alpha, append, w, pitch, ec, ev, cp, cc, cv, x, y); }

int vexhip_gather_f64_i32(int dev, void *stream, int64_t n, const int32_t *idx, const double *src, double *dst) {
    if (n <= 0) return 0;
    VEXHIP_SET_DEVICE(dev);
    int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)info(dev).cus * 16);
    gather_kernel<double, int><<<grid, 256, 0, as_stream(stream)>>>(n, idx, src, dst);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

int vexhip_gather_f32_i32(int dev, void *stream, int64_t n, const int32_t *idx, const float *src, float *dst) {
    if (n <= 0) return 0;
    VEXHIP_SET_DEVICE(dev);
    int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)info(dev).cus * 16);
    gather_kernel<float, int><<<grid, 256, 0, as_stream(stream)>>>(n, idx, src, dst);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

} // extern "C"
