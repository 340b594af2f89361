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
#include "traversal.hpp"
#include "pairing.hpp"

#include <algorithm>
#include <climits>
#include <cstring>
#include <vector>

namespace vexhip {
extern int g_sell8_variant;                                   // sell8.hip: 0 = pair kernels (default), 1 = one gather per entry
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
        const V *__restrict__ x, V *__restrict__ y, trav_dev trav)
{
    __shared__ V s_prod[CSR_TILE];
    __shared__ I s_ptr[CSR_BLOCK + 1];

    // banded / stencil matrices: strip traversal in units of 256-row blocks (vexhip_csr_traversal_i32)
    const long long lb = (trav.chunk > 0 || trav.order) ? traversal_block(trav, nblocks) : logical_block<SWZ>(nblocks);
    if (lb < 0 || lb >= nblocks) return;
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
        const long long lo64 = (my_lo > tb ? my_lo : tb) - tb, hi64 = (my_hi < te ? my_hi : te) - tb;       // in 64 bits: see csr_stream2_kernel
        const int lo = (int)(lo64 < CSR_TILE ? lo64 : CSR_TILE);
        const int hi = (int)(hi64 < 0 ? 0 : hi64);
        for (int j = lo; j < hi; ++j) sum += s_prod[j];
        if (te < end) __syncthreads();
    }

    // "+=" on an empty row leaves y untouched: no read-modify-write traffic for
    // the mostly-empty remote part of a partitioned matrix
    const V *zs = static_cast<const V *>(trav.z);             // != NULL: y = alpha A x + beta zs (round 6; every row is written)
    if (t < rows_here && !(append && !zs && my_lo == my_hi)) {
        V r = alpha * sum;
        if (zs) r = (V)trav.beta * zs[r0 + t] + r;
        else if (append) r = y[r0 + t] + r;
        y[r0 + t] = r;
    }
}

// Second form (round 2, the default; bit 2 of the variant word selects the first form for A/B).
// The first form gathers x in STREAM order: a wave instruction covers 256 consecutive entries, i.e. ~37 rows x 7
// different diagonals -- ~30 cache lines per gather instruction, 529 L1 accesses per 64 rows against 478 per 128 rows
// for SELL (profiles/r02_sq_summary.txt): the kernel is bound by the L1 / address path, not by HBM (61 %).  Removing
// its barrier and LDS row-pointer table changed nothing (tools/r02_csr_ab.py).  Here the stream is only STAGED in
// stream order -- (col, val) with coalesced 16-byte loads into LDS -- and then every lane walks ITS OWN ROW: in step
// j the lanes of a wave gather x[col(row + lane, j)], which for a banded matrix are adjacent elements (8-9 lines per
// instruction, the ELL pattern), up to eight entries in flight per lane.  The row is folded in the same loop: no
// product staging.  Same products, same order: bit-identical.  24 KiB of LDS per workgroup (6 workgroups per CU).
// 512^3, same box: first form 2.86 ms, this one 2.48 ms (70 % of 8 TB/s by CSR bytes); of that, 2.75 -> 2.56 ms came from
// issuing all loads of a tile before the first LDS write (see the staging loop).  Tried and dropped
// (tools/r02_csr_ab.py, profiles/r02_csr_ab.json): half the lanes folding two rows each with 16-byte pair gathers (3.42 ms:
// the fold becomes the serial part of every workgroup); tiles of 1856 / 1920 entries for 7 instead of 6 workgroups per CU
// (no change) and 3072 (4 per CU: 2.65 ms); a workgroup that owns G consecutive row blocks and keeps the next tile's
// loads in flight while it folds the current one (G = 1 2.56, 2 2.58, 3 2.67, 4 2.69 ms: a workgroup's span of the
// traversal grows with G and the x planes fall out of its XCD's L2).
template <typename V, typename I, bool SWZ, int CSR2_TILE, typename P = I>
__global__ __launch_bounds__(CSR_BLOCK)
void csr_stream2_kernel(long long n, long long nblocks, V alpha, int append,
        const P *__restrict__ ptr, const I *__restrict__ col, const V *__restrict__ val,
        const V *__restrict__ x, V *__restrict__ y, trav_dev trav)
{
    constexpr int CSR2_GROUPS = (CSR2_TILE + 4 * CSR_BLOCK - 1) / (4 * CSR_BLOCK);
    __shared__ V s_val[CSR2_TILE];
    __shared__ I s_col[CSR2_TILE];
    const long long lb = (trav.chunk > 0 || trav.order) ? traversal_block(trav, nblocks) : logical_block<SWZ>(nblocks);
    if (lb < 0 || lb >= nblocks) return;
    const int t = threadIdx.x;
    const long long r0 = lb * CSR_BLOCK;
    const long long rows_here = (n - r0 < CSR_BLOCK) ? (n - r0) : CSR_BLOCK;
    const long long base = ptr[r0], end = ptr[r0 + rows_here];            // uniform addresses: scalar loads
    // this lane's row bounds: loaded unconditionally from a clamped row (a load under `if (t < rows_here)` is an exec-
    // masked block that ends in a full vmcnt(0) wait -- one more round trip before the tile loads can be issued)
    // -- and used only behind the barrier (the empty asm keeps the compiler from waiting for them any earlier)
    const long long my_row = t < rows_here ? r0 + t : r0 + rows_here - 1;
    P raw_lo = ptr[my_row], raw_hi = ptr[my_row + 1];

    V sum = 0;
    for (long long tb = base & ~3ll; tb < end; tb += CSR2_TILE) {
        const long long te = (end - tb > CSR2_TILE) ? tb + CSR2_TILE : end;
        // stage the tile: 2 groups of 4 entries per lane, whole groups with 16-byte loads.  ALL loads of the tile are
        // issued before the first LDS write (a loop that loads and writes group after group costs the workgroup one
        // more memory round trip: 2.75 -> 2.56 ms at 512^3 on the same box, tools/r02_csr_ab.py)
        I pc[CSR2_GROUPS][4]; V pv[CSR2_GROUPS][4];
#pragma unroll
        for (int u = 0; u < CSR2_GROUPS; ++u) {
            const long long k = tb + 4 * t + u * (4 * CSR_BLOCK);
            if (te - k >= 4) { load4<false>(col + k, pc[u]); load4<false>(val + k, pv[u]); }
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pc[u][q] = 0; pv[u][q] = V(0);
                    if (k + q < te) { pc[u][q] = col[k + q]; pv[u][q] = val[k + q]; }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < CSR2_GROUPS; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((u + 1) * (4 * CSR_BLOCK) <= CSR2_TILE || 4 * t + u * (4 * CSR_BLOCK) + q < CSR2_TILE) {
                    s_col[4 * t + u * (4 * CSR_BLOCK) + q] = pc[u][q];
                    s_val[4 * t + u * (4 * CSR_BLOCK) + q] = pv[u][q];
                }
        __syncthreads();
        // every lane folds the part of its row that lies in this tile, up to 8 entries in flight
        asm volatile("" : "+v"(raw_lo), "+v"(raw_hi));
        // A lane without a row (ragged last workgroup) folds nothing: its bounds are the tile's start.  (Round 3 gave such lanes
        // the bounds 0, 0: hi = (int)(0 - tb) -- negative, no trip, while tb < 2^31; with 64-bit row pointers and tb >= 2^31 the
        // cast wraps to a large POSITIVE count: 2^28 trips through LDS far beyond the tile and gathers of x at whatever lies
        // there -- the 120 s / the fault that round 3 recorded as "not understood" at 700^3 and on pointers offset past 2^31.)
        const long long my_lo = t < rows_here ? (long long)raw_lo : tb, my_hi = t < rows_here ? (long long)raw_hi : tb;
        const long long lo64 = (my_lo > tb ? my_lo : tb) - tb, hi64 = (my_hi < te ? my_hi : te) - tb;
        const int lo = (int)(lo64 < CSR2_TILE ? lo64 : CSR2_TILE);
        const int hi = (int)(hi64 < 0 ? 0 : hi64);
        for (int j = lo; j < hi; j += 8) {
            I c[8]; V v[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = j + u < hi ? j + u : hi - 1;               // clamped: a harmless re-read of the last entry
                c[u] = s_col[p]; v[u] = s_val[p];
            }
            // a gather no lane of the wave needs is not issued (7-entry rows: the eighth): these kernels are bound by
            // the number of vector-memory instructions
#pragma unroll
            for (int u = 0; u < 8; ++u) { xv[u] = V(0); if (j + u < hi) xv[u] = x[c[u]]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const V added = sum + v[u] * xv[u];
                sum = (j + u < hi) ? added : sum;
            }
        }
        if (te < end) __syncthreads();
    }
    const V *zs = static_cast<const V *>(trav.z);             // != NULL: y = alpha A x + beta zs (round 6; every row is written)
    if (t < rows_here && !(append && !zs && raw_lo == raw_hi)) {
        V r = alpha * sum;
        if (zs) r = (V)trav.beta * zs[r0 + t] + r;
        else if (append) r = y[r0 + t] + r;
        y[r0 + t] = r;
    }
}

// Row-subset CSR, always "+=":  y[rows[k]] += alpha * sum_j val[j] * x[col[j]],  j in [ptr[k], ptr[k+1]).
// The remote part of a partitioned matrix touches only the rows next to a partition boundary
// (2 x 260 100 of 16.8 M rows per GPU for 512^3 over 8 GPUs): the product then reads a row list
// and those rows' entries instead of the row pointers of every row.
template <typename V>
__global__ __launch_bounds__(256)
void csr_rows_kernel(long long nr, V alpha, const int *__restrict__ rows, const int *__restrict__ ptr,
        const int *__restrict__ col, const V *__restrict__ val, const V *__restrict__ x, V *__restrict__ y)
{
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nr; k += (long long)gridDim.x * blockDim.x) {
        V sum = 0;
        for (int j = ptr[k], e = ptr[k + 1]; j < e; ++j) sum += val[j] * x[col[j]];
        const int r = rows[k];
        y[r] = y[r] + alpha * sum;
    }
}

// Fallback for CSR arrays that are not 16-byte aligned (sub-views): the
// reference's one-row-per-work-item loop, unchanged.
template <typename V, typename I, typename P = I>
__global__ __launch_bounds__(256)
void csr_scalar_kernel(long long n, V alpha, int append,
        const P *__restrict__ ptr, const I *__restrict__ col, const V *__restrict__ val,
        const V *__restrict__ x, V *__restrict__ y, const V *zs = nullptr, V beta = V(0))
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        V sum = 0;
        for (P j = ptr[i], e = ptr[i + 1]; j < e; ++j) sum += val[j] * x[col[j]];
        V r = alpha * sum;
        if (zs) r = beta * zs[i] + r;
        else if (append) r = y[i] + r;
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
// Round 3: pair gathers as in the SELL products (one 16-byte load of x where rows 2t and 2t + 1 hold columns c and c + 1)
// were tried here and dropped -- 2.586 against 2.534 ms at 512^3 (profiles/r03_hell_ab.json): this layout is bound by its
// 14 column-plane streams (13.4 GB at 5.3 TB/s), not by its gathers; the slice-local SELL layout moves the same bytes in
// 2.2-2.3 ms, and no default path uses this kernel (vex::SpMat and vexhip_spmat pick SELL storages).
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
        const V *__restrict__ x, V *__restrict__ y, trav_dev trav)
{
    // a traversal (vexhip_hell_order_i32) walks the rows in L2-sized strips; any permutation is correct
    long long lb;
    if (trav.order || trav.chunk > 0) { lb = traversal_block(trav, nblocks); if (lb < 0) return; }
    else { lb = logical_block<SWZ>(nblocks); if (lb >= nblocks) return; }
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

// ---------------------------------------------------------------------------
// Sliced ELL (SELL-512): the ELL part stored slice-major, one slice = the 512
// rows of one workgroup; a slice is ONE contiguous region of w*512*(4+sizeof(V))
// bytes: its w*512 columns (element (r, j) at j*512 + r) followed by its w*512
// values.  A workgroup then streams 42 KiB (w = 7, fp64) front to back instead of
// 2*w segments that lie pitch*4 / pitch*8 bytes apart: fewer DRAM pages and TLB
// entries per workgroup, and the column / value streams never collide on the same
// memory channels.  Same arithmetic, same order.
// ---------------------------------------------------------------------------
constexpr int SELL_ROWS = 512;

// XNT (A/B, round 6, VEXHIP_SELL_XLOAD=nt): the gathers of x as non-temporal loads -- does the memory system move less than a whole line
// per 8-byte gather then?  (profiles/r06_unstructured_counters.log: it does not; TCC_EA0_RDREQ_32B stays 0)
template <typename V, int W, bool NT, bool XNT = false>
__global__ __launch_bounds__(256)
void sell_kernel(long long n, long long nslices, V alpha, int append, int ell_w,
        const char *__restrict__ sell,
        const int *__restrict__ csr_ptr, const int *__restrict__ csr_col, const V *__restrict__ csr_val,
        const V *__restrict__ x, V *__restrict__ y, trav_dev trav)
{
    const long long s = traversal_block(trav, nslices);
    if (s < 0) return;
    const int r = 2 * threadIdx.x;
    const long long i = s * SELL_ROWS + r;
    const int w = W > 0 ? W : ell_w;
    // one contiguous region per slice: w*512 columns, then w*512 values
    const char *slice = sell + s * ((long long)w * SELL_ROWS * (4 + (long long)sizeof(V)));
    const int *cp = reinterpret_cast<const int *>(slice) + r;
    const V *vp = reinterpret_cast<const V *>(slice + (long long)w * SELL_ROWS * 4) + r;

    V sum[2] = {V(0), V(0)};
    if constexpr (W > 0) {
        int c[W][2]; V v[W][2];
#pragma unroll
        for (int j = 0; j < W; ++j) ell_load<V, 2, NT>(cp + j * SELL_ROWS, vp + j * SELL_ROWS, c[j], v[j]);
        V xv[W][2];
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) xv[j][q] = (c[j][q] >= 0) ? x[c[j][q]] : V(0);
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) if (c[j][q] >= 0) sum[q] += v[j][q] * xv[j][q];
    } else {
        for (int j = 0; j < w; ++j) {
            int c[2]; V v[2];
            ell_load<V, 2, NT>(cp + j * SELL_ROWS, vp + j * SELL_ROWS, c, v);
#pragma unroll
            for (int q = 0; q < 2; ++q) if (c[q] >= 0) sum[q] += v[q] * (XNT ? __builtin_nontemporal_load(x + c[q]) : x[c[q]]);
        }
    }
    if (csr_ptr) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (i + q < n)
                for (int j = csr_ptr[i + q], e = csr_ptr[i + q + 1]; j < e; ++j) sum[q] += csr_val[j] * x[csr_col[j]];
    }
    store_pair<V>(n, i, alpha, append, sum, y, trav);
}

// Pair form of sell_kernel (the default for W <= 8; reasoning and measurements: sell8.hip, "PAIR kernels").
// The two rows of a lane are aligned by diagonal at fill time, so in almost every ELL column their columns are
// c and c + 1: ONE 16-byte load of x instead of two 8-byte gathers -- these products are bound by the number of
// vector-memory instructions.  Padding is -1 ("the partner's 16-byte load may cover me") or -2 (it would leave
// x); lanes whose two columns are not adjacent take an 8-byte load per real entry (rare branch).  Gathered values
// of padding entries are replaced by 0 and their stored value is 0: adding +-0 leaves the sum bit-identical.
template <typename V, int W>
__global__ __launch_bounds__(256)
void sell_pair_kernel(long long n, long long nslices, V alpha, int append,
        const char *__restrict__ sell,
        const int *__restrict__ csr_ptr, const int *__restrict__ csr_col, const V *__restrict__ csr_val,
        const V *__restrict__ x, V *__restrict__ y, trav_dev trav)
{
    constexpr long long SLICE = (long long)W * SELL_ROWS * (4 + (long long)sizeof(V));
    typedef V V2 __attribute__((ext_vector_type(2)));
    const long long s = traversal_block(trav, nslices);
    if (s < 0) return;
    const int r = 2 * threadIdx.x;
    const long long i = s * SELL_ROWS + r;
    const int *cp = reinterpret_cast<const int *>(sell + s * SLICE) + r;
    const V *vp = reinterpret_cast<const V *>(sell + s * SLICE + (long long)W * SELL_ROWS * 4) + r;
    int c[W][2]; V v[W][2];
#pragma unroll
    for (int j = 0; j < W; ++j) ell_load<V, 2, true>(cp + j * SELL_ROWS, vp + j * SELL_ROWS, c[j], v[j]);
    V xv[W][2];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int c0 = c[j][0], c1 = c[j][1];
        const bool m0 = c0 >= 0, m1 = c1 >= 0;
        const bool pair = (m0 && m1) ? (c1 == c0 + 1) : (m0 ? c1 == -1 : (m1 ? c0 == -1 : true));
        // the 16-byte load is skipped only when NO lane of the wave has a pair in this column (matrices without
        // band structure); lanes without one read their own stored values instead
        const bool use16 = pair && (m0 || m1);
        V2 p = {V(0), V(0)};
        if (__builtin_amdgcn_ballot_w64(use16) != 0) {
            const V *px = use16 ? x + (m0 ? (long long)c0 : (long long)c1 - 1) : vp;
            __builtin_memcpy(&p, px, sizeof(V2));
        }
        xv[j][0] = p.x; xv[j][1] = p.y;
        if (!pair) {
            if (m0) xv[j][0] = x[c0];
            if (m1) xv[j][1] = x[c1];
        }
        xv[j][0] = m0 ? xv[j][0] : V(0);
        xv[j][1] = m1 ? xv[j][1] : V(0);
    }
    V sum[2] = {V(0), V(0)};
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) sum[q] += v[j][q] * xv[j][q];
    if (csr_ptr) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (i + q < n)
                for (int j = csr_ptr[i + q], e = csr_ptr[i + q + 1]; j < e; ++j) sum[q] += csr_val[j] * x[csr_col[j]];
    }
    store_pair<V>(n, i, alpha, append, sum, y, trav);
}

// One lane per row PAIR: the entries of rows 2t and 2t+1 are aligned by diagonal (pairing.hpp); the empty half of
// a column is -1 when the partner's 16-byte load stays inside x (max_col: largest column of the ELL part), else -2.
template <typename V, typename P>
__global__ __launch_bounds__(256)
void sell_fill_kernel(long long n, long long nslices, int w,
        const P *__restrict__ ptr, const int *__restrict__ col, const V *__restrict__ val,
        const int *__restrict__ max_col_p, char *__restrict__ sell)
{
    const int max_col = *max_col_p;
    for (long long pr = (long long)blockIdx.x * blockDim.x + threadIdx.x; pr < nslices * (SELL_ROWS / 2);
         pr += (long long)gridDim.x * blockDim.x) {
        const long long i = 2 * pr;
        long long b[2] = {0, 0}, e[2] = {0, 0};
        for (int q = 0; q < 2; ++q) if (i + q < n) { b[q] = ptr[i + q]; e[q] = ptr[i + q + 1]; }
        char *slice = sell + (i / SELL_ROWS) * ((long long)w * SELL_ROWS * (4 + (long long)sizeof(V)));
        int *sc = reinterpret_cast<int *>(slice) + (i % SELL_ROWS);
        V *sv = reinterpret_cast<V *>(slice + (long long)w * SELL_ROWS * 4) + (i % SELL_ROWS);
        pair_walk pw;
        pw.init(col, i, b[0], (int)min(e[0] - b[0], (long long)w), b[1], (int)min(e[1] - b[1], (long long)w), w);
        for (int j = 0; j < w; ++j) {
            long long en[2];
            pw.next(en[0], en[1]);
            for (int q = 0; q < 2; ++q) {
                int c; V v = V(0);
                if (en[q] >= 0) { c = col[en[q]]; v = val[en[q]]; }
                else if (en[1 - q] < 0) c = -1;
                else {
                    const long long first = (long long)col[en[1 - q]] - (1 - q);     // where the partner's 16-byte load starts
                    c = (first >= 0 && first + 1 <= max_col) ? -1 : -2;
                }
                sc[j * SELL_ROWS + q] = c;
                sv[j * SELL_ROWS + q] = v;
            }
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
constexpr int kHellOrderRows = 512; // rows per workgroup of the ordered product (RPT = 2)
constexpr int kHellDefault = 7;   // RPT = 2, nontemporal streams, XCD-contiguous rows (sweep: profiles/)

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename V, typename I>
int spmv_csr(int dev, void *stream, int64_t n, V alpha, int append,
        const I *ptr, const I *col, const V *val, const V *x, V *y, const vexhip_traversal *tr = nullptr)
{
    VEXHIP_REQUIRE(n >= 0, "negative row count");
    if (n == 0) return 0;
    VEXHIP_REQUIRE(ptr && x && y, "NULL argument");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    if (!aligned16(col) || !aligned16(val)) {
        int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)info(dev).cus * 32);
        const trav_dev ad = with_addend(trav_dev{nullptr, 0, 0, 0});
        csr_scalar_kernel<V, I><<<grid, 256, 0, s>>>(n, alpha, append, ptr, col, val, x, y, static_cast<const V *>(ad.z), (V)ad.beta);
        VEXHIP_LAUNCH_CHECK();
        return 0;
    }
    long long nb = (n + CSR_BLOCK - 1) / CSR_BLOCK;
    int variant = g_csr_variant < 0 ? kCsrDefault : g_csr_variant;
    bool nt = variant & 1, swz = variant & 2;
    long long grid = swz ? ((nb + 7) / 8) * 8 : nb;
    trav_dev order = {nullptr, 0, 0, 0};
    if (tr && tr->grid_blocks > 0) order = make_traversal(tr, nb, &grid);
    order = with_addend(order);                                    // y = alpha A x + beta z (vexhip_spmat_apply_axpby_*)
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
#define LAUNCH(NT, SWZ) csr_stream_kernel<V, I, NT, SWZ><<<(unsigned)grid, CSR_BLOCK, 0, s>>>( \
        n, nb, alpha, append, ptr, col, val, x, y, order)
    if (!(variant & 4)) {             // second form (default)
        if (swz) csr_stream2_kernel<V, I, true, 2048><<<(unsigned)grid, CSR_BLOCK, 0, s>>>(n, nb, alpha, append, ptr, col, val, x, y, order);
        else csr_stream2_kernel<V, I, false, 2048><<<(unsigned)grid, CSR_BLOCK, 0, s>>>(n, nb, alpha, append, ptr, col, val, x, y, order);
    }
    else if (nt) { if (swz) LAUNCH(true, true); else LAUNCH(true, false); }
    else    { if (swz) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

template <typename V>
int spmv_csr_rows(int dev, void *stream, int64_t nr, V alpha, const int *rows, const int *ptr, const int *col,
        const V *val, const V *x, V *y)
{
    VEXHIP_REQUIRE(nr >= 0, "negative row count");
    if (nr == 0) return 0;
    VEXHIP_REQUIRE(rows && ptr && x && y, "NULL argument");
    VEXHIP_SET_DEVICE(dev);
    const int grid = (int)std::min<int64_t>((nr + 255) / 256, (int64_t)info(dev).cus * 32);
    csr_rows_kernel<V><<<grid, 256, 0, as_stream(stream)>>>(nr, alpha, rows, ptr, col, val, x, y);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

template <typename V, int RPT, bool NT, bool SWZ>
int launch_hell_w(hipStream_t s, long long grid, long long nb, int64_t n, V alpha, int append,
        int w, int64_t pitch, const int *ec, const V *ev,
        const int *cp, const int *cc, const V *cv, const V *x, V *y, trav_dev order)
{
#define CASE(W) case W: hell_kernel<V, RPT, W, NT, SWZ><<<(unsigned)grid, 256, 0, s>>>( \
        n, nb, alpha, append, w, pitch, ec, ev, cp, cc, cv, x, y, order); break;
    switch (w) {
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
        default: hell_kernel<V, RPT, 0, NT, SWZ><<<(unsigned)grid, 256, 0, s>>>(
                n, nb, alpha, append, w, pitch, ec, ev, cp, cc, cv, x, y, order);
    }
#undef CASE
    return 0;
}

template <typename V>
int spmv_hell(int dev, void *stream, int64_t n, V alpha, int append,
        int64_t w, int64_t pitch, const int *ec, const V *ev,
        const int *cp, const int *cc, const V *cv, const V *x, V *y,
        const vexhip_traversal *tr = nullptr)
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

    trav_dev order = {nullptr, 0, 0, 0};
    const bool ordered = tr && tr->grid_blocks > 0;
    if (ordered) {
        // the traversal was built for kHellOrderRows rows per workgroup
        VEXHIP_REQUIRE(w > 0 && (pitch % 16) == 0 && aligned16(ec) && aligned16(ev), "ordered HELL product needs aligned ELL arrays");
        rpt = kHellOrderRows / 256;
        nt = true;
        order = trav_dev{tr->order, (int)tr->chunk, (int)tr->planes, (int)tr->plane_blocks};
    }

    long long nb = (n + (long long)256 * rpt - 1) / ((long long)256 * rpt);
    long long grid = ordered ? tr->grid_blocks : (swz ? ((nb + 7) / 8) * 8 : nb);
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");

#define GO(RPT, NT, SWZ) launch_hell_w<V, RPT, NT, SWZ>(s, grid, nb, n, alpha, append, (int)w, pitch, ec, ev, cp, cc, cv, x, y, order)
#define GO_RPT(RPT) do { \
        if (nt) { if (swz) GO(RPT, true, true); else GO(RPT, true, false); } \
        else    { if (swz) GO(RPT, false, true); else GO(RPT, false, false); } } while (0)
    if (rpt == 1) GO_RPT(1); else if (rpt == 2) GO_RPT(2); else GO_RPT(4);
#undef GO_RPT
#undef GO
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

template <typename V>
int spmv_sell(int dev, void *stream, int64_t n, V alpha, int append, int64_t w,
        const void *sell, const int *cp, const int *cc, const V *cv, const V *x, V *y,
        const vexhip_traversal *tr)
{
    VEXHIP_REQUIRE(n >= 0 && w >= 1 && w < (1 << 20), "bad SELL geometry");
    if (n == 0) return 0;
    VEXHIP_REQUIRE(sell && x && y && aligned16(sell), "the SELL buffer must be 16-byte aligned");
    const char *sc = static_cast<const char *>(sell);
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    long long ns = (n + SELL_ROWS - 1) / SELL_ROWS;
    const bool ordered = tr && tr->grid_blocks > 0;
    long long grid = ordered ? tr->grid_blocks : ns;
    trav_dev order = {nullptr, 0, 0, 0};
    if (ordered) order = trav_dev{tr->order, (int)tr->chunk, (int)tr->planes, (int)tr->plane_blocks};
    order = with_addend(order);                                    // y = alpha A x + beta z (vexhip_spmat_apply_axpby_*): the kernels' store_pair adds it
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
#define CASE(W) case W: if (g_sell8_variant == 0) sell_pair_kernel<V, W><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, sc, cp, cc, cv, x, y, order); \
        else sell_kernel<V, W, true><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, (int)w, sc, cp, cc, cv, x, y, order); break;
    switch (w) {
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
        default:
            if (env(ENV_VEXHIP_SELL_XLOAD)) sell_kernel<V, 0, true, true><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, (int)w, sc, cp, cc, cv, x, y, order);
            else sell_kernel<V, 0, true><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, (int)w, sc, cp, cc, cv, x, y, order);
    }
#undef CASE
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

template <typename V, typename P>
int sell_fill(int dev, void *stream, int64_t n, const P *ptr, const int *col, const V *val, int64_t w, void *sell) {
    VEXHIP_REQUIRE(n >= 0 && w >= 1, "bad SELL geometry");
    if (n == 0) return 0;
    VEXHIP_SET_DEVICE(dev);
    long long ns = (n + SELL_ROWS - 1) / SELL_ROWS;
    int grid = (int)std::min<int64_t>((ns * (SELL_ROWS / 2) + 255) / 256, (int64_t)info(dev).cus * 16);
    int *max_col = nullptr;                                   // largest column of the ELL part (bounds the 16-byte gathers)
    VEXHIP_TRY(hipMalloc(&max_col, sizeof(int)));
    hipError_t e = hipMemsetAsync(max_col, 0xff, sizeof(int), as_stream(stream));
    if (e == hipSuccess) {
        ell_max_col_kernel<P><<<std::max(1, (int)std::min<int64_t>((n + 255) / 256, (int64_t)info(dev).cus * 16)), 256, 0, as_stream(stream)>>>(
                n, (int)std::min<int64_t>(w, INT_MAX), ptr, col, max_col);
        sell_fill_kernel<V, P><<<std::max(1, grid), 256, 0, as_stream(stream)>>>(n, ns, (int)w, ptr, col, val, max_col, static_cast<char *>(sell));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(as_stream(stream));
    (void)hipFree(max_col);
    return check(e, __FILE__, __LINE__);
}

// Agreement of every ELL column j with a constant offset col - row == ref[j].
__global__ __launch_bounds__(256)
void ell_offset_agree_kernel(long long n, int w, long long pitch, const int *__restrict__ ell_col,
        const long long *__restrict__ ref, unsigned long long *__restrict__ agree)
{
    for (int j = 0; j < w; ++j) {
        unsigned long long local = 0;
        const long long off = ref[j];
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
             i += (long long)gridDim.x * blockDim.x) {
            // pitch < 0: SELL-512 storage with (-pitch)-byte values; index in ints from the buffer start
            long long e = pitch > 0 ? i + j * pitch
                                    : (i / SELL_ROWS) * ((long long)w * SELL_ROWS * (4 - pitch) / 4) + (long long)j * SELL_ROWS + i % SELL_ROWS;
            int c = ell_col[e];
            local += (c >= 0 && (long long)c - i == off) ? 1ull : 0ull;
        }
        for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
        if ((threadIdx.x & 63) == 0 && local) atomicAdd(&agree[j], local);
    }
}

} // namespace

// ---- 64-bit row pointers with 32-bit columns (round 3): internal entry points for spmat.hip ---------------------------
int sell_fill_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const double *val, int64_t w, void *sell)
{ return sell_fill<double, long long>(dev, stream, n, ptr, col, val, w, sell); }
int sell_fill_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const float *val, int64_t w, void *sell)
{ return sell_fill<float, long long>(dev, stream, n, ptr, col, val, w, sell); }

// The CSR arrays themselves with 64-bit row pointers (2^31 entries or more kept in CSR): the staged kernel with 64-bit row
// bounds, strips as for 32-bit pointers (spmat.hip passes the traversal).  Round 3 ran the reference's one-row-per-work-item
// loop here (spmat/csr.inl:153-171; 30 ms at 700^3) because this instantiation took 120 s and faulted: lanes without a row in
// the ragged last workgroup -- see the fold of csr_stream2_kernel.  Arrays that are not 16-byte aligned keep the plain loop.
template <typename V>
int spmv_csr_p64_impl(int dev, void *stream, int64_t n, V alpha, int append, const long long *ptr, const int32_t *col, const V *val, const V *x, V *y,
        const vexhip_traversal *tr) {
    VEXHIP_REQUIRE(n >= 0, "negative row count");
    if (n == 0) return 0;
    VEXHIP_REQUIRE(ptr && x && y, "NULL argument");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    if (!aligned16(col) || !aligned16(val) || g_csr_variant == 8) {
        const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)info(dev).cus * 32);
        const trav_dev ad = with_addend(trav_dev{nullptr, 0, 0, 0});
        csr_scalar_kernel<V, int, long long><<<grid, 256, 0, s>>>(n, alpha, append, ptr, col, val, x, y, static_cast<const V *>(ad.z), (V)ad.beta);
        VEXHIP_LAUNCH_CHECK();
        return 0;
    }
    const long long nb = (n + CSR_BLOCK - 1) / CSR_BLOCK;
    long long grid = ((nb + 7) / 8) * 8;                      // no strips: XCD-contiguous row blocks
    trav_dev order = {nullptr, 0, 0, 0};
    const bool strips = tr && tr->grid_blocks > 0;
    if (strips) order = make_traversal(tr, nb, &grid);
    order = with_addend(order);
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    if (strips) csr_stream2_kernel<V, int, false, 2048, long long><<<(unsigned)grid, CSR_BLOCK, 0, s>>>(n, nb, alpha, append, ptr, col, val, x, y, order);
    else csr_stream2_kernel<V, int, true, 2048, long long><<<(unsigned)grid, CSR_BLOCK, 0, s>>>(n, nb, alpha, append, ptr, col, val, x, y, order);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}
int spmv_csr_p64(int dev, void *stream, int64_t n, double alpha, int append, const long long *ptr, const int32_t *col, const double *val, const double *x, double *y, const vexhip_traversal *tr)
{ return spmv_csr_p64_impl<double>(dev, stream, n, alpha, append, ptr, col, val, x, y, tr); }
int spmv_csr_p64(int dev, void *stream, int64_t n, float alpha, int append, const long long *ptr, const int32_t *col, const float *val, const float *x, float *y, const vexhip_traversal *tr)
{ return spmv_csr_p64_impl<float>(dev, stream, n, alpha, append, ptr, col, val, x, y, tr); }

} // namespace vexhip

using namespace vexhip;

extern "C" {

// Workgroup -> row-block order for the HELL product of a banded / stencil matrix.
// If most rows reach the same far column offsets (+-S_big, e.g. +-n^2 for a 3-D
// stencil) the x values of three "planes" are live at once; walking whole planes
// needs 3*S_big*8 bytes of cache.  The order walks the rows in tiles of <= 64 Ki
// rows, plane after plane, inside each XCD's contiguous eighth of the matrix, so
// the live part of x (3 tiles) stays in that XCD's 4 MiB L2 and x is fetched from
// HBM about once instead of three times.  grid_blocks = 0: no reordering pays.
static int build_order(int dev, void *stream, int64_t n, int64_t w, int64_t pitch /* < 0: SELL-512, -pitch = value bytes */,
        const int32_t *ell_col, int mode, int32_t *order, int64_t capacity, vexhip_traversal *out)
{
    VEXHIP_REQUIRE(out, "NULL output");
    std::memset(out, 0, sizeof(*out));
    int64_t grid_value = 0, *grid_blocks = &grid_value;
    const int64_t rpb = kHellOrderRows;
    const int64_t tile_rows_max = 65536;                 // 3 tiles x 8 B = 1.5 MiB of x
    if (w < 1 || w > 32 || n < 8 * tile_rows_max || (pitch > 0 && (pitch % 16) != 0)) return 0;
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);

    // modal (col - row) offset of every ELL column over 64 scattered sample rows
    std::vector<long long> ref(w, 1ll << 62);
    {
        const int ns = 64;
        std::vector<int> sample((size_t)ns * w, -1);
        std::vector<int64_t> rows(ns);
        for (int k = 0; k < ns; ++k) {
            rows[k] = (int64_t)(((unsigned long long)(k + 1) * 0x9E3779B97F4A7C15ull) % (unsigned long long)n);
            for (int j = 0; j < w; ++j)
                VEXHIP_TRY(hipMemcpyAsync(&sample[(size_t)k * w + j],
                            ell_col + (pitch > 0 ? rows[k] + j * pitch
                                             : (rows[k] / SELL_ROWS) * (w * SELL_ROWS * (4 - pitch) / 4) + (int64_t)j * SELL_ROWS + rows[k] % SELL_ROWS),
                            sizeof(int), hipMemcpyDeviceToHost, s));
        }
        VEXHIP_TRY(hipStreamSynchronize(s));
        for (int j = 0; j < w; ++j) {
            std::vector<long long> offs;
            for (int k = 0; k < ns; ++k) if (sample[(size_t)k * w + j] >= 0) offs.push_back((long long)sample[(size_t)k * w + j] - rows[k]);
            std::sort(offs.begin(), offs.end());
            size_t best = 0;
            for (size_t a = 0; a < offs.size();) {
                size_t b = a; while (b < offs.size() && offs[b] == offs[a]) ++b;
                if (b - a > best) { best = b - a; ref[j] = offs[a]; }
                a = b;
            }
        }
    }
    long long *dref = nullptr;
    VEXHIP_TRY(hipMalloc(&dref, sizeof(long long) * 2 * w));
    unsigned long long *dagree = reinterpret_cast<unsigned long long *>(dref + w);
    VEXHIP_TRY(hipMemcpyAsync(dref, ref.data(), sizeof(long long) * w, hipMemcpyHostToDevice, s));
    VEXHIP_TRY(hipMemsetAsync(dagree, 0, sizeof(long long) * w, s));
    int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)info(dev).cus * 16);
    ell_offset_agree_kernel<<<grid, 256, 0, s>>>(n, (int)w, pitch, ell_col, dref, dagree);
    std::vector<unsigned long long> agree(w);
    VEXHIP_TRY(hipMemcpyAsync(agree.data(), dagree, sizeof(long long) * w, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    VEXHIP_TRY(hipFree(dref));

    int64_t s_big = 0, s_mid = 0;
    for (int j = 0; j < w; ++j) {
        if (agree[j] * 2 < (unsigned long long)n) continue;          // not a constant band
        int64_t a = ref[j] < 0 ? -ref[j] : ref[j];
        if (a > s_big) { if (s_big > s_mid) s_mid = s_big; s_big = a; }
        else if (a < s_big && a > s_mid) s_mid = a;
    }
    if (s_big < 2 * tile_rows_max || (s_big % rpb) != 0) return 0;   // planes already fit in L2, or misaligned
    const int64_t unit = (s_mid >= rpb && s_mid <= tile_rows_max && (s_mid % rpb) == 0) ? s_mid : rpb;
    const int64_t tile_blocks = (tile_rows_max / unit) * unit / rpb;
    const int64_t plane_blocks = s_big / rpb;
    const int64_t nb = (n + rpb - 1) / rpb;
    std::vector<int> host;
    int64_t g = 0;
    if (mode == 1) {
        // per-XCD slabs: XCD k walks its own contiguous eighth of the rows, tile by tile
        const int64_t per = (nb + 7) / 8;
        std::vector<std::vector<int>> slab(8);
        for (int k = 0; k < 8; ++k) {
            const int64_t b0 = k * per, b1 = std::min<int64_t>(nb, b0 + per);
            if (b0 >= b1) continue;
            const int64_t p0 = b0 / plane_blocks, p1 = (b1 - 1) / plane_blocks;
            for (int64_t t0 = 0; t0 < plane_blocks; t0 += tile_blocks)
                for (int64_t p = p0; p <= p1; ++p)
                    for (int64_t l = t0; l < std::min(plane_blocks, t0 + tile_blocks); ++l) {
                        int64_t b = p * plane_blocks + l;
                        if (b >= b0 && b < b1) slab[k].push_back((int)b);
                    }
        }
        size_t longest = 0;
        for (auto &v : slab) longest = std::max(longest, v.size());
        g = (int64_t)longest * 8;
        host.assign((size_t)g, -1);
        for (int k = 0; k < 8; ++k)
            for (size_t i = 0; i < slab[k].size(); ++i) host[i * 8 + k] = slab[k][i];   // workgroup b runs on XCD b % 8
    } else if (mode == 2) {
        // global tiles, consecutive row-blocks round-robin over the XCDs
        const int64_t tb = std::max<int64_t>(8, tile_blocks / 8 * 8);
        const int64_t planes = (nb + plane_blocks - 1) / plane_blocks;
        host.reserve((size_t)nb + 8);
        for (int64_t t0 = 0; t0 < plane_blocks; t0 += tb)
            for (int64_t p = 0; p < planes; ++p)
                for (int64_t l = t0; l < t0 + tb; ++l) {
                    int64_t b = p * plane_blocks + l;
                    host.push_back((l < plane_blocks && b < nb) ? (int)b : -1);
                }
        g = (int64_t)host.size();
    } else {
        // chunked tiles (default).  Per-XCD L2s are private, so BOTH kinds of reuse must stay
        // on one XCD: a tile of tb row-blocks is cut into 8 chunks of tb/8 consecutive
        // row-blocks, XCD k owns chunk k of the tile in every plane (neighbouring rows,
        // +-S_mid, share its L2) and sweeps plane after plane (+-S_big re-reads are tb/8
        // workgroups apart: still resident).  All XCDs stay within one tile of each other
        // in memory.  Workgroup b runs on XCD b % 8.
        // Default chunk: an eighth of a plane, at most 64 row-blocks (each XCD then owns a
        // strip of the plane through ALL planes); mode >= 100 forces chunk = mode - 100 (tuning).
        (void)tile_blocks;
        const int64_t chunk = mode >= 100 ? std::max<int64_t>(1, mode - 100)
                                          : std::max<int64_t>(1, std::min<int64_t>(64, plane_blocks / 8));
        // pure arithmetic in the kernel (traversal_block): no map, no dependent load
        const int64_t tb = 8 * chunk;
        const int64_t planes = (nb + plane_blocks - 1) / plane_blocks;
        const int64_t tiles = (plane_blocks + tb - 1) / tb;
        out->grid_blocks = tiles * planes * tb;
        out->chunk = chunk; out->planes = planes; out->plane_blocks = plane_blocks; out->order = nullptr;
        VEXHIP_REQUIRE(out->grid_blocks < (1ll << 31) && plane_blocks < (1ll << 31), "matrix too large for one launch");
        return 0;
    }
    VEXHIP_REQUIRE(order && g <= capacity, "order buffer too small");
    VEXHIP_TRY(hipMemcpyAsync(order, host.data(), sizeof(int) * (size_t)g, hipMemcpyHostToDevice, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    *grid_blocks = g;
    out->grid_blocks = g; out->order = order;
    return 0;
}

int vexhip_hell_order_i32(int dev, void *stream, int64_t n, int64_t w, int64_t pitch,
        const int32_t *ell_col, int mode, int32_t *order, int64_t capacity, vexhip_traversal *out)
{
    VEXHIP_REQUIRE(pitch > 0, "ELL pitch must be positive");
    return build_order(dev, stream, n, w, pitch, ell_col, mode, order, capacity, out);
}

int vexhip_sell_order_i32(int dev, void *stream, int64_t n, int64_t w, int value_bytes,
        const void *sell, int mode, int32_t *order, int64_t capacity, vexhip_traversal *out)
{
    VEXHIP_REQUIRE(value_bytes == 4 || value_bytes == 8, "value_bytes must be 4 or 8");
    return build_order(dev, stream, n, w, -(int64_t)value_bytes, static_cast<const int32_t *>(sell), mode, order, capacity, out);
}

int64_t vexhip_hell_order_capacity(int64_t n) { return 2 * ((n + kHellOrderRows - 1) / kHellOrderRows) + 4096; }

int vexhip_spmv_hell_ordered_f64_i32(int dev, void *stream, int64_t n, double alpha, int append,
        int64_t w, int64_t pitch, const int32_t *ec, const double *ev,
        const int32_t *cp, const int32_t *cc, const double *cv, const double *x, double *y,
        const vexhip_traversal *traversal)
{ return spmv_hell<double>(dev, stream, n, alpha, append, w, pitch, ec, ev, cp, cc, cv, x, y, traversal); }

int vexhip_spmv_hell_ordered_f32_i32(int dev, void *stream, int64_t n, float alpha, int append,
        int64_t w, int64_t pitch, const int32_t *ec, const float *ev,
        const int32_t *cp, const int32_t *cc, const float *cv, const float *x, float *y,
        const vexhip_traversal *traversal)
{ return spmv_hell<float>(dev, stream, n, alpha, append, w, pitch, ec, ev, cp, cc, cv, x, y, traversal); }

int64_t vexhip_sell_bytes(int64_t n, int64_t w, int value_bytes) { return (n + SELL_ROWS - 1) / SELL_ROWS * SELL_ROWS * w * (4 + value_bytes); }

int vexhip_sell_fill_f64_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const double *val,
        int64_t w, void *sell)
{ return sell_fill<double, int32_t>(dev, stream, n, ptr, col, val, w, sell); }

int vexhip_sell_fill_f32_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const float *val,
        int64_t w, void *sell)
{ return sell_fill<float, int32_t>(dev, stream, n, ptr, col, val, w, sell); }

int vexhip_spmv_sell_f64_i32(int dev, void *stream, int64_t n, double alpha, int append, int64_t w,
        const void *sc, const int32_t *cp, const int32_t *cc, const double *cv,
        const double *x, double *y, const vexhip_traversal *traversal)
{ return spmv_sell<double>(dev, stream, n, alpha, append, w, sc, cp, cc, cv, x, y, traversal); }

int vexhip_spmv_sell_f32_i32(int dev, void *stream, int64_t n, float alpha, int append, int64_t w,
        const void *sc, const int32_t *cp, const int32_t *cc, const float *cv,
        const float *x, float *y, const vexhip_traversal *traversal)
{ return spmv_sell<float>(dev, stream, n, alpha, append, w, sc, cp, cc, cv, x, y, traversal); }

int vexhip_spmv_csr_set_variant(int variant) { g_csr_variant = variant; return 0; }
int vexhip_spmv_hell_set_variant(int variant) { g_hell_variant = variant; return 0; }

int vexhip_spmv_csr_f64_i32(int dev, void *stream, int64_t n, double alpha, int append,
        const int32_t *ptr, const int32_t *col, const double *val, const double *x, double *y)
{ return spmv_csr<double, int>(dev, stream, n, alpha, append, ptr, col, val, x, y); }

int vexhip_spmv_csr_f32_i32(int dev, void *stream, int64_t n, float alpha, int append,
        const int32_t *ptr, const int32_t *col, const float *val, const float *x, float *y)
{ return spmv_csr<float, int>(dev, stream, n, alpha, append, ptr, col, val, x, y); }

int vexhip_spmv_csr_rows_f64_i32(int dev, void *stream, int64_t nrows, double alpha, const int32_t *rows,
        const int32_t *ptr, const int32_t *col, const double *val, const double *x, double *y)
{ return spmv_csr_rows<double>(dev, stream, nrows, alpha, rows, ptr, col, val, x, y); }

int vexhip_spmv_csr_rows_f32_i32(int dev, void *stream, int64_t nrows, float alpha, const int32_t *rows,
        const int32_t *ptr, const int32_t *col, const float *val, const float *x, float *y)
{ return spmv_csr_rows<float>(dev, stream, nrows, alpha, rows, ptr, col, val, x, y); }

int vexhip_spmv_csr_ordered_f64_i32(int dev, void *stream, int64_t n, double alpha, int append,
        const int32_t *ptr, const int32_t *col, const double *val, const double *x, double *y, const vexhip_traversal *traversal)
{ return spmv_csr<double, int>(dev, stream, n, alpha, append, ptr, col, val, x, y, traversal); }

int vexhip_spmv_csr_ordered_f32_i32(int dev, void *stream, int64_t n, float alpha, int append,
        const int32_t *ptr, const int32_t *col, const float *val, const float *x, float *y, const vexhip_traversal *traversal)
{ return spmv_csr<float, int>(dev, stream, n, alpha, append, ptr, col, val, x, y, traversal); }

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

VEXHIP_WARM_TU(spmv)
