// vex::mba control-lattice fit on the device (reference: vexcl/mba.hpp:233-480 -- the multilevel B-spline
// approximation of Lee, Wolberg and Shin, fitted on the HOST there; only the evaluation runs on the device).
//
// Here the whole hierarchy is fitted in HBM.  Per level:
//   accumulate  one lane per data point: its cell, the 4 basis values per dimension, and 4^N hardware fp64/fp32
//               atomic adds of (w^2 * proposal, w^2) into the numerator / denominator lattices;
//   finalize    phi = numerator / denominator;
//   residual    one lane per point: value -= spline(point), sum of squares by wave shuffles + one atomic per wave;
//   refine      one lane per node of the finer lattice GATHERS the (<= 3 per dimension) coarse nodes whose
//               subdivision mask reaches it (the reference scatters; gathering needs no atomics).
// The host only reads one scalar per level (the residual that decides whether another level is needed).
// Summation order inside a lattice node differs from the host loop's (atomics), so results agree with a host fit
// to rounding, not bit for bit; tests/cpp/extras_tests.cpp compares the two.
#include "common.hpp"

#include <algorithm>
#include <vector>

namespace vexhip {
namespace {

constexpr int MB = 256;

template <typename R, int N> struct lattice_dev {
    R xmin[N], hinv[N];
    long long n[N], stride[N];
    long long total;
};

template <typename R> __device__ __forceinline__ void basis(R t, R (&w)[4]) {
    w[0] = (t * (t * (-t + 3) - 3) + 1) / 6;
    w[1] = (t * t * (3 * t - 6) + 4) / 6;
    w[2] = (t * (t * (-3 * t + 3) + 3) + 1) / 6;
    w[3] = t * t * t / 6;
}

template <typename R, int N>
__device__ __forceinline__ void locate(const lattice_dev<R, N> &L, const R *p, long long (&cell)[N], R (&w)[N][4]) {
#pragma unroll
    for (int d = 0; d < N; ++d) {
        const R u = (p[d] - L.xmin[d]) * L.hinv[d];
        const R fl = floor(u);
        cell[d] = (long long)fl - 1;
        basis<R>(u - fl, w[d]);
    }
}

/// PRIVATE: the lattice is small enough for a copy per workgroup in LDS -- the coarse levels, where every point of the
/// cloud lands on the same few dozen nodes and global atomics would serialize in L2 (1M points on the 4 x 4 lattice:
/// 16M adds on 16 addresses).  The workgroup accumulates with LDS atomics and flushes its non-zero nodes once.
template <typename R, int N, bool PRIVATE>
__global__ void mba_accumulate(lattice_dev<R, N> L, const R *cmin, const R *cmax, const R *__restrict__ coo,
        const R *__restrict__ val, long long npts, R *__restrict__ gnum, R *__restrict__ gden)
{
    extern __shared__ __attribute__((aligned(16))) char mba_smem[];
    R *num = gnum, *den = gden;
    if constexpr (PRIVATE) {
        num = reinterpret_cast<R *>(mba_smem); den = num + L.total;
        for (long long i = threadIdx.x; i < 2 * L.total; i += MB) num[i] = 0;
        __syncthreads();
    }
    R lo[N], hi[N];
#pragma unroll
    for (int d = 0; d < N; ++d) { lo[d] = cmin[d]; hi[d] = cmax[d]; }
    for (long long i = blockIdx.x * (long long)MB + threadIdx.x; i < npts; i += (long long)gridDim.x * MB) {
        R p[N];
        bool inside = true;
#pragma unroll
        for (int d = 0; d < N; ++d) {
            p[d] = coo[i * N + d];
            const R eps = (R)1e-12;
            if (p[d] - eps < lo[d] || p[d] + eps >= hi[d]) inside = false;
        }
        if (!inside) continue;
        long long cell[N]; R w[N][4];
        locate<R, N>(L, p, cell, w);
        R sw2 = 1;                                     // sum over the tensor product of squares = product of the sums
#pragma unroll
        for (int d = 0; d < N; ++d) sw2 *= w[d][0] * w[d][0] + w[d][1] * w[d][1] + w[d][2] * w[d][2] + w[d][3] * w[d][3];
        const R v = val[i];
        constexpr int NW = N == 1 ? 4 : N == 2 ? 16 : 64;
        for (int t = 0; t < NW; ++t) {
            int rem = t; R wt = 1; long long at = 0;
#pragma unroll
            for (int d = N - 1; d >= 0; --d) { const int k = rem & 3; rem >>= 2; wt *= w[d][k]; at += (cell[d] + k) * L.stride[d]; }
            const R w2 = wt * wt;
            unsafeAtomicAdd(num + at, w2 * (v * wt / sw2));
            unsafeAtomicAdd(den + at, w2);
        }
    }
    if constexpr (PRIVATE) {
        __syncthreads();
        for (long long i = threadIdx.x; i < L.total; i += MB)
            if (den[i] != R(0)) { unsafeAtomicAdd(gnum + i, num[i]); unsafeAtomicAdd(gden + i, den[i]); }
    }
}

template <typename R>
__global__ void mba_finalize(const R *__restrict__ num, const R *__restrict__ den, R *__restrict__ phi, long long total) {
    for (long long i = blockIdx.x * (long long)MB + threadIdx.x; i < total; i += (long long)gridDim.x * MB)
        phi[i] = fabs(den[i]) < (R)1e-32 ? R(0) : num[i] / den[i];
}

template <typename R, int N>
__device__ __forceinline__ R evaluate(const lattice_dev<R, N> &L, const R *__restrict__ phi, const R *p) {
    long long cell[N]; R w[N][4];
    locate<R, N>(L, p, cell, w);
    R f = 0;
    constexpr int NW = N == 1 ? 4 : N == 2 ? 16 : 64;
    for (int t = 0; t < NW; ++t) {
        int rem = t; R wt = 1; long long at = 0; bool in = true;
#pragma unroll
        for (int d = N - 1; d >= 0; --d) {
            const int k = rem & 3; rem >>= 2;
            const long long j = cell[d] + k;
            if (j < 0 || j >= L.n[d]) in = false;
            wt *= w[d][k]; at += j * L.stride[d];
        }
        if (in) f += wt * phi[at];
    }
    return f;
}

/// val -= spline(coo) (phi == null: nothing subtracted); *res += sum val^2
template <typename R, int N>
__global__ void mba_residual(lattice_dev<R, N> L, const R *__restrict__ phi, const R *__restrict__ coo, R *__restrict__ val,
        long long npts, double *res)
{
    double acc = 0;
    for (long long i = blockIdx.x * (long long)MB + threadIdx.x; i < npts; i += (long long)gridDim.x * MB) {
        R v = val[i];
        if (phi) {
            R p[N];
#pragma unroll
            for (int d = 0; d < N; ++d) p[d] = coo[i * N + d];
            v -= evaluate<R, N>(L, phi, p);
            val[i] = v;
        }
        acc += (double)v * (double)v;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) unsafeAtomicAdd(res, acc);
}

/// fine.phi[j] += sum over coarse nodes i and mask positions d with 2 i + d - 3 == j (per dimension) of coarse.phi[i] * prod mask[d]
template <typename R, int N>
__global__ void mba_refine(lattice_dev<R, N> F, R *__restrict__ fphi, lattice_dev<R, N> C, const R *__restrict__ cphi) {
    const R mask[5] = {(R)0.125, (R)0.5, (R)0.75, (R)0.5, (R)0.125};
    for (long long idx = blockIdx.x * (long long)MB + threadIdx.x; idx < F.total; idx += (long long)gridDim.x * MB) {
        long long j[N], rem = idx;
#pragma unroll
        for (int d = 0; d < N; ++d) { j[d] = rem / F.stride[d]; rem -= j[d] * F.stride[d]; }
        // per dimension: coarse i in [ceil((j - 1) / 2), floor((j + 3) / 2)] gives d = j + 3 - 2 i in [0, 4]
        long long i0[N]; int cnt[N];
#pragma unroll
        for (int d = 0; d < N; ++d) {
            const long long lo = (j[d] >= 1) ? (j[d]) / 2 : 0;           // ceil((j - 1) / 2) for j >= 1; 0 otherwise
            const long long hi = (j[d] + 3) / 2;
            i0[d] = lo; cnt[d] = (int)(hi - lo + 1);
        }
        R sum = 0;
        const int c0 = cnt[0], c1 = N > 1 ? cnt[N > 1 ? 1 : 0] : 1, c2 = N > 2 ? cnt[N > 2 ? 2 : 0] : 1;
        for (int a = 0; a < c0; ++a) for (int b = 0; b < c1; ++b) for (int c = 0; c < c2; ++c) {
            const int off[3] = {a, b, c};
            R wgt = 1; long long at = 0; bool in = true;
#pragma unroll
            for (int d = 0; d < N; ++d) {
                const long long i = i0[d] + off[d];
                const long long dd = j[d] + 3 - 2 * i;
                if (i >= C.n[d] || dd < 0 || dd > 4) { in = false; } else { wgt *= mask[dd]; at += i * C.stride[d]; }
            }
            if (in) sum += cphi[at] * wgt;
        }
        fphi[idx] += sum;
    }
}

template <typename R, int N>
lattice_dev<R, N> make_lattice(const double *cmin, const double *cmax, const size_t *grid) {
    lattice_dev<R, N> L;
    for (int d = 0; d < N; ++d) {
        L.hinv[d] = (R)(grid[d] - 1) / ((R)cmax[d] - (R)cmin[d]);
        L.xmin[d] = (R)cmin[d] - 1 / L.hinv[d];
        L.n[d] = (long long)grid[d] + 2;
    }
    L.stride[N - 1] = 1;
    for (int d = N - 2; d >= 0; --d) L.stride[d] = L.stride[d + 1] * L.n[d + 1];
    L.total = L.n[0] * L.stride[0];
    return L;
}

inline unsigned blocks_for(long long n) { return (unsigned)std::max<long long>(1, std::min<long long>((n + MB - 1) / MB, 1 << 16)); }

template <typename R, int N>
int fit(int dev, hipStream_t st, const double *cmin, const double *cmax, const R *coo, R *val, long long npts,
        const size_t *grid0, int levels, double tol, double *xmin_out, double *hinv_out, size_t *n_out, size_t *stride_out,
        void **phi_out, size_t *phi_elems)
{
    VEXHIP_SET_DEVICE(dev);
    size_t grid[N];
    for (int d = 0; d < N; ++d) { VEXHIP_REQUIRE(grid0[d] > 1, "mba: the control grid needs at least 2 points per dimension"); grid[d] = grid0[d]; }
    R *bounds = nullptr; double *res_d = nullptr;
    VEXHIP_TRY(hipMalloc(&bounds, 2 * N * sizeof(R)));
    VEXHIP_TRY(hipMalloc(&res_d, sizeof(double)));
    R hb[2 * N];
    for (int d = 0; d < N; ++d) { hb[d] = (R)cmin[d]; hb[N + d] = (R)cmax[d]; }
    VEXHIP_TRY(hipMemcpyAsync(bounds, hb, sizeof(hb), hipMemcpyHostToDevice, st));

    auto read_res = [&](double &r) -> int {
        VEXHIP_TRY(hipMemcpyAsync(&r, res_d, sizeof(double), hipMemcpyDeviceToHost, st));
        VEXHIP_TRY(hipStreamSynchronize(st));
        return 0;
    };
    auto level = [&](const lattice_dev<R, N> &L, R *&phi, double &res) -> int {
        R *num = nullptr, *den = nullptr;
        VEXHIP_TRY(hipMalloc(&num, L.total * sizeof(R)));
        VEXHIP_TRY(hipMalloc(&den, L.total * sizeof(R)));
        VEXHIP_TRY(hipMalloc(&phi, L.total * sizeof(R)));
        VEXHIP_TRY(hipMemsetAsync(num, 0, L.total * sizeof(R), st));
        VEXHIP_TRY(hipMemsetAsync(den, 0, L.total * sizeof(R), st));
        VEXHIP_TRY(hipMemsetAsync(res_d, 0, sizeof(double), st));
        const size_t priv = 2 * (size_t)L.total * sizeof(R);
        if (npts && priv <= 48 * 1024)             // a few workgroups per CU, each sweeping its share of the cloud
            mba_accumulate<R, N, true><<<std::min(blocks_for(npts), 1024u), MB, priv, st>>>(L, bounds, bounds + N, coo, val, npts, num, den);
        else if (npts)
            mba_accumulate<R, N, false><<<blocks_for(npts), MB, 0, st>>>(L, bounds, bounds + N, coo, val, npts, num, den);
        VEXHIP_LAUNCH_CHECK();
        mba_finalize<R><<<blocks_for(L.total), MB, 0, st>>>(num, den, phi, L.total);
        VEXHIP_LAUNCH_CHECK();
        if (npts) mba_residual<R, N><<<blocks_for(npts), MB, 0, st>>>(L, phi, coo, val, npts, res_d);
        VEXHIP_LAUNCH_CHECK();
        if (int rc = read_res(res)) return rc;
        VEXHIP_TRY(hipFree(num)); VEXHIP_TRY(hipFree(den));
        return 0;
    };

    double res0 = 0, res = 0;
    VEXHIP_TRY(hipMemsetAsync(res_d, 0, sizeof(double), st));
    lattice_dev<R, N> L = make_lattice<R, N>(cmin, cmax, grid);
    if (npts) mba_residual<R, N><<<blocks_for(npts), MB, 0, st>>>(L, (const R *)nullptr, coo, val, npts, res_d);
    VEXHIP_LAUNCH_CHECK();
    if (int rc = read_res(res0)) return rc;

    R *psi = nullptr;
    if (int rc = level(L, psi, res)) return rc;
    for (int k = 1; res > res0 * tol && k < levels; ++k) {
        for (int d = 0; d < N; ++d) grid[d] = 2 * grid[d] - 1;
        lattice_dev<R, N> F = make_lattice<R, N>(cmin, cmax, grid);
        R *f = nullptr;
        if (int rc = level(F, f, res)) return rc;
        mba_refine<R, N><<<blocks_for(F.total), MB, 0, st>>>(F, f, L, psi);
        VEXHIP_LAUNCH_CHECK();
        VEXHIP_TRY(hipStreamSynchronize(st));
        VEXHIP_TRY(hipFree(psi));
        psi = f; L = F;
    }
    VEXHIP_TRY(hipStreamSynchronize(st));
    VEXHIP_TRY(hipFree(bounds)); VEXHIP_TRY(hipFree(res_d));
    for (int d = 0; d < N; ++d) { xmin_out[d] = (double)L.xmin[d]; hinv_out[d] = (double)L.hinv[d]; n_out[d] = (size_t)L.n[d]; stride_out[d] = (size_t)L.stride[d]; }
    *phi_out = psi; *phi_elems = (size_t)L.total;
    return 0;
}

template <typename R>
int fit_dim(int dev, hipStream_t st, int ndim, const double *cmin, const double *cmax, const void *coo, void *val, long long npts,
        const size_t *grid, int levels, double tol, double *xmin, double *hinv, size_t *n, size_t *stride, void **phi, size_t *phi_elems)
{
    switch (ndim) {
        case 1: return fit<R, 1>(dev, st, cmin, cmax, (const R *)coo, (R *)val, npts, grid, levels, tol, xmin, hinv, n, stride, phi, phi_elems);
        case 2: return fit<R, 2>(dev, st, cmin, cmax, (const R *)coo, (R *)val, npts, grid, levels, tol, xmin, hinv, n, stride, phi, phi_elems);
        case 3: return fit<R, 3>(dev, st, cmin, cmax, (const R *)coo, (R *)val, npts, grid, levels, tol, xmin, hinv, n, stride, phi, phi_elems);
    }
    return fail(__FILE__, __LINE__, "mba: the device fit supports 1 to 3 dimensions");
}

} // namespace
} // namespace vexhip

using namespace vexhip;

extern "C" int vexhip_mba_fit(int dev, void *stream, int dtype, int ndim, const double *cmin, const double *cmax,
        const void *coo, void *val, int64_t npts, const size_t *grid, int levels, double tol,
        double *xmin, double *hinv, size_t *n, size_t *stride, void **phi, size_t *phi_elems)
{
    VEXHIP_REQUIRE(cmin && cmax && grid && xmin && hinv && n && stride && phi && phi_elems && (npts == 0 || (coo && val)), "mba: NULL argument");
    VEXHIP_REQUIRE(dtype == VEXHIP_F32 || dtype == VEXHIP_F64, "mba: only float and double data are supported");
    if (dtype == VEXHIP_F64)
        return fit_dim<double>(dev, as_stream(stream), ndim, cmin, cmax, coo, val, npts, grid, levels, tol, xmin, hinv, n, stride, phi, phi_elems);
    return fit_dim<float>(dev, as_stream(stream), ndim, cmin, cmax, coo, val, npts, grid, levels, tol, xmin, hinv, n, stride, phi, phi_elems);
}

VEXHIP_WARM_TU(mba)
