// vex::stencil convolution on gfx950 (vexcl/stencil.hpp:306-405 `slow_conv` /
// `fast_conv`):   y[i] = beta*y[i] + alpha * sum_j s[j] * X(i + j - lhalo)
// where X reads the local segment, the halo buffer of the neighbouring devices,
// or -- at the ends of the whole vector -- the edge element (stencil.hpp:264-290
// `read_x`).  The one place the expression path uses LDS: a workgroup stages its
// 1024 outputs' inputs (+ halo) and the stencil itself in LDS; each x element is read
// from HBM once.  A lane then folds FOUR CONSECUTIVE outputs with a sliding window
// of four inputs in registers: one new LDS read of x and one (broadcast) read of the
// stencil per tap serve four outputs -- 45 LDS reads per four outputs of a 21-point
// stencil instead of 168; the first version was bound by exactly that LDS traffic
// (33.6 GB at 69 TB/s aggregate = 0.49 ms; measured 0.46).  Element i lives at
// i + i/16 in LDS: with lanes 32 bytes apart the skew spreads a wave's 8-byte reads
// over all banks.
#include "common.hpp"

#include <algorithm>

namespace vexhip {
namespace {

constexpr int CB = 256;      // lanes
constexpr int CI = 4;        // outputs per lane
constexpr int CTILE = CB * CI;

template <typename T>
__device__ __forceinline__ T read_x(long long g, long long n, int has_left, int has_right, int lhalo,
        const T *__restrict__ xloc, const T *__restrict__ xrem)
{
    if (g >= 0 && g < n) return xloc[g];
    if (g < 0) return has_left ? xrem[lhalo + g] : xloc[0];
    return has_right ? xrem[lhalo + (g - n)] : xloc[n - 1];
}

__device__ __forceinline__ int skew(int i) { return i + (i >> 4); }
__host__ __device__ inline size_t lds_elems(int lhalo, int rhalo) {
    const size_t span = (size_t)CTILE + lhalo + rhalo + CI;
    return (size_t)(lhalo + rhalo + 1) + span + span / 16 + 2;
}

template <typename T, bool USE_LDS>
__global__ __launch_bounds__(CB)
void stencil_conv_kernel(long long n, int has_left, int has_right, int lhalo, int rhalo,
        const T *__restrict__ s, const T *__restrict__ xloc, const T *__restrict__ xrem,
        T *__restrict__ y, T beta, T alpha)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int width = lhalo + rhalo + 1;
    const long long g0 = (long long)blockIdx.x * CTILE;
    if constexpr (USE_LDS) {
        T *S = reinterpret_cast<T *>(smem_raw);
        T *X = S + width;
        for (int j = threadIdx.x; j < width; j += CB) S[j] = s[j];
        const int span = CTILE + lhalo + rhalo;
        typedef T v2 __attribute__((ext_vector_type(2)));
        const long long first = g0 - lhalo;                       // global index of X[0]
        const long long a = first & ~1ll;                          // ... rounded down to an aligned pair
        if (first >= 1 && a + 2 * (long long)((span + CI + 2) / 2) <= n && (reinterpret_cast<unsigned long long>(xloc) & (2 * sizeof(T) - 1)) == 0) {
            // interior tile: everything it reads is in the local segment -- aligned 16-byte loads, two elements per lane
            const int off = (int)(first - a);
            const v2 *xp = reinterpret_cast<const v2 *>(xloc + a);
            for (int p = threadIdx.x; 2 * p < span + CI + off; p += CB) {
                const v2 xx = xp[p];
                const int j = 2 * p - off;
                if (j >= 0) X[skew(j)] = xx.x;
                if (j + 1 < span + CI) X[skew(j + 1)] = xx.y;
            }
        } else {
            for (int j = threadIdx.x; j < span + CI; j += CB) {       // + CI: the window reads a few elements past the last tap
                long long g = g0 - lhalo + j;
                X[skew(j)] = (j < span && g < n + rhalo) ? read_x<T>(g, n, has_left, has_right, lhalo, xloc, xrem) : T(0);
            }
        }
        __syncthreads();
        const int o = CI * threadIdx.x;                    // this lane's first output within the tile
        T w0 = X[skew(o)], w1 = X[skew(o + 1)], w2 = X[skew(o + 2)], w3 = X[skew(o + 3)];
        T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int j = 0; j < width; ++j) {                  // taps in ascending order for every output
            const T sj = S[j];
            a0 += sj * w0; a1 += sj * w1; a2 += sj * w2; a3 += sj * w3;
            w0 = w1; w1 = w2; w2 = w3; w3 = X[skew(o + j + 4)];
        }
        const T acc[CI] = {a0, a1, a2, a3};
        const long long i0 = g0 + o;
        typedef T v2 __attribute__((ext_vector_type(2)));
        if (i0 + CI <= n && (reinterpret_cast<unsigned long long>(y) & (2 * sizeof(T) - 1)) == 0) {
            v2 *yp = reinterpret_cast<v2 *>(y + i0);           // i0 is a multiple of 4: two aligned vector stores
            v2 lo, hi;
            lo.x = alpha * acc[0]; lo.y = alpha * acc[1]; hi.x = alpha * acc[2]; hi.y = alpha * acc[3];
            if (beta != T(0)) {
                const v2 ol = yp[0], oh = yp[1];
                lo.x = beta * ol.x + lo.x; lo.y = beta * ol.y + lo.y; hi.x = beta * oh.x + hi.x; hi.y = beta * oh.y + hi.y;
            }
            yp[0] = lo; yp[1] = hi;
        } else {
#pragma unroll
            for (int k = 0; k < CI; ++k) {
                const long long i = i0 + k;
                if (i < n) y[i] = (beta != T(0)) ? beta * y[i] + alpha * acc[k] : alpha * acc[k];
            }
        }
    } else {
        for (int k = 0; k < CI; ++k) {
            const long long i = g0 + threadIdx.x + k * CB;
            if (i < n) {
                T sum = 0;
                for (int j = 0; j < width; ++j)
                    sum += s[j] * read_x<T>(i + j - lhalo, n, has_left, has_right, lhalo, xloc, xrem);
                y[i] = (beta != T(0)) ? beta * y[i] + alpha * sum : alpha * sum;
            }
        }
    }
}

template <typename T>
int conv(int dev, void *stream, int64_t n, int has_left, int has_right, int lhalo, int rhalo,
        const T *s, const T *x, const T *xrem, T *y, T beta, T alpha)
{
    VEXHIP_REQUIRE(n >= 0 && lhalo >= 0 && rhalo >= 0, "bad stencil geometry");
    if (n == 0) return 0;
    VEXHIP_REQUIRE(s && x && y && (xrem || (!has_left && !has_right)), "NULL argument");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t st = as_stream(stream);
    const int64_t grid = (n + CTILE - 1) / CTILE;
    VEXHIP_REQUIRE(grid < (1ll << 31), "vector too large for one launch");
    const size_t lds = sizeof(T) * lds_elems(lhalo, rhalo);
    if (lds <= 64 * 1024)
        stencil_conv_kernel<T, true><<<(unsigned)grid, CB, lds, st>>>(n, has_left, has_right, lhalo, rhalo, s, x, xrem, y, beta, alpha);
    else   // stencil wider than LDS: every lane reads x directly (the reference's slow_conv)
        stencil_conv_kernel<T, false><<<(unsigned)grid, CB, 0, st>>>(n, has_left, has_right, lhalo, rhalo, s, x, xrem, y, beta, alpha);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

} // namespace
} // namespace vexhip

using namespace vexhip;

extern "C" {

int vexhip_stencil_conv_f64(int dev, void *stream, int64_t n, int has_left, int has_right, int lhalo, int rhalo,
        const double *s, const double *x, const double *xrem, double *y, double beta, double alpha)
{ return conv<double>(dev, stream, n, has_left, has_right, lhalo, rhalo, s, x, xrem, y, beta, alpha); }

int vexhip_stencil_conv_f32(int dev, void *stream, int64_t n, int has_left, int has_right, int lhalo, int rhalo,
        const float *s, const float *x, const float *xrem, float *y, float beta, float alpha)
{ return conv<float>(dev, stream, n, has_left, has_right, lhalo, rhalo, s, x, xrem, y, beta, alpha); }

} // extern "C"

VEXHIP_WARM_TU(stencil)
