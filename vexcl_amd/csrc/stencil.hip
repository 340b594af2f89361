// vex::stencil convolution on gfx950 (vexcl/stencil.hpp:306-405 `slow_conv` /
// `fast_conv`):   y[i] = beta*y[i] + alpha * sum_j s[j] * X(i + j - lhalo)
// where X reads the local segment, the halo buffer of the neighbouring devices,
// or -- at the ends of the whole vector -- the edge element (stencil.hpp:264-290
// `read_x`).  The one place the expression path uses LDS: a workgroup stages its
// 1024 outputs' inputs (+ halo) and the stencil itself in LDS, then every lane
// folds its four outputs from LDS; each x element is read from HBM once.
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
        for (int j = threadIdx.x; j < span; j += CB) {
            long long g = g0 - lhalo + j;
            X[j] = (g < n + rhalo) ? read_x<T>(g, n, has_left, has_right, lhalo, xloc, xrem) : T(0);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CI; ++k) {
            const int o = threadIdx.x + k * CB;
            const long long i = g0 + o;
            if (i < n) {
                T sum = 0;
                for (int j = 0; j < width; ++j) sum += S[j] * X[o + j];
                y[i] = (beta != T(0)) ? beta * y[i] + alpha * sum : alpha * sum;
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
    const size_t lds = sizeof(T) * (size_t)(CTILE + 2 * (lhalo + rhalo) + 1);
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
