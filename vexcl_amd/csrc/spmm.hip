// Multi-right-hand-side products  Y_k (+)= alpha * A * X_k,  k < nrhs, for the
// SELL-512 and SELL8 storage of spmv.hip / sell8.hip: `SpMat * multivector`
// (reference: vexcl/spmat.hpp:388-398 applies the product component by
// component, streaming the matrix nrhs times).
//
// The matrix stream (9-12 B per entry) dominates a product's HBM traffic, the
// x/y vectors are 16 B per ROW.  One launch therefore reads every slice ONCE and
// reuses each (column, value) pair for up to four right-hand sides held in
// registers: for the 7-point Poisson matrix two right-hand sides cost 1.18x the
// bytes of one instead of 2x.  Per right-hand side the arithmetic and its order
// are exactly the single-vector kernels' (this file is compiled with
// -ffp-contract=off too), so every Y_k is bit-identical to the separate product.
#include "common.hpp"
#include "traversal.hpp"

namespace vexhip {
namespace {

constexpr int ROWS = 512;
constexpr unsigned PAD8 = 254;        // codes 254 and 255 are padding (sell8.hip, pair kernels)
constexpr int MAX_NR = 4;

typedef int    int2v    __attribute__((ext_vector_type(2)));
typedef double double2v __attribute__((ext_vector_type(2)));
typedef float  float2v  __attribute__((ext_vector_type(2)));
template <typename V> struct vec2;
template <> struct vec2<double> { typedef double2v type; };
template <> struct vec2<float> { typedef float2v type; };

template <typename V> struct rhs_set { const V *x[MAX_NR]; V *y[MAX_NR]; };

// x for the two rows of a lane in ELL column j of a diagonal-coded slice, as the single-vector pair kernels read it
// (sell8.hip, "PAIR kernels"): ONE 16-byte load when both rows sit on the same diagonal or one of them is padding that
// allows it (code 255), an 8-byte load per real entry otherwise; padding reads as 0.  c0 / c1: the two rows' codes,
// d: s_delta[c0 is real ? c0 : c1], looked up once per column by the caller.
template <typename V>
__device__ __forceinline__ void gather_pair(const V *__restrict__ x, long long i, unsigned c0, unsigned c1, int d,
        const int *s_delta, const int *__restrict__ safe, V &x0, V &x1)
{
    typedef typename vec2<V>::type V2;
    const bool m0 = c0 < PAD8, m1 = c1 < PAD8;
    const bool pair = (c0 == c1) || (c0 == 255u && m1) || (c1 == 255u && m0);
    const bool use16 = pair && (m0 || m1);
    V2 p = {V(0), V(0)};
    if (__builtin_amdgcn_ballot_w64(use16) != 0) {          // skipped only when no lane of the wave has a pair here
        const V *px = use16 ? x + (i + d) : reinterpret_cast<const V *>(safe);
        __builtin_memcpy(&p, px, sizeof(V2));
    }
    x0 = p.x; x1 = p.y;
    if (!pair) {
        if (m0) x0 = x[i + s_delta[c0]];
        if (m1) x1 = x[i + 1 + s_delta[c1]];
    }
    x0 = m0 ? x0 : V(0);
    x1 = m1 ? x1 : V(0);
}

template <typename V, int NR>
__device__ __forceinline__ void tail_and_store(long long n, long long i, V alpha, int append,
        const int *__restrict__ csr_ptr, const int *__restrict__ csr_col, const V *__restrict__ csr_val,
        const rhs_set<V> &io, V (&sum)[NR][2])
{
    if (csr_ptr) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (i + q < n)
                for (int j = csr_ptr[i + q], e = csr_ptr[i + q + 1]; j < e; ++j) {
                    const int c = csr_col[j]; const V v = csr_val[j];
#pragma unroll
                    for (int k = 0; k < NR; ++k) sum[k][q] += v * io.x[k][c];
                }
    }
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (i + q < n) {
                V o = alpha * sum[k][q];
                if (append) o = io.y[k][i + q] + o;
                io.y[k][i + q] = o;
            }
}

// ---- SELL8: 1-byte diagonal codes (layout: sell8.hip) -----------------------------------
template <typename V, int W, int NR>
__global__ __launch_bounds__(256)
void spmm_sell8_kernel(long long n, long long nslices, V alpha, int append, int ell_w,
        const char *__restrict__ buf, const int *__restrict__ deltas,
        const int *__restrict__ csr_ptr, const int *__restrict__ csr_col, const V *__restrict__ csr_val,
        rhs_set<V> io, trav_dev trav, const char *__restrict__ pool, const int *__restrict__ blocks)
{
    __shared__ int s_delta[256];
    s_delta[threadIdx.x] = deltas[threadIdx.x];
    __syncthreads();

    const long long s = traversal_block(trav, nslices);
    if (s < 0) return;
    const int t = threadIdx.x;
    const long long i = s * ROWS + 2 * t;
    const int w = W > 0 ? W : ell_w;
    const int wp = (w + 1) / 2;
    const char *slice = buf + s * ((long long)wp * 1024 + (long long)w * ROWS * sizeof(V));
    // slice dictionary (sell8.hip): codes from the pool of distinct code blocks, values from the slice
    const unsigned *cw = reinterpret_cast<const unsigned *>(blocks ? pool + (long long)blocks[s] * ((long long)wp * 1024) : slice) + t;
    const V *vp = reinterpret_cast<const V *>(slice + (long long)wp * 1024) + 2 * t;
    typedef typename vec2<V>::type V2;

    V sum[NR][2];
#pragma unroll
    for (int k = 0; k < NR; ++k) { sum[k][0] = V(0); sum[k][1] = V(0); }

    if constexpr (W > 0) {
        constexpr int WP = (W + 1) / 2;
        unsigned c[WP]; V2 v[W];
#pragma unroll
        for (int jp = 0; jp < WP; ++jp) c[jp] = blocks ? cw[jp * 256] : __builtin_nontemporal_load(cw + jp * 256);
#pragma unroll
        for (int j = 0; j < W; ++j) v[j] = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(vp + j * ROWS));
        int d[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const unsigned c0 = (c[j >> 1] >> (16 * (j & 1))) & 255u, c1 = (c[j >> 1] >> (16 * (j & 1) + 8)) & 255u;
            d[j] = s_delta[c0 < PAD8 ? c0 : c1];
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            V xv[W][2];
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const unsigned c0 = (c[j >> 1] >> (16 * (j & 1))) & 255u, c1 = (c[j >> 1] >> (16 * (j & 1) + 8)) & 255u;
                gather_pair<V>(io.x[k], i, c0, c1, d[j], s_delta, deltas, xv[j][0], xv[j][1]);
            }
            // padding: stored value 0, gathered value 0 -- sum + (+-0) == sum bit for bit (the sum starts at +0)
#pragma unroll
            for (int j = 0; j < W; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) sum[k][q] += v[j][q] * xv[j][q];
        }
    } else {
        for (int j = 0; j < w; ++j) {
            const unsigned cword = cw[(j >> 1) * 256];
            const V2 vv = *reinterpret_cast<const V2 *>(vp + (long long)j * ROWS);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned code = (cword >> (8 * ((j & 1) * 2 + q))) & 255u;
                if (code < PAD8) {
                    const long long cidx = i + q + s_delta[code];
#pragma unroll
                    for (int k = 0; k < NR; ++k) sum[k][q] += vv[q] * io.x[k][cidx];
                }
            }
        }
    }
    tail_and_store<V, NR>(n, i, alpha, append, csr_ptr, csr_col, csr_val, io, sum);
}

// ---- SELL8V: diagonal codes and value codes (layout: sell8.hip) -----------------------------
template <typename V, int W, int NR>
__global__ __launch_bounds__(256)
void spmm_sell8v_kernel(long long n, long long nslices, V alpha, int append, int ell_w,
        const char *__restrict__ buf, const int *__restrict__ deltas, const V *__restrict__ values,
        const int *__restrict__ csr_ptr, const int *__restrict__ csr_col, const V *__restrict__ csr_val,
        rhs_set<V> io, trav_dev trav, const int *__restrict__ blocks)
{
    __shared__ int s_delta[256];
    __shared__ V s_value[256];
    s_delta[threadIdx.x] = deltas[threadIdx.x];
    s_value[threadIdx.x] = values[threadIdx.x];
    __syncthreads();

    const long long s = traversal_block(trav, nslices);
    if (s < 0) return;
    const int t = threadIdx.x;
    const long long i = s * ROWS + 2 * t;
    const int w = W > 0 ? W : ell_w;
    const int wp = (w + 1) / 2;
    const long long sb = blocks ? (long long)blocks[s] : s;                 // slice dictionary (sell8.hip)
    const unsigned *cw = reinterpret_cast<const unsigned *>(buf + sb * ((long long)wp * 2048)) + t;
    const unsigned *vw = cw + wp * 256;

    V sum[NR][2];
#pragma unroll
    for (int k = 0; k < NR; ++k) { sum[k][0] = V(0); sum[k][1] = V(0); }

    if constexpr (W > 0) {
        constexpr int WP = (W + 1) / 2;
        unsigned c[WP], vc[WP];
#pragma unroll
        for (int jp = 0; jp < WP; ++jp) {
            if (blocks) { c[jp] = cw[jp * 256]; vc[jp] = vw[jp * 256]; }           // pooled blocks: cached loads
            else { c[jp] = __builtin_nontemporal_load(cw + jp * 256); vc[jp] = __builtin_nontemporal_load(vw + jp * 256); }
        }
        int d[W]; V val[W][2];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const unsigned c0 = (c[j >> 1] >> (16 * (j & 1))) & 255u, c1 = (c[j >> 1] >> (16 * (j & 1) + 8)) & 255u;
            d[j] = s_delta[c0 < PAD8 ? c0 : c1];
            val[j][0] = s_value[c0 < PAD8 ? (vc[j >> 1] >> (16 * (j & 1))) & 255u : 255u];          // entry 255 is 0.0
            val[j][1] = s_value[c1 < PAD8 ? (vc[j >> 1] >> (16 * (j & 1) + 8)) & 255u : 255u];
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            V xv[W][2];
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const unsigned c0 = (c[j >> 1] >> (16 * (j & 1))) & 255u, c1 = (c[j >> 1] >> (16 * (j & 1) + 8)) & 255u;
                gather_pair<V>(io.x[k], i, c0, c1, d[j], s_delta, deltas, xv[j][0], xv[j][1]);
            }
#pragma unroll
            for (int j = 0; j < W; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) sum[k][q] += val[j][q] * xv[j][q];
        }
    } else {
        for (int j = 0; j < w; ++j) {
            const unsigned cword = cw[(j >> 1) * 256], vword = vw[(j >> 1) * 256];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int sh = 8 * ((j & 1) * 2 + q);
                const unsigned code = (cword >> sh) & 255u;
                if (code < PAD8) {
                    const long long cidx = i + q + s_delta[code];
                    const V v = s_value[(vword >> sh) & 255u];
#pragma unroll
                    for (int k = 0; k < NR; ++k) sum[k][q] += v * io.x[k][cidx];
                }
            }
        }
    }
    tail_and_store<V, NR>(n, i, alpha, append, csr_ptr, csr_col, csr_val, io, sum);
}

// ---- SELL-512 with 32-bit columns (layout: spmv.hip) ---------------------------------------
template <typename V, int W, int NR>
__global__ __launch_bounds__(256)
void spmm_sell_kernel(long long n, long long nslices, V alpha, int append, int ell_w,
        const char *__restrict__ sell,
        const int *__restrict__ csr_ptr, const int *__restrict__ csr_col, const V *__restrict__ csr_val,
        rhs_set<V> io, trav_dev trav)
{
    const long long s = traversal_block(trav, nslices);
    if (s < 0) return;
    const int r = 2 * threadIdx.x;
    const long long i = s * ROWS + r;
    const int w = W > 0 ? W : ell_w;
    const char *slice = sell + s * ((long long)w * ROWS * (4 + (long long)sizeof(V)));
    const int *cp = reinterpret_cast<const int *>(slice) + r;
    const V *vp = reinterpret_cast<const V *>(slice + (long long)w * ROWS * 4) + r;
    typedef typename vec2<V>::type V2;

    V sum[NR][2];
#pragma unroll
    for (int k = 0; k < NR; ++k) { sum[k][0] = V(0); sum[k][1] = V(0); }

    if constexpr (W > 0) {
        int2v c[W]; V2 v[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            c[j] = __builtin_nontemporal_load(reinterpret_cast<const int2v *>(cp + j * ROWS));
            v[j] = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(vp + j * ROWS));
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            V xv[W][2];
#pragma unroll
            for (int j = 0; j < W; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) xv[j][q] = (c[j][q] >= 0) ? io.x[k][c[j][q]] : V(0);
#pragma unroll
            for (int j = 0; j < W; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) if (c[j][q] >= 0) sum[k][q] += v[j][q] * xv[j][q];
        }
    } else {
        for (int j = 0; j < w; ++j) {
            const int2v c = *reinterpret_cast<const int2v *>(cp + (long long)j * ROWS);
            const V2 v = *reinterpret_cast<const V2 *>(vp + (long long)j * ROWS);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (c[q] >= 0) {
#pragma unroll
                    for (int k = 0; k < NR; ++k) sum[k][q] += v[q] * io.x[k][c[q]];
                }
        }
    }
    tail_and_store<V, NR>(n, i, alpha, append, csr_ptr, csr_col, csr_val, io, sum);
}

// CODES: 0 = 32-bit columns, 1 = diagonal codes, 2 = diagonal and value codes
template <typename V, int CODES, int NR>
void launch(hipStream_t s, long long grid, long long n, long long ns, V alpha, int append, int w, const char *buf,
        const int *deltas, const V *values, const int *cp, const int *cc, const V *cv, const rhs_set<V> &io, const trav_dev &t,
        const char *pool, const int *blocks)
{
#define LAUNCH(W)                                                                                              \
    do {                                                                                                       \
        if constexpr (CODES == 2) spmm_sell8v_kernel<V, W, NR><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, w, buf, deltas, values, cp, cc, cv, io, t, blocks); \
        else if constexpr (CODES == 1) spmm_sell8_kernel<V, W, NR><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, w, buf, deltas, cp, cc, cv, io, t, pool, blocks); \
        else spmm_sell_kernel<V, W, NR><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, w, buf, cp, cc, cv, io, t);               \
    } while (0)
    switch (w) {          // unrolled for the usual stencil widths (1-D, 2-D 5/9-point, 3-D 7-point); any other width loops
        case 3: LAUNCH(3); break;
        case 5: LAUNCH(5); break;
        case 7: LAUNCH(7); break;
        case 9: LAUNCH(9); break;
        default: LAUNCH(0);
    }
#undef LAUNCH
}

template <typename V, int CODES>
int spmm(int dev, void *stream, int64_t n, int nrhs, V alpha, int append, int64_t w, const void *buf, const int *deltas, const V *values,
        const int *cp, const int *cc, const V *cv, const V *const *x, V *const *y, const vexhip_traversal *tr, const int *blocks = nullptr, const void *pool_ = nullptr)
{
    VEXHIP_REQUIRE(n >= 0 && w >= 1 && w < (1 << 20) && nrhs >= 1, "bad SpMM geometry");
    if (n == 0) return 0;
    VEXHIP_REQUIRE(buf && x && y && (CODES == 0 || deltas) && (CODES != 2 || values) && (reinterpret_cast<uintptr_t>(buf) & 15) == 0,
            "NULL argument or misaligned matrix buffer");
    for (int k = 0; k < nrhs; ++k) VEXHIP_REQUIRE(x[k] && y[k], "NULL right-hand side or result");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    const long long ns = (n + ROWS - 1) / ROWS;
    long long grid = 0;
    const trav_dev t = make_traversal(tr, ns, &grid);
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    const char *b = static_cast<const char *>(buf), *pool = static_cast<const char *>(pool_);
    for (int k0 = 0; k0 < nrhs; k0 += MAX_NR) {
        const int nr = nrhs - k0 < MAX_NR ? nrhs - k0 : MAX_NR;
        rhs_set<V> io;
        for (int k = 0; k < MAX_NR; ++k) { io.x[k] = x[k0 + (k < nr ? k : 0)]; io.y[k] = y[k0 + (k < nr ? k : 0)]; }
        switch (nr) {
            case 1: launch<V, CODES, 1>(s, grid, n, ns, alpha, append, (int)w, b, deltas, values, cp, cc, cv, io, t, pool, blocks); break;
            case 2: launch<V, CODES, 2>(s, grid, n, ns, alpha, append, (int)w, b, deltas, values, cp, cc, cv, io, t, pool, blocks); break;
            case 3: launch<V, CODES, 3>(s, grid, n, ns, alpha, append, (int)w, b, deltas, values, cp, cc, cv, io, t, pool, blocks); break;
            default: launch<V, CODES, 4>(s, grid, n, ns, alpha, append, (int)w, b, deltas, values, cp, cc, cv, io, t, pool, blocks);
        }
        VEXHIP_LAUNCH_CHECK();
    }
    return 0;
}

} // namespace
} // namespace vexhip

using namespace vexhip;

extern "C" {

int vexhip_spmm_sell8_f64_i32(int dev, void *stream, int64_t n, int nrhs, double alpha, int append, int64_t w,
        const void *buf, const int32_t *deltas, const int32_t *cp, const int32_t *cc, const double *cv,
        const double *const *x, double *const *y, const vexhip_traversal *traversal)
{ return spmm<double, 1>(dev, stream, n, nrhs, alpha, append, w, buf, deltas, nullptr, cp, cc, cv, x, y, traversal); }

int vexhip_spmm_sell8_f32_i32(int dev, void *stream, int64_t n, int nrhs, float alpha, int append, int64_t w,
        const void *buf, const int32_t *deltas, const int32_t *cp, const int32_t *cc, const float *cv,
        const float *const *x, float *const *y, const vexhip_traversal *traversal)
{ return spmm<float, 1>(dev, stream, n, nrhs, alpha, append, w, buf, deltas, nullptr, cp, cc, cv, x, y, traversal); }

int vexhip_spmm_sell_f64_i32(int dev, void *stream, int64_t n, int nrhs, double alpha, int append, int64_t w,
        const void *sell, const int32_t *cp, const int32_t *cc, const double *cv,
        const double *const *x, double *const *y, const vexhip_traversal *traversal)
{ return spmm<double, 0>(dev, stream, n, nrhs, alpha, append, w, sell, nullptr, nullptr, cp, cc, cv, x, y, traversal); }

int vexhip_spmm_sell_f32_i32(int dev, void *stream, int64_t n, int nrhs, float alpha, int append, int64_t w,
        const void *sell, const int32_t *cp, const int32_t *cc, const float *cv,
        const float *const *x, float *const *y, const vexhip_traversal *traversal)
{ return spmm<float, 0>(dev, stream, n, nrhs, alpha, append, w, sell, nullptr, nullptr, cp, cc, cv, x, y, traversal); }

int vexhip_spmm_sell8v_f64_i32(int dev, void *stream, int64_t n, int nrhs, double alpha, int append, int64_t w,
        const void *buf, const int32_t *deltas, const double *values, const int32_t *cp, const int32_t *cc, const double *cv,
        const double *const *x, double *const *y, const vexhip_traversal *traversal)
{ return spmm<double, 2>(dev, stream, n, nrhs, alpha, append, w, buf, deltas, values, cp, cc, cv, x, y, traversal); }

int vexhip_spmm_sell8v_f32_i32(int dev, void *stream, int64_t n, int nrhs, float alpha, int append, int64_t w,
        const void *buf, const int32_t *deltas, const float *values, const int32_t *cp, const int32_t *cc, const float *cv,
        const float *const *x, float *const *y, const vexhip_traversal *traversal)
{ return spmm<float, 2>(dev, stream, n, nrhs, alpha, append, w, buf, deltas, values, cp, cc, cv, x, y, traversal); }

int vexhip_spmm_sell8_dict_f64_i32(int dev, void *stream, int64_t n, int nrhs, double alpha, int append, int64_t w,
        const void *buf, const void *pool, const int32_t *blocks, const int32_t *deltas, const int32_t *cp, const int32_t *cc, const double *cv,
        const double *const *x, double *const *y, const vexhip_traversal *traversal)
{ return spmm<double, 1>(dev, stream, n, nrhs, alpha, append, w, buf, deltas, nullptr, cp, cc, cv, x, y, traversal, blocks, pool); }

int vexhip_spmm_sell8_dict_f32_i32(int dev, void *stream, int64_t n, int nrhs, float alpha, int append, int64_t w,
        const void *buf, const void *pool, const int32_t *blocks, const int32_t *deltas, const int32_t *cp, const int32_t *cc, const float *cv,
        const float *const *x, float *const *y, const vexhip_traversal *traversal)
{ return spmm<float, 1>(dev, stream, n, nrhs, alpha, append, w, buf, deltas, nullptr, cp, cc, cv, x, y, traversal, blocks, pool); }

int vexhip_spmm_sell8v_dict_f64_i32(int dev, void *stream, int64_t n, int nrhs, double alpha, int append, int64_t w,
        const void *pool, const int32_t *blocks, const int32_t *deltas, const double *values, const int32_t *cp, const int32_t *cc, const double *cv,
        const double *const *x, double *const *y, const vexhip_traversal *traversal)
{ return spmm<double, 2>(dev, stream, n, nrhs, alpha, append, w, pool, deltas, values, cp, cc, cv, x, y, traversal, blocks); }

int vexhip_spmm_sell8v_dict_f32_i32(int dev, void *stream, int64_t n, int nrhs, float alpha, int append, int64_t w,
        const void *pool, const int32_t *blocks, const int32_t *deltas, const float *values, const int32_t *cp, const int32_t *cc, const float *cv,
        const float *const *x, float *const *y, const vexhip_traversal *traversal)
{ return spmm<float, 2>(dev, stream, n, nrhs, alpha, append, w, pool, deltas, values, cp, cc, cv, x, y, traversal, blocks); }

} // extern "C"

VEXHIP_WARM_TU(spmm)
