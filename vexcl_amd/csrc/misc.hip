// Setup-side kernels: CSR -> hybrid-ELL conversion (sparse/ell.hpp:348-508,
// width rule spmat/hybrid_ell.inl:66-114), the benchmark's 3-D Poisson matrix
// built directly in HBM (examples/benchmark.cpp:364-415), and fills.
// None of this is on the timed path.
#include "common.hpp"

#include <algorithm>
#include <vector>

namespace vexhip {

// scan.hip
int scan_exclusive_i32_internal(int dev, hipStream_t s, const int *in, int *out, int64_t n);

namespace {

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <typename T> __device__ __forceinline__ T hash_value(unsigned long long h);
template <> __device__ __forceinline__ unsigned hash_value<unsigned>(unsigned long long h) { return (unsigned)(h >> 32); }
template <> __device__ __forceinline__ int hash_value<int>(unsigned long long h) { return (int)(h >> 32); }
template <> __device__ __forceinline__ unsigned long long hash_value<unsigned long long>(unsigned long long h) { return h; }
template <> __device__ __forceinline__ long long hash_value<long long>(unsigned long long h) { return (long long)h; }
template <> __device__ __forceinline__ double hash_value<double>(unsigned long long h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }
template <> __device__ __forceinline__ float hash_value<float>(unsigned long long h) { return (float)(h >> 40) * (1.0f / 16777216.0f); }

template <typename T>
__global__ __launch_bounds__(256)
void fill_hash_kernel(unsigned long long seed, T *out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = hash_value<T>(mix64(seed + (unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ull));
}

template <typename T>
__global__ __launch_bounds__(256)
void fill_value_kernel(T v, T *out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = v;
}

// ---- Poisson -------------------------------------------------------------
// number of interior coordinates strictly below t (interior = 1..n-2)
__device__ __host__ inline long long interior_below(long long t, long long n) {
    long long v = t - 1;
    if (v < 0) v = 0;
    if (v > n - 2) v = n - 2;
    return v;
}

// nnz in rows [0, idx): every row has 1 entry, interior rows 6 more
__device__ __host__ inline long long poisson_nnz_before(long long idx, long long n) {
    if (n < 3) return idx;
    long long N = n * n * n;
    if (idx >= N) { long long in = (n - 2) * (n - 2) * (n - 2); return N + 6 * in; }
    long long i = idx % n, j = (idx / n) % n, k = idx / (n * n);
    long long m = n - 2;
    long long cnt = interior_below(k, n) * m * m;
    if (k >= 1 && k <= n - 2) {
        cnt += interior_below(j, n) * m;
        if (j >= 1 && j <= n - 2) cnt += interior_below(i, n);
    }
    return idx + 6 * cnt;
}

// Coefficient of the face between grid points `lo` and `lo + step` (axis = 0, 1, 2) of the variable-coefficient
// operator below: 0.5 + u, u in [0, 1) from a counter hash of (seed, lo, axis).  Integer arithmetic and one
// exactly rounded conversion, so the host restatement the tests compare with (vxo_diffusion3d_*) gives the same bits.
__device__ __forceinline__ double face_coefficient(unsigned long long seed, long long lo, int axis) {
    const unsigned long long h = mix64(seed + ((unsigned long long)lo * 3ull + (unsigned long long)axis + 1ull) * 0x9E3779B97F4A7C15ull);
    return 0.5 + (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

// VAR = false: the benchmark's Poisson matrix.  VAR = true: the same 7-point pattern with a different coefficient on
// every face -- -div(k grad u), k in [0.5, 1.5) -- i.e. about 4 N distinct values: what a finite-volume code assembles, and
// the matrix no value coding applies to (bench.py's "variable coefficient" row).  Symmetric; boundary rows identity.
template <typename V, bool VAR, typename P = int>
__global__ __launch_bounds__(256)
void poisson_kernel(long long n, long long row_begin, long long row_end, unsigned long long seed,
        P *__restrict__ ptr, int *__restrict__ col, V *__restrict__ val)
{
    const long long nn = n * n;
    const V h2i = (V)((double)(n - 1) * (double)(n - 1));
    const long long base = poisson_nnz_before(row_begin, n);
    for (long long idx = row_begin + (long long)blockIdx.x * blockDim.x + threadIdx.x; idx <= row_end;
         idx += (long long)gridDim.x * blockDim.x) {
        long long p = poisson_nnz_before(idx, n) - base;
        ptr[idx - row_begin] = (P)p;
        if (idx == row_end) break;
        long long i = idx % n, j = (idx / n) % n, k = idx / nn;
        bool bnd = (i == 0 || i == n - 1 || j == 0 || j == n - 1 || k == 0 || k == n - 1);
        if (bnd) {
            col[p] = (int)idx; val[p] = V(1);
        } else if constexpr (VAR) {
            const double k0 = face_coefficient(seed, idx - nn, 2), k1 = face_coefficient(seed, idx - n, 1), k2 = face_coefficient(seed, idx - 1, 0);
            const double k4 = face_coefficient(seed, idx, 0), k5 = face_coefficient(seed, idx, 1), k6 = face_coefficient(seed, idx, 2);
            col[p + 0] = (int)(idx - nn); val[p + 0] = (V)(-(double)h2i * k0);
            col[p + 1] = (int)(idx - n);  val[p + 1] = (V)(-(double)h2i * k1);
            col[p + 2] = (int)(idx - 1);  val[p + 2] = (V)(-(double)h2i * k2);
            col[p + 3] = (int)(idx);      val[p + 3] = (V)((double)h2i * (((((k0 + k1) + k2) + k4) + k5) + k6));
            col[p + 4] = (int)(idx + 1);  val[p + 4] = (V)(-(double)h2i * k4);
            col[p + 5] = (int)(idx + n);  val[p + 5] = (V)(-(double)h2i * k5);
            col[p + 6] = (int)(idx + nn); val[p + 6] = (V)(-(double)h2i * k6);
        } else {
            col[p + 0] = (int)(idx - nn); val[p + 0] = -h2i;
            col[p + 1] = (int)(idx - n);  val[p + 1] = -h2i;
            col[p + 2] = (int)(idx - 1);  val[p + 2] = -h2i;
            col[p + 3] = (int)(idx);      val[p + 3] = 6 * h2i;
            col[p + 4] = (int)(idx + 1);  val[p + 4] = -h2i;
            col[p + 5] = (int)(idx + n);  val[p + 5] = -h2i;
            col[p + 6] = (int)(idx + nn); val[p + 6] = -h2i;
        }
    }
}

// ---- hybrid ELL analysis / fill -------------------------------------------
template <typename P>
__global__ __launch_bounds__(256)
void width_max_kernel(long long n, const P *__restrict__ ptr, int *maxw) {
    int m = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        int w = (int)(ptr[i + 1] - ptr[i]);
        m = w > m ? w : m;
    }
    for (int o = 32; o > 0; o >>= 1) { int v = __shfl_down(m, o, 64); m = v > m ? v : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(maxw, m);
}

// histogram of min(width, cap) -- cap bucket collects everything wider
template <typename P>
__global__ __launch_bounds__(256)
void width_hist_kernel(long long n, const P *__restrict__ ptr, int cap, unsigned long long *hist) {
    // widths below 256 (every practical matrix) are counted in LDS first: a
    // regular matrix would otherwise serialise all rows on one global atomic
    __shared__ unsigned s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        int w = (int)(ptr[i + 1] - ptr[i]);
        if (w > cap) w = cap;
        if (w < 256) atomicAdd(&s_h[w], 1u);
        else atomicAdd(&hist[w], 1ull);
    }
    __syncthreads();
    if (s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s_h[threadIdx.x]);
}

template <typename P>
__global__ __launch_bounds__(256)
void tail_count_kernel(long long n, const P *__restrict__ ptr, int w, int *cnt, unsigned long long *total) {
    unsigned long long local = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        int rw = (int)(ptr[i + 1] - ptr[i]);
        int t = rw > w ? rw - w : 0;
        if (cnt) cnt[i] = t;
        local += (unsigned long long)t;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
    if (total && (threadIdx.x & 63) == 0 && local) atomicAdd(total, local);
}

template <typename V, typename P>
__global__ __launch_bounds__(256)
void hell_fill_kernel(long long n, long long pitch, int w,
        const P *__restrict__ ptr, const int *__restrict__ col, const V *__restrict__ val,
        int *__restrict__ ell_col, V *__restrict__ ell_val,
        const int *__restrict__ csr_ptr, int *__restrict__ csr_col, V *__restrict__ csr_val)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pitch;
         i += (long long)gridDim.x * blockDim.x) {
        P b = 0, e = 0;
        if (i < n) { b = ptr[i]; e = ptr[i + 1]; }
        if (ell_col) {                       // NULL: only the CSR tail is wanted (SELL storage)
            int j = 0;
            for (; j < w && b + j < e; ++j) {
                ell_col[i + j * pitch] = col[b + j];
                ell_val[i + j * pitch] = val[b + j];
            }
            for (; j < w; ++j) {
                ell_col[i + j * pitch] = -1;
                ell_val[i + j * pitch] = V(0);
            }
        }
        if (csr_ptr && i < n) {
            int o = csr_ptr[i];
            for (P q = b + w; q < e; ++q, ++o) { csr_col[o] = col[q]; csr_val[o] = val[q]; }
        }
    }
}

__global__ void set_last_kernel(int *csr_ptr, long long n, const int *cnt) {
    // exclusive scan leaves csr_ptr[n-1]; close the array
    csr_ptr[n] = csr_ptr[n - 1] + cnt[n - 1];
}

inline int grid_for(int dev, int64_t n) {
    return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, (int64_t)info(dev).cus * 16));
}

template <typename V, typename P>
int hell_fill(int dev, void *stream, int64_t n, const P *ptr, const int *col, const V *val,
        int64_t w, int64_t pitch, int *ell_col, V *ell_val, int *csr_ptr, int *csr_col, V *csr_val)
{
    VEXHIP_REQUIRE(n >= 0 && w >= 0 && pitch >= n, "bad ELL geometry");
    if (n == 0) return 0;
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    if (csr_ptr) {
        int *cnt = nullptr;
        VEXHIP_TRY(hipMalloc(&cnt, sizeof(int) * (size_t)n));
        tail_count_kernel<P><<<grid_for(dev, n), 256, 0, s>>>(n, ptr, (int)w, cnt, nullptr);
        int rc = scan_exclusive_i32_internal(dev, s, cnt, csr_ptr, n);
        if (rc) { (void)hipFree(cnt); return rc; }
        set_last_kernel<<<1, 1, 0, s>>>(csr_ptr, n, cnt);
        hell_fill_kernel<V, P><<<grid_for(dev, pitch), 256, 0, s>>>(n, pitch, (int)w, ptr, col, val,
                ell_col, ell_val, csr_ptr, csr_col, csr_val);
        VEXHIP_LAUNCH_CHECK();
        VEXHIP_TRY(hipStreamSynchronize(s));
        VEXHIP_TRY(hipFree(cnt));
    } else {
        hell_fill_kernel<V, P><<<grid_for(dev, pitch), 256, 0, s>>>(n, pitch, (int)w, ptr, col, val,
                ell_col, ell_val, nullptr, nullptr, nullptr);
        VEXHIP_LAUNCH_CHECK();
    }
    return 0;
}

template <typename P>
int hell_analyze(int dev, void *stream, int64_t n, const P *ptr,
        int64_t *ell_width, int64_t *tail_nnz)
{
    VEXHIP_REQUIRE(ell_width && tail_nnz, "NULL output");
    *ell_width = 0; *tail_nnz = 0;
    if (n <= 0) return 0;
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    const int cap = 4096;                      // widths above this share one bucket
    unsigned long long *d = nullptr;           // [0] max (as int), [1] tail total, [2..] histogram
    size_t bytes = sizeof(unsigned long long) * (size_t)(cap + 3);
    setup_trace trace(s);
    VEXHIP_TRY(hipMalloc(&d, bytes));
    trace.mark("  width: hipMalloc");
    VEXHIP_TRY(hipMemsetAsync(d, 0, bytes, s));
    trace.mark("  width: memset");
    width_max_kernel<P><<<grid_for(dev, n), 256, 0, s>>>(n, ptr, reinterpret_cast<int *>(d));
    trace.mark("  width: max kernel");
    width_hist_kernel<P><<<grid_for(dev, n), 256, 0, s>>>(n, ptr, cap, d + 2);
    trace.mark("  width: hist kernel");
    std::vector<unsigned long long> h(cap + 3);
    VEXHIP_TRY(hipMemcpyAsync(h.data(), d, bytes, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    trace.mark("  width: D2H");
    int64_t maxw = (int64_t)(int)(h[0] & 0xffffffffu);
    // hybrid_ell.inl:103-110: smallest i with 3 * (#rows wider than i) < n
    const double ell_vs_csr = 3.0;
    int64_t w = maxw, rows = n;
    for (int64_t i = 0; i < maxw && i <= cap; ++i) {
        rows -= (int64_t)h[2 + i];
        if (ell_vs_csr * (double)rows < (double)n) { w = i; break; }
    }
    if (w < maxw) {
        tail_count_kernel<P><<<grid_for(dev, n), 256, 0, s>>>(n, ptr, (int)w, nullptr, d + 1);
        unsigned long long t = 0;
        VEXHIP_TRY(hipMemcpyAsync(&t, d + 1, sizeof(t), hipMemcpyDeviceToHost, s));
        VEXHIP_TRY(hipStreamSynchronize(s));
        *tail_nnz = (int64_t)t;
    }
    VEXHIP_TRY(hipFree(d));
    trace.mark("  width: hipFree");
    *ell_width = w;
    return 0;
}


} // namespace

// 64-bit row pointers: internal entry points for spmat.hip (the tail CSR keeps 32-bit pointers: it must stay below 2^31 entries)
int hell_analyze_p64(int dev, void *stream, int64_t n, const long long *ptr, int64_t *ell_width, int64_t *tail_nnz)
{ return hell_analyze<long long>(dev, stream, n, ptr, ell_width, tail_nnz); }
int hell_tail_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const double *val, int64_t w, int32_t *csr_ptr, int32_t *csr_col, double *csr_val)
{ return hell_fill<double, long long>(dev, stream, n, ptr, col, val, w, (n + 15) / 16 * 16, nullptr, nullptr, csr_ptr, csr_col, csr_val); }
int hell_tail_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const float *val, int64_t w, int32_t *csr_ptr, int32_t *csr_col, float *csr_val)
{ return hell_fill<float, long long>(dev, stream, n, ptr, col, val, w, (n + 15) / 16 * 16, nullptr, nullptr, csr_ptr, csr_col, csr_val); }

} // namespace vexhip

using namespace vexhip;

extern "C" {

int64_t vexhip_poisson3d_nnz(int64_t n) { return poisson_nnz_before(n * n * n, n); }

int64_t vexhip_poisson3d_strip_nnz(int64_t n, int64_t rb, int64_t re) {
    return poisson_nnz_before(re, n) - poisson_nnz_before(rb, n);
}

int vexhip_poisson3d_strip_f64_i32(int dev, void *stream, int64_t n, int64_t rb, int64_t re,
        int32_t *ptr, int32_t *col, double *val)
{
    VEXHIP_REQUIRE(n >= 1 && rb >= 0 && re >= rb && re <= n * n * n, "bad Poisson strip");
    VEXHIP_REQUIRE(n * n * n < (1ll << 31) && vexhip_poisson3d_strip_nnz(n, rb, re) < (1ll << 31),
                   "Poisson problem too large for int32 indices");
    VEXHIP_SET_DEVICE(dev);
    poisson_kernel<double, false><<<grid_for(dev, re - rb + 1), 256, 0, as_stream(stream)>>>(n, rb, re, 0ull, ptr, col, val);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

/* the same strip with 64-bit row pointers: 2^31 entries and more (columns stay 32-bit: n^3 < 2^31) */
int vexhip_poisson3d_strip_f64_p64(int dev, void *stream, int64_t n, int64_t rb, int64_t re,
        int64_t *ptr, int32_t *col, double *val)
{
    VEXHIP_REQUIRE(n >= 1 && rb >= 0 && re >= rb && re <= n * n * n, "bad Poisson strip");
    VEXHIP_REQUIRE(n * n * n < (1ll << 31), "Poisson problem too large for int32 column indices");
    VEXHIP_SET_DEVICE(dev);
    poisson_kernel<double, false, long long><<<grid_for(dev, re - rb + 1), 256, 0, as_stream(stream)>>>(n, rb, re, 0ull,
            reinterpret_cast<long long *>(ptr), col, val);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

int vexhip_diffusion3d_strip_f64_i32(int dev, void *stream, int64_t n, int64_t rb, int64_t re, uint64_t seed,
        int32_t *ptr, int32_t *col, double *val)
{
    VEXHIP_REQUIRE(n >= 1 && rb >= 0 && re >= rb && re <= n * n * n, "bad strip");
    VEXHIP_REQUIRE(n * n * n < (1ll << 31) && vexhip_poisson3d_strip_nnz(n, rb, re) < (1ll << 31),
                   "problem too large for int32 indices");
    VEXHIP_SET_DEVICE(dev);
    poisson_kernel<double, true><<<grid_for(dev, re - rb + 1), 256, 0, as_stream(stream)>>>(n, rb, re, (unsigned long long)seed, ptr, col, val);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

int vexhip_poisson3d_csr_f64_i32(int dev, void *stream, int64_t n, int32_t *ptr, int32_t *col, double *val) {
    return vexhip_poisson3d_strip_f64_i32(dev, stream, n, 0, n * n * n, ptr, col, val);
}

int vexhip_fill_hash(int dev, void *stream, int dtype, uint64_t seed, void *out, int64_t n) {
    if (n <= 0) return 0;
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    int g = grid_for(dev, n);
    switch (dtype) {
        case VEXHIP_F64: fill_hash_kernel<double><<<g, 256, 0, s>>>(seed, (double *)out, n); break;
        case VEXHIP_F32: fill_hash_kernel<float><<<g, 256, 0, s>>>(seed, (float *)out, n); break;
        case VEXHIP_I32: fill_hash_kernel<int><<<g, 256, 0, s>>>(seed, (int *)out, n); break;
        case VEXHIP_U32: fill_hash_kernel<unsigned><<<g, 256, 0, s>>>(seed, (unsigned *)out, n); break;
        case VEXHIP_I64: fill_hash_kernel<long long><<<g, 256, 0, s>>>(seed, (long long *)out, n); break;
        case VEXHIP_U64: fill_hash_kernel<unsigned long long><<<g, 256, 0, s>>>(seed, (unsigned long long *)out, n); break;
        default: return fail(__FILE__, __LINE__, "unknown dtype");
    }
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

int vexhip_fill_value(int dev, void *stream, int dtype, const void *v, void *out, int64_t n) {
    if (n <= 0) return 0;
    VEXHIP_REQUIRE(v, "value is NULL");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    int g = grid_for(dev, n);
    switch (dtype) {
        case VEXHIP_F64: fill_value_kernel<double><<<g, 256, 0, s>>>(*(const double *)v, (double *)out, n); break;
        case VEXHIP_F32: fill_value_kernel<float><<<g, 256, 0, s>>>(*(const float *)v, (float *)out, n); break;
        case VEXHIP_I32: case VEXHIP_U32:
            fill_value_kernel<unsigned><<<g, 256, 0, s>>>(*(const unsigned *)v, (unsigned *)out, n); break;
        case VEXHIP_I64: case VEXHIP_U64:
            fill_value_kernel<unsigned long long><<<g, 256, 0, s>>>(*(const unsigned long long *)v, (unsigned long long *)out, n); break;
        default: return fail(__FILE__, __LINE__, "unknown dtype");
    }
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

int vexhip_hell_analyze_i32(int dev, void *stream, int64_t n, const int32_t *ptr,
        int64_t *ell_width, int64_t *tail_nnz)
{ return hell_analyze<int32_t>(dev, stream, n, ptr, ell_width, tail_nnz); }

int vexhip_hell_fill_f64_i32(int dev, void *stream, int64_t n,
        const int32_t *ptr, const int32_t *col, const double *val,
        int64_t w, int64_t pitch, int32_t *ell_col, double *ell_val,
        int32_t *csr_ptr, int32_t *csr_col, double *csr_val)
{ return hell_fill<double, int32_t>(dev, stream, n, ptr, col, val, w, pitch, ell_col, ell_val, csr_ptr, csr_col, csr_val); }

int vexhip_hell_fill_f32_i32(int dev, void *stream, int64_t n,
        const int32_t *ptr, const int32_t *col, const float *val,
        int64_t w, int64_t pitch, int32_t *ell_col, float *ell_val,
        int32_t *csr_ptr, int32_t *csr_col, float *csr_val)
{ return hell_fill<float, int32_t>(dev, stream, n, ptr, col, val, w, pitch, ell_col, ell_val, csr_ptr, csr_col, csr_val); }

} // extern "C"

VEXHIP_WARM_TU(misc)
