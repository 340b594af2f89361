// vex::inclusive_scan / vex::exclusive_scan with vex::plus on gfx950
// (vexcl/scan.hpp:66-414).  Reduce-then-scan: (A) one pass of tile sums,
// (B) the same scan applied recursively to the tile sums, (C) a second pass
// that scans every tile from its exclusive offset.  16-byte coalesced loads
// and stores, wave-64 shuffle scans, one LDS hop per 256-lane slab.
// Traffic: 2 reads + 1 write per element (the reference's three kernels read
// the input twice as well, scan.hpp:378-411).
#include "common.hpp"

#include <algorithm>
#include <vector>

namespace vexhip {
namespace {

constexpr int SBLOCK = 256;
constexpr int SWAVES = SBLOCK / kWave;
constexpr int SK = 4;                       // 16-byte vectors per lane per tile

template <typename T> struct cfg {
    static constexpr int VN = 16 / sizeof(T);
    static constexpr int TILE = SBLOCK * VN * SK;
    typedef T vtype __attribute__((ext_vector_type(16 / sizeof(T))));
};

template <typename T>
__device__ __forceinline__ void load_vec(const T *in, long long v, long long n, int vec_ok, T (&x)[cfg<T>::VN]) {
    constexpr int VN = cfg<T>::VN;
    long long e = v * VN;
    if (vec_ok && e + VN <= n) {
        typename cfg<T>::vtype q = *reinterpret_cast<const typename cfg<T>::vtype *>(in + e);
#pragma unroll
        for (int j = 0; j < VN; ++j) x[j] = q[j];
    } else {
#pragma unroll
        for (int j = 0; j < VN; ++j) x[j] = (e + j < n) ? in[e + j] : T(0);
    }
}

template <typename T>
__device__ __forceinline__ void store_vec(T *out, long long v, long long n, int vec_ok, const T (&x)[cfg<T>::VN]) {
    constexpr int VN = cfg<T>::VN;
    long long e = v * VN;
    if (vec_ok && e + VN <= n) {
        typename cfg<T>::vtype q;
#pragma unroll
        for (int j = 0; j < VN; ++j) q[j] = x[j];
        *reinterpret_cast<typename cfg<T>::vtype *>(out + e) = q;
    } else {
#pragma unroll
        for (int j = 0; j < VN; ++j) if (e + j < n) out[e + j] = x[j];
    }
}

template <typename T>
__device__ __forceinline__ T wave_inclusive(T v) {
    const int lane = threadIdx.x % kWave;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        T u = __shfl_up(v, off, 64);
        if (lane >= off) v += u;
    }
    return v;
}

// exclusive prefix of `mine` over the workgroup; total = sum over the workgroup
template <typename T>
__device__ __forceinline__ T block_exclusive(T mine, T &total, T *s_wave) {
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    T inc = wave_inclusive(mine);
    if (lane == kWave - 1) s_wave[wave] = inc;
    __syncthreads();
    T off = T(0), tot = T(0);
#pragma unroll
    for (int w = 0; w < SWAVES; ++w) {
        T s = s_wave[w];
        if (w < wave) off += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return off + (inc - mine);
}

template <typename T>
__global__ __launch_bounds__(SBLOCK)
void tile_sum_kernel(const T *__restrict__ in, long long n, T *__restrict__ sums, int vec_ok) {
    __shared__ T s_wave[SWAVES];
    constexpr int VN = cfg<T>::VN;
    const long long vbase = (long long)blockIdx.x * SBLOCK * SK;
    T acc = T(0);
    T x[SK][VN];
#pragma unroll
    for (int k = 0; k < SK; ++k) load_vec<T>(in, vbase + k * SBLOCK + threadIdx.x, n, vec_ok, x[k]);
#pragma unroll
    for (int k = 0; k < SK; ++k)
#pragma unroll
        for (int j = 0; j < VN; ++j) acc += x[k][j];
    T total;
    (void)block_exclusive(acc, total, s_wave);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// offsets == nullptr: single-tile scan starting from `init`.
template <typename T, bool EXCLUSIVE>
__global__ __launch_bounds__(SBLOCK)
void tile_scan_kernel(const T *in, T *out, long long n, const T *__restrict__ offsets, T init, int vec_ok) {
    __shared__ T s_wave[SWAVES];
    constexpr int VN = cfg<T>::VN;
    const long long vbase = (long long)blockIdx.x * SBLOCK * SK;
    T carry = init;
    if (offsets) carry += offsets[blockIdx.x];
    T x[SK][VN];
#pragma unroll
    for (int k = 0; k < SK; ++k) load_vec<T>(in, vbase + k * SBLOCK + threadIdx.x, n, vec_ok, x[k]);
#pragma unroll
    for (int k = 0; k < SK; ++k) {
        T mine = T(0);
#pragma unroll
        for (int j = 0; j < VN; ++j) mine += x[k][j];
        T total;
        T run = carry + block_exclusive(mine, total, s_wave);
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            T v = x[k][j];
            if constexpr (EXCLUSIVE) { x[k][j] = run; run += v; }
            else { run += v; x[k][j] = run; }
        }
        carry += total;
        store_vec<T>(out, vbase + k * SBLOCK + threadIdx.x, n, vec_ok, x[k]);
    }
}

template <typename T>
size_t tmp_elems(int64_t n) {
    size_t total = 0;
    while (n > cfg<T>::TILE) {
        int64_t nt = (n + cfg<T>::TILE - 1) / cfg<T>::TILE;
        total += (size_t)((nt + 3) / 4 * 4);          // keep 16-byte alignment per level
        n = nt;
    }
    return total + 4;
}

template <typename T>
int scan_impl(hipStream_t s, const T *in, T *out, int64_t n, bool exclusive, T init, T *tmp) {
    if (n <= 0) return 0;
    int vec_ok = ((reinterpret_cast<uintptr_t>(in) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    constexpr int64_t TILE = cfg<T>::TILE;
    int64_t nt = (n + TILE - 1) / TILE;
    if (nt == 1) {
        if (exclusive) tile_scan_kernel<T, true><<<1, SBLOCK, 0, s>>>(in, out, n, nullptr, init, vec_ok);
        else           tile_scan_kernel<T, false><<<1, SBLOCK, 0, s>>>(in, out, n, nullptr, init, vec_ok);
        VEXHIP_LAUNCH_CHECK();
        return 0;
    }
    VEXHIP_REQUIRE(nt < (1ll << 31), "scan too large");
    tile_sum_kernel<T><<<(unsigned)nt, SBLOCK, 0, s>>>(in, n, tmp, vec_ok);
    VEXHIP_LAUNCH_CHECK();
    if (int rc = scan_impl<T>(s, tmp, tmp, nt, true, T(0), tmp + (nt + 3) / 4 * 4)) return rc;
    if (exclusive) tile_scan_kernel<T, true><<<(unsigned)nt, SBLOCK, 0, s>>>(in, out, n, tmp, init, vec_ok);
    else           tile_scan_kernel<T, false><<<(unsigned)nt, SBLOCK, 0, s>>>(in, out, n, tmp, init, vec_ok);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

} // namespace

int scan_exclusive_i32_internal(int dev, hipStream_t s, const int *in, int *out, int64_t n) {
    VEXHIP_SET_DEVICE(dev);
    unsigned *tmp = nullptr;
    VEXHIP_TRY(hipMalloc(&tmp, sizeof(unsigned) * tmp_elems<unsigned>(n)));
    int rc = scan_impl<unsigned>(s, reinterpret_cast<const unsigned *>(in), reinterpret_cast<unsigned *>(out), n, true, 0u, tmp);
    if (!rc) rc = check(hipStreamSynchronize(s), __FILE__, __LINE__);
    (void)hipFree(tmp);
    return rc;
}

// used by sort.hip for the digit tables
int scan_exclusive_u32_tmp(hipStream_t s, const unsigned *in, unsigned *out, int64_t n, unsigned *tmp) {
    return scan_impl<unsigned>(s, in, out, n, true, 0u, tmp);
}
size_t scan_tmp_elems_u32(int64_t n) { return tmp_elems<unsigned>(n); }

} // namespace vexhip

using namespace vexhip;

extern "C" {

size_t vexhip_scan_tmp_bytes(int dtype, int64_t n) {
    switch (dtype) {
        case VEXHIP_F64: return sizeof(double) * tmp_elems<double>(n);
        case VEXHIP_F32: return sizeof(float) * tmp_elems<float>(n);
        case VEXHIP_I32: case VEXHIP_U32: return sizeof(unsigned) * tmp_elems<unsigned>(n);
        case VEXHIP_I64: case VEXHIP_U64: return sizeof(unsigned long long) * tmp_elems<unsigned long long>(n);
    }
    return 0;
}

int vexhip_scan(int dev, void *stream, int dtype, int exclusive, const void *init_host,
        const void *in, void *out, int64_t n, void *tmp)
{
    VEXHIP_REQUIRE(n >= 0, "negative size");
    if (n == 0) return 0;
    VEXHIP_REQUIRE(in && out && tmp, "NULL argument");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    // signed integers are scanned as unsigned: identical bits, no UB on wrap
    switch (dtype) {
        case VEXHIP_F64: return scan_impl<double>(s, (const double *)in, (double *)out, n, exclusive != 0,
                                 (exclusive && init_host) ? *(const double *)init_host : 0.0, (double *)tmp);
        case VEXHIP_F32: return scan_impl<float>(s, (const float *)in, (float *)out, n, exclusive != 0,
                                 (exclusive && init_host) ? *(const float *)init_host : 0.0f, (float *)tmp);
        case VEXHIP_I32: case VEXHIP_U32:
            return scan_impl<unsigned>(s, (const unsigned *)in, (unsigned *)out, n, exclusive != 0,
                                 (exclusive && init_host) ? *(const unsigned *)init_host : 0u, (unsigned *)tmp);
        case VEXHIP_I64: case VEXHIP_U64:
            return scan_impl<unsigned long long>(s, (const unsigned long long *)in, (unsigned long long *)out, n, exclusive != 0,
                                 (exclusive && init_host) ? *(const unsigned long long *)init_host : 0ull, (unsigned long long *)tmp);
    }
    return fail(__FILE__, __LINE__, "unknown dtype");
}

} // extern "C"
