// vex::inclusive_scan / vex::exclusive_scan with vex::plus on gfx950
// (vexcl/scan.hpp:66-414).  Reduce-then-scan: (A) one pass of tile sums,
// (B) the same scan applied recursively to the tile sums, (C) a second pass
// that scans every tile from its exclusive offset.  16-byte coalesced loads
// and stores, wave-64 shuffle scans, one LDS hop per 256-lane slab.
// Traffic: 2 reads + 1 write per element (the reference's three kernels read
// the input twice as well, scan.hpp:378-411).  Integer types take the
// single-pass decoupled look-back kernel below instead (1 read + 1 write).
#include "common.hpp"

#include <algorithm>
#include <type_traits>
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

// NT: non-temporal access for data that is touched exactly once (the single-pass scan):
// copy at 1e8 fp64, 16 B per lane: 5.4 -> 6.0 TB/s (tools/bw_probe.py)
template <typename T, bool NT = false>
__device__ __forceinline__ void load_vec(const T *in, long long v, long long n, int vec_ok, T (&x)[cfg<T>::VN]) {
    constexpr int VN = cfg<T>::VN;
    long long e = v * VN;
    if (vec_ok && e + VN <= n) {
        typedef typename cfg<T>::vtype vt;
        const vt *p = reinterpret_cast<const vt *>(in + e);
        vt q;
        if constexpr (NT) q = __builtin_nontemporal_load(p); else q = *p;
#pragma unroll
        for (int j = 0; j < VN; ++j) x[j] = q[j];
    } else {
#pragma unroll
        for (int j = 0; j < VN; ++j) x[j] = (e + j < n) ? in[e + j] : T(0);
    }
}

template <typename T, bool NT = false>
__device__ __forceinline__ void store_vec(T *out, long long v, long long n, int vec_ok, const T (&x)[cfg<T>::VN]) {
    constexpr int VN = cfg<T>::VN;
    long long e = v * VN;
    if (vec_ok && e + VN <= n) {
        typename cfg<T>::vtype q;
#pragma unroll
        for (int j = 0; j < VN; ++j) q[j] = x[j];
        if constexpr (NT) __builtin_nontemporal_store(q, reinterpret_cast<typename cfg<T>::vtype *>(out + e));
        else *reinterpret_cast<typename cfg<T>::vtype *>(out + e) = q;
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
template <typename T, int WAVES = SWAVES>
__device__ __forceinline__ T block_exclusive(T mine, T &total, T *s_wave) {
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    T inc = wave_inclusive(mine);
    if (lane == kWave - 1) s_wave[wave] = inc;
    __syncthreads();
    T off = T(0), tot = T(0);
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
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


// ---------------------------------------------------------------------------
// Single-pass scan with decoupled look-back (integer types): one read + one
// write per element.  Tiles are handed out in launch order by an atomic ticket
// (a tile can only wait for tiles that already run: no deadlock whatever the
// dispatch order).  Each tile publishes {value, flag} in ONE naturally aligned
// 8-byte word with a relaxed agent-scope atomic store and predecessors are read
// with relaxed agent-scope atomic loads: value and flag travel together, so no
// fence is needed (MI355X_MICROARCH.md, "8-B agent atomics both sides"); per-XCD
// L2s never serve a stale copy of such words.  64-bit values use two words
// (low/high half, same flag); a torn pair shows different flags and is re-read.
// Integer addition is associative and commutative mod 2^k, so the result does
// not depend on which predecessors happened to be complete; floating-point
// scans keep the deterministic reduce-then-scan path.
// ---------------------------------------------------------------------------
enum : unsigned { ST_INVALID = 0u, ST_AGGREGATE = 1u, ST_INCLUSIVE = 2u };
// A tile only waits for tiles with SMALLER tickets, and a ticket is taken by a workgroup that already runs: the walk
// cannot deadlock, it can only be slow (a predecessor preempted by a debugger, a profiler or another process).  The
// limit is therefore a guard against a bug, minutes away (2^30 polls of ~0.2 us), and hitting it ABORTS the kernel
// (trap -> the next synchronisation reports an error) -- never a truncated prefix that every later tile would consume.
constexpr long long kSpinLimit = 1ll << 30;

__device__ __forceinline__ unsigned long long ld_status(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T> struct tile_status;
template <> struct tile_status<unsigned> {
    static constexpr int WORDS = 1;
    __device__ static void publish(unsigned long long *st, long long tile, unsigned v, unsigned flag) {
        st_status(st + tile, ((unsigned long long)flag << 32) | v);
    }
    __device__ static unsigned read(const unsigned long long *st, long long tile, unsigned &v) {
        unsigned long long w = ld_status(st + tile);
        v = (unsigned)w;
        return (unsigned)(w >> 32);
    }
};
template <> struct tile_status<unsigned long long> {
    static constexpr int WORDS = 2;
    __device__ static void publish(unsigned long long *st, long long tile, unsigned long long v, unsigned flag) {
        st_status(st + 2 * tile, ((unsigned long long)flag << 32) | (unsigned)v);
        st_status(st + 2 * tile + 1, ((unsigned long long)flag << 32) | (unsigned)(v >> 32));
    }
    __device__ static unsigned read(const unsigned long long *st, long long tile, unsigned long long &v) {
        unsigned long long lo = ld_status(st + 2 * tile), hi = ld_status(st + 2 * tile + 1);
        unsigned fl = (unsigned)(lo >> 32), fh = (unsigned)(hi >> 32);
        v = ((hi & 0xffffffffull) << 32) | (lo & 0xffffffffull);
        return fl == fh ? fl : ST_INVALID;
    }
};

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ws[0] = ticket counter, ws[1] = reserved, ws + 2 = tile status words (all zero before launch)
template <typename T, bool EXCLUSIVE, int SK, int BLOCK>
__global__ __launch_bounds__(BLOCK)
void lookback_scan_kernel(const T *in, T *out, long long n, T init, unsigned long long *ws, int vec_ok) {
    __shared__ T s_wave[BLOCK / kWave];
    __shared__ long long s_tile;
    __shared__ T s_prefix;
    constexpr int VN = cfg<T>::VN;
    unsigned long long *status = ws + 2;

    if (threadIdx.x == 0) s_tile = (long long)atomicAdd(&ws[0], 1ull);
    __syncthreads();
    const long long tile = s_tile;
    const long long vbase = tile * BLOCK * SK;

    T x[SK][VN];
#pragma unroll
    for (int k = 0; k < SK; ++k) load_vec<T, true>(in, vbase + k * BLOCK + threadIdx.x, n, vec_ok, x[k]);
    T mine[SK], acc = T(0);
#pragma unroll
    for (int k = 0; k < SK; ++k) {
        mine[k] = T(0);
#pragma unroll
        for (int j = 0; j < VN; ++j) mine[k] += x[k][j];
        acc += mine[k];
    }
    T aggregate;
    (void)block_exclusive<T, BLOCK / kWave>(acc, aggregate, s_wave);

    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    if (wave == 0) {
        if (lane == 0) tile_status<T>::publish(status, tile, aggregate, tile == 0 ? ST_INCLUSIVE : ST_AGGREGATE);
        T exclusive = T(0);
        long long base = tile - 1;
        long long spins = 0;
        while (base >= 0) {
            const long long idx = base - lane;
            T v = T(0);
            unsigned flag = ST_INCLUSIVE;                 // lanes before tile 0 terminate the walk with 0
            if (idx >= 0) flag = tile_status<T>::read(status, idx, v);
            while (__any(flag == ST_INVALID)) {
                __builtin_amdgcn_s_sleep(8);
                if (idx >= 0 && flag == ST_INVALID) flag = tile_status<T>::read(status, idx, v);
                if (++spins > kSpinLimit) __builtin_trap();
            }
            const unsigned long long incl = __ballot(flag == ST_INCLUSIVE);
            if (incl) {
                const int first = __builtin_ctzll(incl);  // nearest predecessor with a complete prefix
                exclusive += wave_sum<T>(lane <= first ? v : T(0));
                break;
            }
            exclusive += wave_sum<T>(v);
            base -= kWave;
        }
        if (lane == 0) {
            if (tile > 0) tile_status<T>::publish(status, tile, exclusive + aggregate, ST_INCLUSIVE);
            s_prefix = exclusive;
        }
    }
    __syncthreads();

    T carry = init + s_prefix;
#pragma unroll
    for (int k = 0; k < SK; ++k) {
        T total;
        T run = carry + block_exclusive<T, BLOCK / kWave>(mine[k], total, s_wave);
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            T v = x[k][j];
            if constexpr (EXCLUSIVE) { x[k][j] = run; run += v; }
            else { run += v; x[k][j] = run; }
        }
        carry += total;
        store_vec<T, true>(out, vbase + k * BLOCK + threadIdx.x, n, vec_ok, x[k]);
    }
}

template <typename T> constexpr bool lookback_type() { return std::is_integral<T>::value; }

template <typename T>
size_t lookback_words(int64_t n) {
    int64_t nt = (n + cfg<T>::TILE - 1) / cfg<T>::TILE;      // sized for the smallest tile
    return 2 + (size_t)nt * tile_status<typename std::conditional<sizeof(T) == 8, unsigned long long, unsigned>::type>::WORDS;
}

int g_scan_lookback = 1;      // 0: force reduce-then-scan (tests / A-B)

template <typename T>
size_t tmp_elems_rts(int64_t n) {
    size_t total = 0;
    while (n > cfg<T>::TILE) {
        int64_t nt = (n + cfg<T>::TILE - 1) / cfg<T>::TILE;
        total += (size_t)((nt + 3) / 4 * 4);          // keep 16-byte alignment per level
        n = nt;
    }
    return total + 4;
}

template <typename T>
size_t tmp_elems(int64_t n) {
    size_t a = tmp_elems_rts<T>(n);
    if constexpr (std::is_integral<T>::value) {
        size_t b = (lookback_words<T>(n) * 8 + sizeof(T) - 1) / sizeof(T) + 4;
        return a > b ? a : b;
    }
    return a;
}

template <typename T>
int scan_impl(hipStream_t s, const T *in, T *out, int64_t n, bool exclusive, T init, T *tmp);

template <typename T>
int scan_lookback(hipStream_t s, const T *in, T *out, int64_t n, bool exclusive, T init, T *tmp) {
    int vec_ok = ((reinterpret_cast<uintptr_t>(in) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    // (vectors per lane, lanes per workgroup): bigger tiles = fewer look-backs
    static const int cfgs[][2] = {{4, 256}, {4, 256}, {8, 256}, {16, 256}, {16, 512}, {8, 1024}, {16, 1024}, {32, 256}};
    // 1 = auto: 32 vectors per lane for 4-byte types (32 Ki elements per tile), 16 for 8-byte types
    const int mode = (g_scan_lookback >= 2 && g_scan_lookback <= 7) ? g_scan_lookback : (sizeof(T) == 4 ? 7 : 3);
    const int lsk = cfgs[mode][0], blk = cfgs[mode][1];
    const int64_t tile = (int64_t)blk * cfg<T>::VN * lsk;
    int64_t nt = (n + tile - 1) / tile;
    VEXHIP_REQUIRE(nt < (1ll << 31), "scan too large");
    unsigned long long *ws = reinterpret_cast<unsigned long long *>(tmp);
    VEXHIP_TRY(hipMemsetAsync(ws, 0, (2 + (size_t)nt * 2) * 8, s));
#define GO(LSK, BLK) do { \
        if (exclusive) lookback_scan_kernel<T, true, LSK, BLK><<<(unsigned)nt, BLK, 0, s>>>(in, out, n, init, ws, vec_ok); \
        else           lookback_scan_kernel<T, false, LSK, BLK><<<(unsigned)nt, BLK, 0, s>>>(in, out, n, init, ws, vec_ok); } while (0)
    switch (mode) {
        case 2: GO(8, 256); break;   case 3: GO(16, 256); break; case 4: GO(16, 512); break;
        case 5: GO(8, 1024); break;  case 6: GO(16, 1024); break; default: GO(32, 256);
    }
#undef GO
    VEXHIP_LAUNCH_CHECK();
    return 0;
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
    if constexpr (std::is_integral<T>::value) {
        // single pass for integers once there are enough tiles to matter
        if (g_scan_lookback && nt >= 64 && (reinterpret_cast<uintptr_t>(tmp) & 7) == 0)
            return scan_lookback<T>(s, in, out, n, exclusive, init, tmp);
    }
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

int vexhip_scan_set_lookback(int enable) { g_scan_lookback = enable; return 0; }

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

VEXHIP_WARM_TU(scan)
