// vex::sort / vex::sort_by_key on gfx950 (vexcl/sort.hpp:2158-2182).
// The reference is a merge sort (sort.hpp:820-1696) and its contract -- pinned
// by tests/sort.cpp:22-45 -- is std::stable_sort.  Here: stable LSD radix sort,
// 8-bit digits.  Per pass:
//   (1) digit histogram per tile (12 288 u32 keys) -> table[digit][tile]
//   (2) exclusive scan of the table (scan.hip)   -> global base per (digit,tile)
//   (3) scatter: wave-64 ballot match ranks the keys of a wave stably, LDS
//       per-wave digit counters order the 16 waves, the tile is re-ordered in
//       LDS and written out as runs of equal digits (coalesced).
// Signed and floating keys are mapped to order-preserving unsigned bits on the
// fly; the stored keys stay untouched.
#include "common.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace vexhip {

int scan_exclusive_u32_tmp(hipStream_t s, const unsigned *in, unsigned *out, int64_t n, unsigned *tmp);
size_t scan_tmp_elems_u32(int64_t n);

namespace {

constexpr int RB = 1024;             // lanes per scatter workgroup (16 waves)
constexpr int RW = RB / kWave;
constexpr int HB = 256;              // lanes per histogram workgroup
constexpr int RADIX = 256;

// Keys per lane.  A tile is re-ordered in LDS, so (key + value) bytes x tile must
// stay near 48 KiB (two workgroups per CU); longer tiles = longer runs of equal
// digits = better coalesced scatter writes (12 288 u32 keys: 48-key runs).
constexpr int keys_per_lane(int key_bytes, int value_bytes) { return 48 / (key_bytes + value_bytes); }

enum { KEY_UNSIGNED = 0, KEY_SIGNED = 1, KEY_FLOAT = 2 };

template <typename K> struct kbits { static constexpr K sign = (K)1 << (sizeof(K) * 8 - 1); };

template <typename K, int MODE, bool DESC>
__device__ __forceinline__ K to_ordered(K k) {
    if constexpr (MODE == KEY_SIGNED) k ^= kbits<K>::sign;
    if constexpr (MODE == KEY_FLOAT)  k = (k & kbits<K>::sign) ? (K)~k : (K)(k ^ kbits<K>::sign);
    if constexpr (DESC) k = (K)~k;
    return k;
}

// One LDS counter bump per key.  Keys with a small range have constant upper digits: a wave
// whose (active) lanes agree on the digit adds their number once instead of serialising up to
// 64 same-address LDS atomics.
__device__ __forceinline__ void count_digit(unsigned *s_h, unsigned d) {
    const unsigned long long act = __ballot(1);
    const unsigned d0 = __builtin_amdgcn_readfirstlane(d);
    if (__ballot(d == d0) == act) {
        if ((int)(threadIdx.x % kWave) == __ffsll((long long)act) - 1) atomicAdd(&s_h[d0], (unsigned)__popcll(act));
    } else {
        atomicAdd(&s_h[d], 1u);
    }
}

template <typename K, int MODE, bool DESC, int KPT, int UNROLL = 6>
__global__ __launch_bounds__(HB)
void radix_hist_kernel(const K *__restrict__ keys, long long n, int shift, unsigned nblocks, unsigned *__restrict__ table, int vec_ok)
{
    constexpr int TILE = RB * KPT;
    constexpr int VN = 16 / (int)sizeof(K);
    typedef K vtype __attribute__((ext_vector_type(16 / sizeof(K))));
    __shared__ unsigned s_h[RADIX];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    // XCD-contiguous tile order (as the scatter): workgroup b runs on XCD b % 8.  A tile writes ONE 4-byte counter into
    // each of 256 table rows; the counters of neighbouring tiles share a cache line, and on the same XCD they meet in one
    // L2 and leave it as full lines instead of 256 partial-line writes per tile.
    const unsigned per = (nblocks + 7) / 8;
    const unsigned tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (tile >= nblocks) return;
    const long long base = (long long)tile * TILE;
    const int count = (int)((n - base < TILE) ? (n - base) : TILE);
    int done = 0;
    if (vec_ok) {
        const int nv = count / VN;
        const vtype *kv = reinterpret_cast<const vtype *>(keys + base);
        // six 16-byte loads in flight per lane before the first counter bump: with one load per trip the kernel sat
        // at 4.3 TB/s with its waves parked 89 % of the time (profiles/r02_sort_sq.txt) -- latency, not HBM
        constexpr int UN = UNROLL;
        int v = threadIdx.x;
        for (; v + (UN - 1) * HB < nv; v += UN * HB) {
            vtype q[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) q[u] = __builtin_nontemporal_load(kv + v + u * HB);
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int j = 0; j < VN; ++j)
                    count_digit(s_h, (unsigned)(to_ordered<K, MODE, DESC>(q[u][j]) >> shift) & (RADIX - 1));
        }
        for (; v < nv; v += HB) {
            vtype q = __builtin_nontemporal_load(kv + v);
#pragma unroll
            for (int j = 0; j < VN; ++j)
                count_digit(s_h, (unsigned)(to_ordered<K, MODE, DESC>(q[j]) >> shift) & (RADIX - 1));
        }
        done = nv * VN;
    }
    for (int i = done + threadIdx.x; i < count; i += HB)
        count_digit(s_h, (unsigned)(to_ordered<K, MODE, DESC>(keys[base + i]) >> shift) & (RADIX - 1));
    __syncthreads();
    table[(size_t)threadIdx.x * nblocks + tile] = s_h[threadIdx.x];
}

template <int VB> struct valtype;
template <> struct valtype<0> { typedef char type; };
template <> struct valtype<4> { typedef unsigned type; };
template <> struct valtype<8> { typedef unsigned long long type; };

// LDS of one scatter workgroup.  `raw` holds the re-ordered tile (keys, then values) -- and,
// while the keys are being ranked (the tile is still in registers), the per-wave digit
// match masks.
template <typename K, int VB, int KPT>
struct scatter_lds {
    static constexpr int TILE = RB * KPT;
    static constexpr int TILE_BYTES = TILE * ((int)sizeof(K) + VB);
    static constexpr int MATCH_BYTES = RW * RADIX * 8;
    static constexpr int RAW_WORDS = ((TILE_BYTES > MATCH_BYTES ? TILE_BYTES : MATCH_BYTES) + 7) / 8;
    unsigned long long raw[RAW_WORDS];
    unsigned hist[RW][RADIX];
    unsigned dstart[RADIX];
    unsigned gbase[RADIX];
    unsigned wtot[RADIX / kWave];
};

// FULL: every slot of the tile holds a key (all tiles but the last): no validity masks.
// VERIFY (with ATOMIC_RANK) and vflag (uniform per workgroup): the tile is ranked by the LDS atomics AND by the match words, which
// do not depend on the order in which the LDS serves the lanes of one atomic; every key's two ranks must agree (trap otherwise).
// One complete tile in SORT_VERIFY_EVERY is such a tile, inside the production launch: every sort checks the lane-order
// property on ~6 % of its keys, all keys of every lane of those tiles.  (A separate launch for these tiles cost 12 % of the
// sort -- 9.96 against 8.90 ms per 1e9 keys: the holes it leaves in the runs of every digit break the write combining.)
template <typename K, int MODE, bool DESC, int VB, int KPT, bool FULL, bool ATOMIC_RANK, bool VERIFY = false>
__device__ __forceinline__ void scatter_tile(scatter_lds<K, VB, KPT> &L, const unsigned tile,
        const K *__restrict__ keys_in, K *__restrict__ keys_out,
        const typename valtype<VB>::type *__restrict__ vals_in, typename valtype<VB>::type *__restrict__ vals_out,
        long long n, int shift, unsigned nblocks, const unsigned *__restrict__ table, const bool vflag = false)
{
    typedef typename valtype<VB>::type VT;
    constexpr int TILE = RB * KPT;
    K *s_keys = reinterpret_cast<K *>(L.raw);
    VT *s_vals = reinterpret_cast<VT *>(reinterpret_cast<char *>(L.raw) + (size_t)TILE * sizeof(K));
    unsigned long long *s_match = L.raw;

    const int t = threadIdx.x, wave = t / kWave, lane = t % kWave;
    const long long base = (long long)tile * TILE;
    const long long wbase = base + (long long)wave * (kWave * KPT);
    const int nvalid = FULL ? TILE : (int)(n - base);

    for (int i = t; i < RW * RADIX; i += RB) {
        (&L.hist[0][0])[i] = 0;
        if (!ATOMIC_RANK || (VERIFY && vflag)) s_match[i] = 0ull; // the match words are only used by the fallback ranking (and the verified tiles)
    }

    K key[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const long long i = wbase + k * kWave + lane;
        key[k] = (FULL || i < n) ? __builtin_nontemporal_load(keys_in + i) : K(0);      // read once per pass
    }
    // payloads are fetched with the keys: their latency hides behind the ranking
    VT val[VB ? KPT : 1];
    if constexpr (VB != 0) {
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const long long i = wbase + k * kWave + lane;
            val[k] = (FULL || i < n) ? __builtin_nontemporal_load(vals_in + i) : VT(0);
        }
    }
    __syncthreads();

    unsigned rd[KPT];          // rank within (wave, digit) | digit << 16; ~0u = padding slot
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const bool valid = FULL || (wbase + k * kWave + lane < n);
        const unsigned d = (unsigned)(to_ordered<K, MODE, DESC>(key[k]) >> shift) & (RADIX - 1);
        const unsigned long long act = FULL ? ~0ull : __ballot(valid);
        const unsigned d0 = __builtin_amdgcn_readfirstlane(d);
        const bool uniform = __ballot(valid && d == d0) == act;      // small key ranges: constant upper digits
        if constexpr (ATOMIC_RANK) {
            // ONE LDS operation per key: the returned old value of the wave's digit counter IS the key's rank within
            // (wave, digit), because gfx950 services the lanes of one LDS atomic that hit the same address in lane order
            // (not an architectural promise: vexhip_sort verifies it on the device before it uses this path,
            // tools/r02_lds_atomic_order.py: 0 violations in 9.4e8 returned values) and a wave's LDS operations execute in
            // program order (k ascending = tile order).  The match-word scheme below needs five conflicting LDS operations
            // per key and left the kernel LDS-bound (500 of 838 LDS-active cycles per wave were bank conflicts).
            unsigned r;
            unsigned expect = 0;                                     // VERIFY: the rank the match words give
            if constexpr (VERIFY) {
                if (vflag && !uniform) {
                    unsigned long long *word = s_match + wave * RADIX + d;
                    const unsigned prev = L.hist[wave][d];           // the counter before this key's atomic (a wave's LDS operations run in program order)
                    if (valid) atomicOr(word, 1ull << lane);
                    __builtin_amdgcn_wave_barrier();
                    const unsigned long long m = valid ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0ull;
                    __builtin_amdgcn_wave_barrier();
                    if (valid && (m & lt_mask) == 0) __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __builtin_amdgcn_wave_barrier();
                    expect = prev + (unsigned)__popcll(m & lt_mask);
                }
            }
            if (uniform) {                                           // one counter bump instead of 64 same-address atomics
                const unsigned prev = L.hist[wave][d0];
                __builtin_amdgcn_wave_barrier();
                r = prev + (unsigned)__popcll(act & lt_mask);
                if (valid && (act & lt_mask) == 0) L.hist[wave][d0] = prev + (unsigned)__popcll(act);
                __builtin_amdgcn_wave_barrier();
            } else {
                r = valid ? atomicAdd(&L.hist[wave][d], 1u) : 0u;
                if constexpr (VERIFY) if (vflag && valid && r != expect) __builtin_trap();      // a lane was served out of lane order
            }
            rd[k] = valid ? (r | (d << 16)) : ~0u;
            // Tripwire (round 3): two NEIGHBOURING lanes with the same digit must have received consecutive ranks.  Lane order
            // of same-address LDS atomics is what the hardware does, not what the ISA promises; the once-per-device self-test
            // samples it under its own conditions.  Should a part ever serve the lanes differently, a sort must fail loudly
            // (the trap surfaces as an error at the next synchronisation) and not return a silently unstable pass.  Checked on
            // the first key of every lane only (one DPP move and three ALU operations per lane and tile: checking all twelve
            // cost 4 % of the sort): with random digits 1/256 of the neighbour pairs share a digit, ~300 000 checks per pass.
            if (k == 0) {
                const unsigned left = (unsigned)__builtin_amdgcn_update_dpp((int)~0u, (int)rd[k], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                if (valid && left != ~0u && (left >> 16) == d && ((left + 1u) & 0xffffu) != (r & 0xffffu)) __builtin_trap();
            }
        } else {
        // m = the real (non-padding) lanes of this wave holding the same digit.  Every lane ORs
        // its bit into the wave's mask word of its digit in LDS and reads the word back (three LDS
        // operations instead of ~45 vector instructions for eight ballots and per-lane selects);
        // the group's first lane clears the word for the next key.
        unsigned long long m = act;
        if (!uniform) {
            unsigned long long *word = s_match + wave * RADIX + d;
            if (valid) atomicOr(word, 1ull << lane);
            __builtin_amdgcn_wave_barrier();
            m = valid ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0ull;
            __builtin_amdgcn_wave_barrier();
            if (valid && (m & lt_mask) == 0) __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __builtin_amdgcn_wave_barrier();
        } else if (!valid) {
            m = 0ull;
        }
        const unsigned before = __popcll(m & lt_mask);
        const unsigned cnt = __popcll(m);
        const unsigned prev = L.hist[wave][d];
        __builtin_amdgcn_wave_barrier();
        // padding lanes take no rank and no slot: tile positions 0..nvalid-1 are exactly the real keys
        if (valid) {
            if (before == 0) L.hist[wave][d] = prev + cnt;
            rd[k] = (prev + before) | (d << 16);
        } else {
            rd[k] = ~0u;
        }
        __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();

    // lanes 0..255: digit t -> exclusive offsets across the 16 waves, tile count, tile-local start
    unsigned count = 0, inc = 0;
    if (t < RADIX) {
#pragma unroll
        for (int w = 0; w < RW; ++w) { unsigned c = L.hist[w][t]; L.hist[w][t] = count; count += c; }
        inc = count;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            unsigned u = __shfl_up(inc, off, 64);
            if (lane >= off) inc += u;
        }
        if (lane == kWave - 1) L.wtot[wave] = inc;
    }
    __syncthreads();
    if (t < RADIX) {
        unsigned woff = 0;
#pragma unroll
        for (int w = 0; w < RADIX / kWave; ++w) if (w < wave) woff += L.wtot[w];
        unsigned dstart = woff + inc - count;
        L.dstart[t] = dstart;
        L.gbase[t] = table[(size_t)t * nblocks + tile] - dstart;
        // fold the digit's tile-local start into the per-wave offsets: one table read per key in the re-order below
#pragma unroll
        for (int w = 0; w < RW; ++w) L.hist[w][t] += dstart;
    }
    __syncthreads();

    // re-order the tile in LDS
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        if (FULL || rd[k] != ~0u) {
            const unsigned d = rd[k] >> 16;
            const unsigned pos = L.hist[wave][d] + (rd[k] & 0xffffu);
            s_keys[pos] = key[k];
            if constexpr (VB != 0) s_vals[pos] = val[k];
        }
    }
    __syncthreads();

    // Tried (round 2, tools/r02_sort_ab.py): a lane writing 4 consecutive tile positions with ONE 16-byte store when they lie
    // in one run (15 of 16 groups; 3 store instructions per lane instead of 12): 10.93 against 10.20 ms -- the runs start at
    // arbitrary 4-byte offsets and the wide stores straddle cache lines.
    if constexpr (FULL) {
#pragma unroll 4
        for (int k = 0; k < KPT; ++k) {
            const K kk = s_keys[t + k * RB];
            const unsigned d = (unsigned)(to_ordered<K, MODE, DESC>(kk) >> shift) & (RADIX - 1);
            const unsigned g = L.gbase[d] + (unsigned)(t + k * RB);
            keys_out[g] = kk;
            if constexpr (VB != 0) vals_out[g] = s_vals[t + k * RB];
        }
    } else {
        for (int i = t; i < nvalid; i += RB) {
            K kk = s_keys[i];
            unsigned d = (unsigned)(to_ordered<K, MODE, DESC>(kk) >> shift) & (RADIX - 1);
            unsigned g = L.gbase[d] + (unsigned)i;
            keys_out[g] = kk;
            if constexpr (VB != 0) vals_out[g] = s_vals[i];
        }
    }
}

// 8 waves per SIMD = two 1024-lane workgroups per CU (<= 64 VGPRs)
// FULL = true: launched over the complete tiles (first_tile = their number); FULL = false: one
// workgroup for the ragged last tile (first_tile = its index).
constexpr unsigned SORT_VERIFY_EVERY = 16;       // one complete tile in 16 is ranked twice (scatter_tile VERIFY)

template <typename K, int MODE, bool DESC, int VB, int KPT, bool FULL, bool ATOMIC_RANK, bool VERIFY = false>
__global__ __launch_bounds__(RB, 8)
void radix_scatter_kernel(const K *__restrict__ keys_in, K *__restrict__ keys_out,
        const void *__restrict__ vals_in_, void *__restrict__ vals_out_,
        long long n, int shift, unsigned nblocks, unsigned first_tile, const unsigned *__restrict__ table, unsigned verify_every)
{
    typedef typename valtype<VB>::type VT;
    __shared__ scatter_lds<K, VB, KPT> L;
    unsigned tile = first_tile + blockIdx.x;
    if constexpr (FULL) {
        // XCD-aware order: workgroup b runs on XCD b % 8; give every XCD ONE contiguous range of
        // tiles.  Tiles that are neighbours in the input write neighbouring runs of every digit
        // (a run is 24-48 elements: a fraction of a cache line at either end); on the same XCD
        // those partial lines meet in one L2 and leave it as full lines.
        // Tried (round 2): every XCD taking `run` consecutive tiles of each group of 8 * run, so that the eight XCDs sweep
        // the array front to back together -- run 4 / 16 / 32 / 64 / 256: 10.71 / 10.21 / 10.15 / 10.10 / 10.05 ms against
        // 10.00 ms for the contiguous eighths on the same box (1e9 u32 keys).
        const unsigned per = (first_tile + 7) / 8;          // FULL launches pass the number of complete tiles here
        tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
        if (tile >= first_tile) return;
    }
    // every verify_every-th complete tile is ranked both ways (same launch, same tile order: its runs meet their neighbours' in L2)
    const bool vflag = VERIFY && FULL && verify_every && tile % verify_every == verify_every - 1;
    scatter_tile<K, MODE, DESC, VB, KPT, FULL, ATOMIC_RANK, VERIFY>(L, tile, keys_in, keys_out,
            reinterpret_cast<const VT *>(vals_in_), reinterpret_cast<VT *>(vals_out_), n, shift, nblocks, table, vflag);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: the lean scatter.  Round 4's kernel issued 39 VALU and 55 SALU instructions per key round (the uniform-digit
// branches, the verified-tile branches, 64-bit address arithmetic); profiles/r04_bench_kernel_stats.csv: 1.79 ms per pass at
// 1e9 keys = 0.56 of HBM, bound by instruction issue.  Here:
//   * the tile's digit counts come from the scanned table (entry [d][tile + 1] - entry [d][tile]) BEFORE the keys arrive, so
//     the tile-local digit starts are computed beside the key loads, and a tile whose keys all share the digit (constant upper
//     digits of a small key range) is copied straight from registers -- no per-round uniformity test;
//   * ranks with a COMPLETE check of the property they rest on, on every key of every tile (CHECK 1 / 2): each lane ORs its lane
//     bit into a 64-bit word of its (wave, digit) and gets the bits of the lanes that were served before it; a lane that sees
//     the bit of a HIGHER lane was served out of lane order -> trap.  If no lane of a group sees a higher bit, the group was
//     served in lane order (of two lanes a < b one is served first; b first would show b's bit to a).
//       CHECK 2 (default): the rank itself is popcount(returned bits) + the counter read before the round: nothing depends on
//                the order in which any OTHER instruction serves its lanes (the counter is bumped by non-returning adds);
//       CHECK 1: the rank is the return of the counter atomic (round 4), the OR word checks the same wave round beside it;
//       CHECK 0: the counter atomic alone (A/B only).
//   * no branches in the rank loop (violations are OR-ed into two registers and tested once), byte offsets with a scalar base
//     per key round in the write-out (one add per key).
template <typename T, int AUX = 0>
__device__ __forceinline__ void store_elem(T v, __amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned scalar_bytes) {
    if constexpr (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)lane_bytes, (int)scalar_bytes, AUX);
    else {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, (int)lane_bytes, (int)scalar_bytes, AUX);
    }
}

template <typename K, int VB, int KPT>
struct lean_lds {
    static constexpr int TILE = RB * KPT;
    static constexpr int TILE_BYTES = TILE * ((int)sizeof(K) + VB);
    static constexpr int CHK_BYTES = RW * RADIX * 8;
    static constexpr int RAW_WORDS = ((TILE_BYTES > CHK_BYTES ? TILE_BYTES : CHK_BYTES) + 15) / 16 * 2;
    unsigned long long raw[RAW_WORDS];      // the order words while the keys are ranked, then the re-ordered tile
    unsigned hist[RW][RADIX];               // per (wave, digit): count, then tile position of the wave's first key of that digit
    unsigned gbase[RADIX];                  // global index of the tile's first key of a digit, minus its tile position
    unsigned dstart[RADIX];
    unsigned cnt[RADIX];
    unsigned wtot[RADIX / kWave];
    int uni;
};

template <typename K, int MODE, bool DESC, int VB, int KPT, int CHECK, bool WIDE>
__global__ __launch_bounds__(RB, 8)
void radix_scatter_lean_kernel(const K *__restrict__ keys_in, K *__restrict__ keys_out,
        const void *__restrict__ vals_in_, void *__restrict__ vals_out_,
        long long n, int shift, unsigned nblocks, unsigned nfull, const unsigned *__restrict__ table)
{
    typedef typename valtype<VB>::type VT;
    constexpr int TILE = RB * KPT;
    __shared__ __attribute__((aligned(16))) lean_lds<K, VB, KPT> L;
    const VT *__restrict__ vals_in = reinterpret_cast<const VT *>(vals_in_);
    VT *__restrict__ vals_out = reinterpret_cast<VT *>(vals_out_);
    K *s_keys = reinterpret_cast<K *>(L.raw);
    VT *s_vals = reinterpret_cast<VT *>(reinterpret_cast<char *>(L.raw) + (size_t)TILE * sizeof(K));
    unsigned long long *s_chk = L.raw;

    // XCD-contiguous tile order, as the histogram kernel (see radix_scatter_kernel)
    const unsigned per = (nfull + 7) / 8;
    const unsigned tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (tile >= nfull) return;

    const int t = threadIdx.x, wave = t / kWave, lane = t % kWave;
    const long long wbase = (long long)tile * TILE + (long long)wave * (kWave * KPT);

    K key[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) key[k] = __builtin_nontemporal_load(keys_in + wbase + k * kWave + lane);
    VT val[VB ? KPT : 1];
    if constexpr (VB != 0) {
#pragma unroll
        for (int k = 0; k < KPT; ++k) val[k] = __builtin_nontemporal_load(vals_in + wbase + k * kWave + lane);
    }
    // the tile's count and global base of digit t, straight from the scanned table (independent of the keys: in flight with them)
    unsigned b0 = 0, cnt = 0;
    if (t < RADIX) {
        const size_t idx = (size_t)t * nblocks + tile;
        b0 = table[idx];
        const unsigned b1 = (idx + 1 < (size_t)RADIX * nblocks) ? table[idx + 1] : (unsigned)n;
        cnt = b1 - b0;
    }
    {   // zero the counters and the order words: 4 + 8 words per lane
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 z = {0u, 0u, 0u, 0u};
        reinterpret_cast<u4 *>(&L.hist[0][0])[t] = z;
        if constexpr (CHECK != 0) {
            reinterpret_cast<u4 *>(s_chk)[t] = z;
            reinterpret_cast<u4 *>(s_chk)[t + RB] = z;
        }
        if (t == 0) L.uni = 0;
    }
    unsigned inc = 0;
    if (t < RADIX) {
        inc = cnt;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const unsigned u = __shfl_up(inc, off, 64);
            if (lane >= off) inc += u;
        }
        if (lane == kWave - 1) L.wtot[wave] = inc;
    }
    __syncthreads();
    if (t < RADIX) {
        unsigned woff = 0;
#pragma unroll
        for (int w = 0; w < RADIX / kWave; ++w) if (w < wave) woff += L.wtot[w];
        const unsigned ds = woff + inc - cnt;
        L.dstart[t] = ds;
        L.cnt[t] = cnt;
        L.gbase[t] = b0 - ds + (unsigned)TILE;               // biased by TILE: never negative, so that base + position stays a plain 32-bit sum
        if (cnt == (unsigned)TILE) L.uni = t + 1;
    }
    __syncthreads();
    if (L.uni) {
        // every key of the tile has digit uni - 1: the tile keeps its order (a stable pass moves it as a block)
        const unsigned g = L.gbase[L.uni - 1] - (unsigned)TILE + (unsigned)(wave * (kWave * KPT) + lane);
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            keys_out[(size_t)g + k * kWave] = key[k];
            if constexpr (VB != 0) vals_out[(size_t)g + k * kWave] = val[k];
        }
        return;
    }

    unsigned rr[KPT];
    {
        unsigned *hw = L.hist[wave];
        unsigned long long *cw = s_chk + wave * RADIX;
        const unsigned long long lanebit = 1ull << lane;
        const unsigned long long ge = ~(lanebit - 1ull);        // this lane and the higher ones (its own bit is never in a returned word)
        unsigned long long bad = 0ull;
        // rounds in groups of GR: the LDS operations of a group are issued back to back, then their returns are folded into
        // ranks (all KPT returned pairs held back until the end cost 3-5 spilled registers at 64 per lane)
        constexpr int GR = CHECK == 2 ? (KPT % 4 == 0 ? 4 : (KPT % 3 == 0 ? 3 : 1)) : KPT;
#pragma unroll
        for (int k0 = 0; k0 < KPT; k0 += GR) {
            unsigned prev[GR];
            unsigned long long old[GR];
#pragma unroll
            for (int j = 0; j < GR; ++j) {
                const int k = k0 + j;
                const unsigned d = (unsigned)(to_ordered<K, MODE, DESC>(key[k]) >> shift) & (RADIX - 1);
                if constexpr (CHECK == 0) {
                    rr[k] = atomicAdd(&hw[d], 1u);
                } else if constexpr (CHECK == 1) {
                    rr[k] = atomicAdd(&hw[d], 1u);
                    old[j] = atomicOr(&cw[d], lanebit);
                    atomicXor(&cw[d], lanebit);                 // back to zero for the next round (no return: any order)
                } else {
                    prev[j] = hw[d];                             // a wave's LDS operations execute in program order: before this round's adds
                    __builtin_amdgcn_wave_barrier();
                    old[j] = atomicOr(&cw[d], lanebit);
                    atomicXor(&cw[d], lanebit);
                    atomicAdd(&hw[d], 1u);                      // no return: only the sum matters
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if constexpr (CHECK == 2) __builtin_amdgcn_sched_barrier(0);
            if constexpr (CHECK != 0) {
#pragma unroll
                for (int j = 0; j < GR; ++j) {
                    bad |= old[j] & ge;
                    if constexpr (CHECK == 2) {
                        rr[k0 + j] = prev[j] + (unsigned)__popcll(old[j]);
                        asm volatile("" : "+v"(rr[k0 + j]));     // computed HERE: the optimiser otherwise sinks the sum to its use and keeps three registers alive for one
                    }
                }
            }
            if constexpr (CHECK == 2) __builtin_amdgcn_sched_barrier(0);
        }
        if (bad) __builtin_trap();                               // a lane of one LDS atomic was served before a lower lane with the same address
    }
    __syncthreads();

    if (t < RADIX) {
        unsigned run = L.dstart[t];
#pragma unroll
        for (int w = 0; w < RW; ++w) { const unsigned c = L.hist[w][t]; L.hist[w][t] = run; run += c; }
        if (run - L.dstart[t] != L.cnt[t]) __builtin_trap();     // the table and the keys disagree (the input changed between the passes' kernels)
    }
    __syncthreads();

    // all offsets first, then all stores: a store into the tile may alias the offset table as far as the compiler knows, and
    // read -> wait -> write per key serialises twelve LDS round trips
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const unsigned d = (unsigned)(to_ordered<K, MODE, DESC>(key[k]) >> shift) & (RADIX - 1);      // recomputed: one operation against a register per key
        rr[k] += L.hist[wave][d];
    }
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        s_keys[rr[k]] = key[k];
        if constexpr (VB != 0) s_vals[rr[k]] = val[k];
    }
    __syncthreads();

    // descriptors that start TILE elements in front of the outputs: with the biased bases every lane offset is a plain unsigned number
    __amdgpu_buffer_rsrc_t rk, rv;
    if constexpr (!WIDE) {
        rk = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(keys_out) - (long long)TILE * (long long)sizeof(K), 0, -1, 0x00020000);
        if constexpr (VB != 0) rv = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(vals_out) - (long long)TILE * (long long)VB, 0, -1, 0x00020000);
    }
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const K kk = s_keys[t + k * RB];
        const unsigned d = (unsigned)(to_ordered<K, MODE, DESC>(kk) >> shift) & (RADIX - 1);
        const unsigned e = L.gbase[d] + (unsigned)t;            // + k * RB - TILE: the scalar offset and the descriptor base
        if constexpr (WIDE) {
            const unsigned g = e + (unsigned)(k * RB) - (unsigned)TILE;
            keys_out[(size_t)g] = kk;
            if constexpr (VB != 0) vals_out[(size_t)g] = s_vals[t + k * RB];
        } else {
            // (n + 2 TILE) x element bytes fits 32 bits (checked by the host).  Buffer stores: descriptor base + 32-bit lane offset
            // + scalar offset of the key round -- one add and one shift per key, no 64-bit address per lane
            store_elem<K>(kk, rk, e * (unsigned)sizeof(K), (unsigned)(k * RB) * (unsigned)sizeof(K));
            if constexpr (VB != 0) store_elem<VT>(s_vals[t + k * RB], rv, e * (unsigned)VB, (unsigned)(k * RB) * (unsigned)VB);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5, the default: ranks from ONE returning LDS atomic per key whose own return proves the order it was served in.
// A ranking UNIT is a half wave (32 lanes); its counter of digit d is a 64-bit word [ keys so far : 32 | lane bits : 32 ].  A lane
// adds (1 << 32) | (1 << (lane & 31)): the word it gets back holds, in its upper half, the number of the unit's keys with this digit
// that were counted before it -- earlier rounds AND the lanes of this round that were served first -- and in its lower half the
// bits of exactly those lanes of this round.  A lane that finds the bit of a HIGHER lane there was served out of lane order:
// trap.  If no lane does, the upper half IS the stable rank (of two lanes a < b with one digit, b first would show b's bit to a).
// A second, non-returning atomic takes the lane's bit out again.  Two LDS operations per key -- round 4's unchecked rank took
// one, the forms above three (CHECK 1) and four (CHECK 2) -- and nothing is assumed about any instruction but the one that
// delivers the rank.  768 lanes x 16 keys (4-byte keys): 24 units x 256 words = 48 KiB, which the re-ordered tile then reuses;
// the units' tile offsets live on as 16-bit numbers (12 KiB): two workgroups per CU.
constexpr int UB = 768, UW = UB / kWave, UU = 2 * UW;
template <typename K, int VB> constexpr int unit_kpt() { return keys_per_lane((int)sizeof(K), VB) * RB / UB; }
template <typename K, int VB> constexpr bool unit_ok() { return keys_per_lane((int)sizeof(K), VB) * RB % UB == 0; }

template <typename K, int VB, int KPT>
struct unit_lds {
    static constexpr int TILE = UB * KPT;
    static constexpr int TILE_BYTES = TILE * ((int)sizeof(K) + VB);
    static constexpr int WORD_BYTES = UU * RADIX * 8;
    static constexpr int RAW_WORDS = ((TILE_BYTES > WORD_BYTES ? TILE_BYTES : WORD_BYTES) + 15) / 16 * 2;
    unsigned long long raw[RAW_WORDS];      // the units' counter words while the keys are ranked, then the re-ordered tile
    unsigned short off[UU][RADIX];          // tile position of a unit's first key of a digit
    unsigned gbase[RADIX];
    unsigned dstart[RADIX];
    unsigned wtot[RADIX / kWave];
    int uni;
};

template <typename K, int MODE, bool DESC, int VB, int KPT, bool WIDE, int AUX = 0>
__global__ __launch_bounds__(UB, 6)
void radix_scatter_unit_kernel(const K *__restrict__ keys_in, K *__restrict__ keys_out,
        const void *__restrict__ vals_in_, void *__restrict__ vals_out_,
        long long n, int shift, unsigned nblocks, unsigned nfull, const unsigned *__restrict__ table)
{
    typedef typename valtype<VB>::type VT;
    constexpr int TILE = UB * KPT;
    static_assert(TILE == RB * keys_per_lane((int)sizeof(K), VB), "the tile of the histogram kernel");
    __shared__ __attribute__((aligned(16))) unit_lds<K, VB, KPT> L;
    const VT *__restrict__ vals_in = reinterpret_cast<const VT *>(vals_in_);
    VT *__restrict__ vals_out = reinterpret_cast<VT *>(vals_out_);
    K *s_keys = reinterpret_cast<K *>(L.raw);
    VT *s_vals = reinterpret_cast<VT *>(reinterpret_cast<char *>(L.raw) + (size_t)TILE * sizeof(K));
    unsigned long long *s_word = L.raw;

    const unsigned per = (nfull + 7) / 8;                       // XCD-contiguous tile order (radix_scatter_kernel)
    const unsigned tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (tile >= nfull) return;

    const int t = threadIdx.x, wave = t / kWave, lane = t % kWave;
    // a UNIT's keys are consecutive in the tile (32 * KPT of them, round k = the next 32): the units' keys then follow each other
    // in tile order, which is what a stable pass ranks by.  (A wave's load covers two 128-byte pieces 32 * KPT elements apart.)
    const int unit = 2 * wave + (lane >> 5);
    const int upos = unit * (32 * KPT) + (lane & 31);           // tile position of the lane's first key
    const long long ubase = (long long)tile * TILE + upos;

    K key[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) key[k] = __builtin_nontemporal_load(keys_in + ubase + k * 32);
    VT val[VB ? KPT : 1];
    if constexpr (VB != 0) {
#pragma unroll
        for (int k = 0; k < KPT; ++k) val[k] = __builtin_nontemporal_load(vals_in + ubase + k * 32);
    }
    unsigned b0 = 0, cnt = 0;                                   // the tile's count and global base of digit t, from the scanned table
    if (t < RADIX) {
        const size_t idx = (size_t)t * nblocks + tile;
        b0 = table[idx];
        const unsigned b1 = (idx + 1 < (size_t)RADIX * nblocks) ? table[idx + 1] : (unsigned)n;
        cnt = b1 - b0;
    }
    {   // zero the counter words: UU * 256 * 8 bytes = 4 x 16 bytes per lane
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < UU * RADIX * 8 / 16 / UB; ++q) reinterpret_cast<u4 *>(s_word)[t + q * UB] = z;
        static_assert(UU * RADIX * 8 / 16 % UB == 0, "whole rounds");
        if (t == 0) L.uni = 0;
    }
    unsigned inc = 0;
    if (t < RADIX) {
        inc = cnt;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const unsigned u = __shfl_up(inc, o, 64);
            if (lane >= o) inc += u;
        }
        if (lane == kWave - 1) L.wtot[wave] = inc;
    }
    __syncthreads();
    if (t < RADIX) {
        unsigned woff = 0;
#pragma unroll
        for (int w = 0; w < RADIX / kWave; ++w) if (w < wave) woff += L.wtot[w];
        const unsigned ds = woff + inc - cnt;
        L.dstart[t] = ds;
        L.gbase[t] = b0 - ds + (unsigned)TILE;               // biased by TILE (see the lean kernel)
        if (cnt == (unsigned)TILE) L.uni = t + 1;
    }
    __syncthreads();
    if (L.uni) {                                              // every key of the tile has one digit: moved as a block
        const unsigned g = L.gbase[L.uni - 1] - (unsigned)TILE + (unsigned)upos;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            keys_out[(size_t)g + k * 32] = key[k];
            if constexpr (VB != 0) vals_out[(size_t)g + k * 32] = val[k];
        }
        return;
    }

    unsigned rr[KPT];
    {
        unsigned long long *uw = s_word + unit * RADIX;
        const unsigned mybit = 1u << (lane & 31);
        const unsigned long long add = (1ull << 32) | mybit;
        unsigned seen = 0;                                     // lane bits returned over all rounds
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const unsigned d = (unsigned)(to_ordered<K, MODE, DESC>(key[k]) >> shift) & (RADIX - 1);
            const unsigned long long old = atomicAdd(&uw[d], add);
            atomicAdd(&uw[d], (unsigned long long)(0ull - (unsigned long long)mybit));       // the lane's bit out again (no return: any order)
            rr[k] = (unsigned)(old >> 32);
            seen |= (unsigned)old;
        }
        if (seen >> (lane & 31)) __builtin_trap();             // a lane was served before a lower lane of its unit that hit the same word
    }
    __syncthreads();

    if (t < RADIX) {
        unsigned run = L.dstart[t];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            const unsigned c = (unsigned)(s_word[u * RADIX + t] >> 32);
            L.off[u][t] = (unsigned short)run;
            run += c;
        }
        if (run - L.dstart[t] != cnt) __builtin_trap();       // the table and the keys disagree
    }
    __syncthreads();

#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const unsigned d = (unsigned)(to_ordered<K, MODE, DESC>(key[k]) >> shift) & (RADIX - 1);
        rr[k] += L.off[unit][d];
    }
    __syncthreads();                                          // (the tile goes where the counter words are)
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        s_keys[rr[k]] = key[k];
        if constexpr (VB != 0) s_vals[rr[k]] = val[k];
    }
    __syncthreads();

    __amdgpu_buffer_rsrc_t rk, rv;
    if constexpr (!WIDE) {
        rk = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(keys_out) - (long long)TILE * (long long)sizeof(K), 0, -1, 0x00020000);
        if constexpr (VB != 0) rv = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(vals_out) - (long long)TILE * (long long)VB, 0, -1, 0x00020000);
    }
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const K kk = s_keys[t + k * UB];
        const unsigned d = (unsigned)(to_ordered<K, MODE, DESC>(kk) >> shift) & (RADIX - 1);
        const unsigned e = L.gbase[d] + (unsigned)t;
        if constexpr (WIDE) {
            const unsigned g = e + (unsigned)(k * UB) - (unsigned)TILE;
            keys_out[(size_t)g] = kk;
            if constexpr (VB != 0) vals_out[(size_t)g] = s_vals[t + k * UB];
        } else {
            store_elem<K, AUX>(kk, rk, e * (unsigned)sizeof(K), (unsigned)(k * UB) * (unsigned)sizeof(K));
            if constexpr (VB != 0) store_elem<VT, AUX>(s_vals[t + k * UB], rv, e * (unsigned)VB, (unsigned)(k * UB) * (unsigned)VB);
        }
    }
}

extern int g_sort_rank;

template <typename K, int VB> constexpr int kpt_for() { return keys_per_lane((int)sizeof(K), VB); }

template <typename K, int MODE, bool DESC, int VB, int CHECK>
void launch_lean(hipStream_t s, bool wide, unsigned nfull, const K *src, K *dst, const void *vsrc, void *vdst,
        int64_t n, int shift, unsigned nblocks, const unsigned *table) {
    constexpr int KPT = kpt_for<K, VB>();
    const unsigned grid = (nfull + 7) / 8 * 8;
    if (wide) radix_scatter_lean_kernel<K, MODE, DESC, VB, KPT, CHECK, true><<<grid, RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table);
    else      radix_scatter_lean_kernel<K, MODE, DESC, VB, KPT, CHECK, false><<<grid, RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table);
}

template <typename K, int MODE, bool DESC, int VB>
void launch_unit(hipStream_t s, bool wide, unsigned nfull, const K *src, K *dst, const void *vsrc, void *vdst,
        int64_t n, int shift, unsigned nblocks, const unsigned *table) {
    if constexpr (unit_ok<K, VB>()) {
        constexpr int KPT = unit_kpt<K, VB>();
        const unsigned grid = (nfull + 7) / 8 * 8;
        static const int aux = std::getenv("VEXHIP_SORT_STORE_AUX") ? std::atoi(std::getenv("VEXHIP_SORT_STORE_AUX")) : 0;     // (A/B, round 6: 2 = nt, 18 = nt sc1, 17 = sc0 sc1)
        if (wide) radix_scatter_unit_kernel<K, MODE, DESC, VB, KPT, true><<<grid, UB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table);
        else if (aux == 2)  radix_scatter_unit_kernel<K, MODE, DESC, VB, KPT, false, 2><<<grid, UB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table);
        else if (aux == 18) radix_scatter_unit_kernel<K, MODE, DESC, VB, KPT, false, 18><<<grid, UB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table);
        else if (aux == 17) radix_scatter_unit_kernel<K, MODE, DESC, VB, KPT, false, 17><<<grid, UB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table);
        else      radix_scatter_unit_kernel<K, MODE, DESC, VB, KPT, false><<<grid, UB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table);
    } else launch_lean<K, MODE, DESC, VB, 2>(s, wide, nfull, src, dst, vsrc, vdst, n, shift, nblocks, table);      // (4-byte keys with 8-byte values: 4096 pairs per tile do not split over 768 lanes)
}

// rank: 0 match words (round 2), 1 counter atomics with one verified tile in 16 (round 4), 2 the same without verified tiles,
//       3 / 4 / 5 the lean scatter with CHECK 0 / 1 / 2, 6 the unit scatter (one returning atomic per key that proves its own order: the default)
template <typename K, int MODE, bool DESC, int VB>
int sort_passes(hipStream_t s, K *keys, K *keys_tmp, void *vals, void *vals_tmp, int64_t n, unsigned *tmp, int rank) {
    const bool atomic_rank = rank == 1 || rank == 2;
    const unsigned every = rank == 1 ? SORT_VERIFY_EVERY : 0u;
    constexpr int KPT = kpt_for<K, VB>();
    constexpr int TILE = RB * KPT;
    const unsigned nblocks = (unsigned)((n + TILE - 1) / TILE);
    const int64_t tn = (int64_t)nblocks * RADIX;
    unsigned *table = tmp;
    unsigned *scan_tmp = tmp + (tn + 3) / 4 * 4;
    K *src = keys, *dst = keys_tmp;
    void *vsrc = vals, *vdst = vals_tmp;
    constexpr int passes = (int)sizeof(K);
    const int vec_ok = ((reinterpret_cast<uintptr_t>(keys) & 15) == 0) && ((reinterpret_cast<uintptr_t>(keys_tmp) & 15) == 0);
    constexpr int64_t widest = (int64_t)sizeof(K) > VB ? (int64_t)sizeof(K) : VB;
    const bool wide = (n + 2 * TILE) * widest >= (1ll << 32) - 16;  // byte offsets of the write-out (lane + scalar part, range-checked together) beyond 32 bits
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        static const int hun = std::getenv("VEXHIP_SORT_HIST_UNROLL") ? std::atoi(std::getenv("VEXHIP_SORT_HIST_UNROLL")) : 6;     // (A/B, round 6)
        if (hun == 12) radix_hist_kernel<K, MODE, DESC, KPT, 12><<<(nblocks + 7) / 8 * 8, HB, 0, s>>>(src, n, shift, nblocks, table, vec_ok);
        else if (hun == 3) radix_hist_kernel<K, MODE, DESC, KPT, 3><<<(nblocks + 7) / 8 * 8, HB, 0, s>>>(src, n, shift, nblocks, table, vec_ok);
        else radix_hist_kernel<K, MODE, DESC, KPT><<<(nblocks + 7) / 8 * 8, HB, 0, s>>>(src, n, shift, nblocks, table, vec_ok);
        VEXHIP_LAUNCH_CHECK();
        if (int rc = scan_exclusive_u32_tmp(s, table, table, tn, scan_tmp)) return rc;
        const unsigned nfull = (unsigned)(n / TILE);
        if (nfull) {
            if (rank == 3) launch_lean<K, MODE, DESC, VB, 0>(s, wide, nfull, src, dst, vsrc, vdst, n, shift, nblocks, table);
            else if (rank == 4) launch_lean<K, MODE, DESC, VB, 1>(s, wide, nfull, src, dst, vsrc, vdst, n, shift, nblocks, table);
            else if (rank == 5) launch_lean<K, MODE, DESC, VB, 2>(s, wide, nfull, src, dst, vsrc, vdst, n, shift, nblocks, table);
            else if (rank == 6) launch_unit<K, MODE, DESC, VB>(s, wide, nfull, src, dst, vsrc, vdst, n, shift, nblocks, table);
            else if (atomic_rank) {
                radix_scatter_kernel<K, MODE, DESC, VB, KPT, true, true, true><<<(nfull + 7) / 8 * 8, RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table, every);
            } else radix_scatter_kernel<K, MODE, DESC, VB, KPT, true, false><<<(nfull + 7) / 8 * 8, RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table, 0u);
            VEXHIP_LAUNCH_CHECK();
        }
        if (nfull < nblocks) {
            // the ragged last tile: one workgroup; ranked by the match words (order-independent) unless the round 4 forms are asked for
            if (atomic_rank) radix_scatter_kernel<K, MODE, DESC, VB, KPT, false, true><<<1, RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table, 0u);
            else radix_scatter_kernel<K, MODE, DESC, VB, KPT, false, false><<<1, RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table, 0u);
            VEXHIP_LAUNCH_CHECK();
        }
        std::swap(src, dst);
        std::swap(vsrc, vdst);
    }
    // sizeof(K) passes is even => the result is back in `keys` / `vals`
    static_assert(sizeof(K) % 2 == 0, "ping-pong parity");
    return 0;
}

template <typename K, int MODE>
int sort_dispatch(hipStream_t s, int desc, int vb, void *keys, void *keys_tmp, void *vals, void *vals_tmp, int64_t n, void *tmp, int rank) {
#define GO(DESC, VB) return sort_passes<K, MODE, DESC, VB>(s, (K *)keys, (K *)keys_tmp, vals, vals_tmp, n, (unsigned *)tmp, rank)
    if (desc) { if (vb == 0) GO(true, 0); if (vb == 4) GO(true, 4); if (vb == 8) GO(true, 8); }
    else      { if (vb == 0) GO(false, 0); if (vb == 4) GO(false, 4); if (vb == 8) GO(false, 8); }
#undef GO
    return fail(__FILE__, __LINE__, "value_bytes must be 0, 4 or 8");
}

// ---- does this device service same-address lanes of one LDS atomic in lane order?  (checked once per device) ----
__global__ __launch_bounds__(1024)
void lds_atomic_order_kernel(unsigned seed, int rounds, unsigned *violations) {
    __shared__ unsigned cnt[RW][RADIX];
    const int t = threadIdx.x, wave = t / kWave, lane = t % kWave;
    unsigned bad = 0;
    unsigned h = seed ^ (blockIdx.x * 0x9E3779B9u) ^ (t * 0x85EBCA6Bu);
    for (int r = 0; r < rounds; ++r) {
        for (int i = t; i < RW * RADIX; i += RB) (&cnt[0][0])[i] = 0;
        __syncthreads();
        h = h * 1664525u + 1013904223u;
        const unsigned nd = 1u << (r % 9);                        // 1 .. 256 distinct digits
        const unsigned d = (h >> 16) & (nd - 1);
        const unsigned old = atomicAdd(&cnt[wave][d], 1u);
        unsigned expect = 0;
        for (int l = 0; l < kWave; ++l) { const unsigned dl = __shfl(d, l, 64); if (l < lane && dl == d) ++expect; }
        if (old != expect) ++bad;
        __syncthreads();
    }
    if (bad) atomicAdd(violations, bad);
}

int g_sort_rank = -1;               // -1: the default (6: the unit scatter); 0: match words; 1 / 2: round 4's counter-atomic ranks with / without verified tiles (1 only if the
                                    // device self-test agrees); 3 / 4 / 5: the lean scatter, ranks unchecked / checked beside / taken from the checked words

int atomic_rank_ok(int dev, hipStream_t s, bool *ok) {
    static std::atomic<int> verdict[64];          // 0 unknown, 1 in order, 2 not (two threads may both run the test: same answer)
    if (dev < 0 || dev >= 64) { *ok = false; return 0; }
    if (!verdict[dev].load()) {
        unsigned *d = nullptr, h = 1;
        VEXHIP_TRY(hipMalloc(&d, sizeof(unsigned)));
        hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned), s);
        if (e == hipSuccess) { lds_atomic_order_kernel<<<512, RB, 0, s>>>(12345u, 72, d); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipFree(d);
        VEXHIP_TRY(e);
        verdict[dev].store(h == 0 ? 1 : 2);
    }
    *ok = verdict[dev].load() == 1;
    return 0;
}

} // namespace
} // namespace vexhip

using namespace vexhip;

extern "C" {

int vexhip_sort_set_rank(int mode) { g_sort_rank = mode; return 0; }

size_t vexhip_sort_tmp_bytes(int key_dtype, int64_t n) {
    (void)key_dtype;
    const int64_t smallest_tile = RB * 3;                 // (8-byte key, 8-byte value)
    int64_t nblocks = (n + smallest_tile - 1) / smallest_tile;
    int64_t tn = nblocks * RADIX;
    return sizeof(unsigned) * (size_t)((tn + 3) / 4 * 4 + (int64_t)scan_tmp_elems_u32(tn) + 4);
}

int vexhip_sort(int dev, void *stream, int key_dtype, int descending,
        void *keys, void *keys_tmp, int value_bytes, void *vals, void *vals_tmp, int64_t n, void *tmp)
{
    VEXHIP_REQUIRE(n >= 0, "negative size");
    if (n <= 1) return 0;
    VEXHIP_REQUIRE(n < (1ll << 31), "at most 2^31-1 keys per call (sort.hpp:1738 has the same limit)");
    VEXHIP_REQUIRE(keys && keys_tmp && tmp, "NULL argument");
    VEXHIP_REQUIRE(value_bytes == 0 || (vals && vals_tmp), "NULL value buffers");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    int ar = g_sort_rank < 0 ? 6 : g_sort_rank;
    if (ar == 1) { bool ok = false; if (int rc = atomic_rank_ok(dev, s, &ok)) return rc; if (!ok) ar = 0; }
    switch (key_dtype) {
        case VEXHIP_U32: return sort_dispatch<unsigned, KEY_UNSIGNED>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, ar);
        case VEXHIP_I32: return sort_dispatch<unsigned, KEY_SIGNED>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, ar);
        case VEXHIP_F32: return sort_dispatch<unsigned, KEY_FLOAT>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, ar);
        case VEXHIP_U64: return sort_dispatch<unsigned long long, KEY_UNSIGNED>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, ar);
        case VEXHIP_I64: return sort_dispatch<unsigned long long, KEY_SIGNED>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, ar);
        case VEXHIP_F64: return sort_dispatch<unsigned long long, KEY_FLOAT>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, ar);
    }
    return fail(__FILE__, __LINE__, "unknown key dtype");
}

} // extern "C"

VEXHIP_WARM_TU(sort)
