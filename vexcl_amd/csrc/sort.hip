// vex::sort / vex::sort_by_key on gfx950 (vexcl/sort.hpp:2158-2182).
// The reference is a merge sort (sort.hpp:820-1696) and its contract -- pinned
// by tests/sort.cpp:22-45 -- is std::stable_sort.  Here: stable LSD radix sort,
// 8-bit digits.  Per pass:
//   (1) digit histogram per tile (12 288 u32 keys) -> table[digit][tile]
//   (2) exclusive scan of the table (scan.hip)   -> global base per (digit,tile)
//   (3) scatter: half-wave ranking units rank their keys stably with one returning
//       64-bit LDS atomic per key (radix_scatter_unit_kernel), the tile is re-ordered in
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
// The tile of a sort: 768 lanes (24 half-wave ranking units) x as many keys per lane as fit ~48 KiB of (key + value) bytes --
// 12 288 / 6144 / 3840 / 6144 / 3840 / 3072 elements for (4,0) (4,4) (4,8) (8,0) (8,4) (8,8)-byte (key, value) pairs.
constexpr int UB = 768;
constexpr int unit_keys_per_lane(int key_bytes, int value_bytes) { return 48 * 1024 / (key_bytes + value_bytes) / UB; }
template <typename K, int VB> constexpr int tile_keys() { return UB * unit_keys_per_lane((int)sizeof(K), VB); }
// keys per lane of the 1024-lane match-word kernel that holds such a tile (slots beyond the tile stay empty)
template <typename K, int VB> constexpr int slots_per_lane() { return (tile_keys<K, VB>() + 1023) / 1024; }

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

template <typename K, int MODE, bool DESC, int TILE, int UNROLL = 6>
__global__ __launch_bounds__(HB)
void radix_hist_kernel(const K *__restrict__ keys, long long n, int shift, unsigned nblocks, unsigned *__restrict__ table, int vec_ok)
{
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

// The first generation of the scatter (round 2), kept for two jobs: the ragged LAST tile of every sort, and nothing else -- its ranks
// come from match words and do not depend on the order in which the LDS serves the lanes of one atomic.
// FULL: every slot of the tile holds a key: no validity masks.
template <typename K, int MODE, bool DESC, int VB, int KPT, int TILE_KEYS, bool FULL>
__device__ __forceinline__ void scatter_tile(scatter_lds<K, VB, KPT> &L, const unsigned tile,
        const K *__restrict__ keys_in, K *__restrict__ keys_out,
        const typename valtype<VB>::type *__restrict__ vals_in, typename valtype<VB>::type *__restrict__ vals_out,
        long long n, int shift, unsigned nblocks, const unsigned *__restrict__ table)
{
    typedef typename valtype<VB>::type VT;
    constexpr int TILE = RB * KPT;
    K *s_keys = reinterpret_cast<K *>(L.raw);
    VT *s_vals = reinterpret_cast<VT *>(reinterpret_cast<char *>(L.raw) + (size_t)TILE * sizeof(K));
    unsigned long long *s_match = L.raw;

    static_assert(TILE_KEYS <= TILE && (!FULL || TILE_KEYS == TILE), "the tile must fit the kernel's slots (FULL: fill them)");
    const int t = threadIdx.x, wave = t / kWave, lane = t % kWave;
    const long long base = (long long)tile * TILE_KEYS;
    const int wfirst = wave * (kWave * KPT);                         // tile position of the wave's first slot
    const long long wbase = base + wfirst;
    const int nvalid = FULL ? TILE : (int)((n - base < TILE_KEYS) ? (n - base) : TILE_KEYS);

    for (int i = t; i < RW * RADIX; i += RB) { (&L.hist[0][0])[i] = 0; s_match[i] = 0ull; }

    K key[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const long long i = wbase + k * kWave + lane;
        key[k] = (FULL || wfirst + k * kWave + lane < nvalid) ? __builtin_nontemporal_load(keys_in + i) : K(0);      // read once per pass
    }
    // payloads are fetched with the keys: their latency hides behind the ranking
    VT val[VB ? KPT : 1];
    if constexpr (VB != 0) {
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const long long i = wbase + k * kWave + lane;
            val[k] = (FULL || wfirst + k * kWave + lane < nvalid) ? __builtin_nontemporal_load(vals_in + i) : VT(0);
        }
    }
    __syncthreads();

    unsigned rd[KPT];          // rank within (wave, digit) | digit << 16; ~0u = padding slot
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const bool valid = FULL || (wfirst + k * kWave + lane < nvalid);
        const unsigned d = (unsigned)(to_ordered<K, MODE, DESC>(key[k]) >> shift) & (RADIX - 1);
        const unsigned long long act = FULL ? ~0ull : __ballot(valid);
        const unsigned d0 = __builtin_amdgcn_readfirstlane(d);
        const bool uniform = __ballot(valid && d == d0) == act;      // small key ranges: constant upper digits
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

// FULL = true: launched over the complete tiles (first_tile = their number; rank mode 0: the whole sort by match words, A/B and
// tests); FULL = false: one workgroup for the ragged last tile (first_tile = its index).
template <typename K, int MODE, bool DESC, int VB, int KPT, int TILE_KEYS, bool FULL>
__global__ __launch_bounds__(RB, 8)
void radix_scatter_kernel(const K *__restrict__ keys_in, K *__restrict__ keys_out,
        const void *__restrict__ vals_in_, void *__restrict__ vals_out_,
        long long n, int shift, unsigned nblocks, unsigned first_tile, const unsigned *__restrict__ table)
{
    typedef typename valtype<VB>::type VT;
    __shared__ scatter_lds<K, VB, KPT> L;
    unsigned tile = first_tile + blockIdx.x;
    if constexpr (FULL) {
        // XCD-aware order: workgroup b runs on XCD b % 8; give every XCD ONE contiguous range of
        // tiles.  Tiles that are neighbours in the input write neighbouring runs of every digit
        // (a run is 24-48 elements: a fraction of a cache line at either end); on the same XCD
        // those partial lines meet in one L2 and leave it as full lines (stores that go past the L2 -- nt, sc1 -- take 13.6 - 19.6 ms
        // for the sort instead of 10.2: profiles/r06_sort_ab.log).
        const unsigned per = (first_tile + 7) / 8;          // FULL launches pass the number of complete tiles here
        tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
        if (tile >= first_tile) return;
    }
    scatter_tile<K, MODE, DESC, VB, KPT, TILE_KEYS, FULL>(L, tile, keys_in, keys_out,
            reinterpret_cast<const VT *>(vals_in_), reinterpret_cast<VT *>(vals_out_), n, shift, nblocks, table);
}

template <typename T, int AUX = 0>
__device__ __forceinline__ void store_elem(T v, __amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned scalar_bytes) {
    if constexpr (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)lane_bytes, (int)scalar_bytes, AUX);
    else {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, (int)lane_bytes, (int)scalar_bytes, AUX);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5, the default: ranks from ONE returning LDS atomic per key whose own return proves the order it was served in.
// A ranking UNIT is a half wave (32 lanes); its counter of digit d is a 64-bit word [ keys so far : 32 | lane bits : 32 ].  A lane
// adds (1 << 32) | (1 << (lane & 31)): the word it gets back holds, in its upper half, the number of the unit's keys with this digit
// that were counted before it -- earlier rounds AND the lanes of this round that were served first -- and in its lower half the
// bits of exactly those lanes of this round.  A lane that finds the bit of a HIGHER lane there was served out of lane order.
// If no lane does, the upper half IS the stable rank (of two lanes a < b with one digit, b first would show b's bit to a).
// A second, non-returning atomic takes the lane's bit out again.  Two LDS operations per key, and nothing is assumed about any
// instruction but the one that delivers the rank.  768 lanes x 16 keys (4-byte keys): 24 units x 256 words = 48 KiB, which the
// re-ordered tile then reuses; the units' tile offsets live on as 16-bit numbers (12 KiB): two workgroups per CU.
// Round 6 -- what happens when a lane WAS served out of order (no part has been seen doing it): the workgroup writes nothing, puts its
// tile on a list, and the small kernel that follows every scatter (radix_redo_kernel: a few workgroups that find the list empty and
// leave, ~3 us per pass) ranks the listed tiles by match words, which no order of service can upset; the event is counted in the
// sort's status (vexhip_sort_status).  Until round 5 the kernel trapped, which kills the context.  (Ranking the tile again inside
// the same kernel -- by ballots -- was tried first: the registers of a path that never runs took the sort from 10.2 to 14.3 ms.)  The ranks of this kernel were also delivered by two older generations (counter atomics checked on one tile in 16;
// "lean" kernels with separate order words): same time within 3 %, deleted (profiles/r05_sort_time*.json).
constexpr int UW = UB / kWave, UU = 2 * UW;
// words of a sort's status: tiles on the redo list of the current pass / tiles whose keys disagreed with the table / ticket of the redo kernel / redo tiles of all passes
enum { SORT_STATUS_ORDER = 0, SORT_STATUS_TABLE = 1, SORT_STATUS_TICKET = 2, SORT_STATUS_REDONE = 3, SORT_STATUS_WORDS = 4 };

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

template <typename K, int MODE, bool DESC, int VB, int KPT, bool WIDE, bool DISTRUST = false>
__global__ __launch_bounds__(UB, 6)
void radix_scatter_unit_kernel(const K *__restrict__ keys_in, K *__restrict__ keys_out,
        const void *__restrict__ vals_in_, void *__restrict__ vals_out_,
        long long n, int shift, unsigned nblocks, unsigned nfull, const unsigned *__restrict__ table, unsigned *__restrict__ status, unsigned *__restrict__ redo)
{
    typedef typename valtype<VB>::type VT;
    constexpr int TILE = UB * KPT;
    static_assert(TILE == tile_keys<K, VB>(), "the tile of the histogram kernel");
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
        L.gbase[t] = b0 - ds + (unsigned)TILE;               // biased by TILE: never negative, so that base + position stays a plain 32-bit sum
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
    unsigned long long *uw = s_word + unit * RADIX;
    bool out_of_order;
    {
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
        out_of_order = (seen >> (lane & 31)) != 0;             // a lane was served before a lower lane of its unit that hit the same word
        if constexpr (DISTRUST) out_of_order = true;           // (tests: every tile takes the path below)
    }
    if (__syncthreads_or(out_of_order ? 1 : 0)) {
        // the tile is handed to the kernel behind this launch, which ranks it by match words (radix_redo_kernel): nothing of it is written here
        if (t == 0) redo[atomicAdd(&status[SORT_STATUS_ORDER], 1u)] = tile;
        return;
    }

    bool mismatch = false;
    if (t < RADIX) {
        unsigned run = L.dstart[t];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            const unsigned c = (unsigned)(s_word[u * RADIX + t] >> 32);
            L.off[u][t] = (unsigned short)run;
            run += c;
        }
        mismatch = run - L.dstart[t] != cnt;                   // the table and the keys disagree: the input changed between the pass's kernels
    }
    if (__syncthreads_or(mismatch ? 1 : 0)) {
        // (a caller's error -- nothing sensible can be written: the tile is dropped and the event counted; until round 5: a trap)
        if (t == 0) atomicAdd(&status[SORT_STATUS_TABLE], 1u);
        return;
    }

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
            store_elem<K>(kk, rk, e * (unsigned)sizeof(K), (unsigned)(k * UB) * (unsigned)sizeof(K));
            if constexpr (VB != 0) store_elem<VT>(s_vals[t + k * UB], rv, e * (unsigned)VB, (unsigned)(k * UB) * (unsigned)VB);
        }
    }
}

// Behind every scatter of the unit kernel: the tiles it declined (a lane served out of order; tests: all of them), ranked by match words.
template <typename K, int MODE, bool DESC, int VB, int KPT, int TILE_KEYS>
__global__ __launch_bounds__(RB, 8)
void radix_redo_kernel(const K *__restrict__ keys_in, K *__restrict__ keys_out,
        const void *__restrict__ vals_in_, void *__restrict__ vals_out_,
        long long n, int shift, unsigned nblocks, const unsigned *__restrict__ table, const unsigned *__restrict__ status, const unsigned *__restrict__ redo,
        unsigned *__restrict__ total)
{
    typedef typename valtype<VB>::type VT;
    __shared__ scatter_lds<K, VB, KPT> L;
    const unsigned count = status[SORT_STATUS_ORDER];
    for (unsigned i = blockIdx.x; i < count; i += gridDim.x) {
        scatter_tile<K, MODE, DESC, VB, KPT, TILE_KEYS, TILE_KEYS == RB * KPT>(L, redo[i], keys_in, keys_out,
                reinterpret_cast<const VT *>(vals_in_), reinterpret_cast<VT *>(vals_out_), n, shift, nblocks, table);
        __syncthreads();
    }
    // the list is emptied for the next pass by the LAST workgroup to get here (all have read `count` by then: a ticket); the sort's total stays
    if (threadIdx.x == 0) {
        unsigned *ticket = const_cast<unsigned *>(status) + 2;
        if (atomicAdd(ticket, 1u) + 1u == gridDim.x) { *ticket = 0u; atomicAdd(total, count); const_cast<unsigned *>(status)[SORT_STATUS_ORDER] = 0u; }
    }
}

extern int g_sort_rank;

// rank: -1 / 6 the unit scatter (one returning atomic per key that proves its own order: the default); 0 ranks from match words in
// every tile (round 2's kernel: ordered by construction, 25 % slower; A/B and tests); 7 the unit scatter with every tile ranked a
// second time by ballots (tests of that path)
template <typename K, int MODE, bool DESC, int VB>
int sort_passes(hipStream_t s, K *keys, K *keys_tmp, void *vals, void *vals_tmp, int64_t n, unsigned *tmp, unsigned *status, unsigned *redo, int rank) {
    constexpr int TILE = tile_keys<K, VB>();
    constexpr int UKPT = unit_keys_per_lane((int)sizeof(K), VB);
    constexpr int SLOTS = slots_per_lane<K, VB>();
    constexpr bool fills = TILE == RB * SLOTS;               // the tile fills the slots of the match-word kernel
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
        radix_hist_kernel<K, MODE, DESC, TILE><<<(nblocks + 7) / 8 * 8, HB, 0, s>>>(src, n, shift, nblocks, table, vec_ok);
        VEXHIP_LAUNCH_CHECK();
        if (int rc = scan_exclusive_u32_tmp(s, table, table, tn, scan_tmp)) return rc;
        const unsigned nfull = (unsigned)(n / TILE);
        const unsigned grid = (nfull + 7) / 8 * 8;
        if (nfull && rank != 0) {
            if (rank == 7) {
                if (wide) radix_scatter_unit_kernel<K, MODE, DESC, VB, UKPT, true, true><<<grid, UB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table, status, redo);
                else      radix_scatter_unit_kernel<K, MODE, DESC, VB, UKPT, false, true><<<grid, UB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table, status, redo);
            } else {
                if (wide) radix_scatter_unit_kernel<K, MODE, DESC, VB, UKPT, true><<<grid, UB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table, status, redo);
                else      radix_scatter_unit_kernel<K, MODE, DESC, VB, UKPT, false><<<grid, UB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table, status, redo);
            }
            VEXHIP_LAUNCH_CHECK();
            radix_redo_kernel<K, MODE, DESC, VB, SLOTS, TILE><<<std::min(nfull, 512u), RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, table, status, redo, status + SORT_STATUS_REDONE);
            VEXHIP_LAUNCH_CHECK();
        } else if (nfull) {
            // match words in every tile: complete tiles that fill the kernel's slots take the mask-free form
            if constexpr (fills) radix_scatter_kernel<K, MODE, DESC, VB, SLOTS, TILE, true><<<grid, RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table);
            else radix_scatter_kernel<K, MODE, DESC, VB, SLOTS, TILE, false><<<nfull, RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, 0u, table);
            VEXHIP_LAUNCH_CHECK();
        }
        if (nfull < nblocks) {
            // the ragged last tile: one workgroup, ranked by the match words
            radix_scatter_kernel<K, MODE, DESC, VB, SLOTS, TILE, false><<<1, RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nblocks, nfull, table);
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
int sort_dispatch(hipStream_t s, int desc, int vb, void *keys, void *keys_tmp, void *vals, void *vals_tmp, int64_t n, void *tmp, unsigned *status, unsigned *redo, int rank) {
#define GO(DESC, VB) return sort_passes<K, MODE, DESC, VB>(s, (K *)keys, (K *)keys_tmp, vals, vals_tmp, n, (unsigned *)tmp, status, redo, rank)
    if (desc) { if (vb == 0) GO(true, 0); if (vb == 4) GO(true, 4); if (vb == 8) GO(true, 8); }
    else      { if (vb == 0) GO(false, 0); if (vb == 4) GO(false, 4); if (vb == 8) GO(false, 8); }
#undef GO
    return fail(__FILE__, __LINE__, "value_bytes must be 0, 4 or 8");
}

int g_sort_rank = -1;               // -1: the default (6: the unit scatter); 0: match words in every tile; 7: the unit scatter, every tile ranked again by ballots (tests)

// the status words of a sort live at the end of its workspace (vexhip_sort_tmp_bytes reserves them)
// [ table | scan workspace | redo list (one entry per tile) | status words ]
inline size_t sort_redo_elems(int64_t n) { const int64_t smallest_tile = UB * 4; return (size_t)((n + smallest_tile - 1) / smallest_tile + 4); }     // (8-byte key, 8-byte value): 3072 elements
inline size_t sort_tmp_elems(int64_t n) {
    const int64_t smallest_tile = UB * 4;
    const int64_t nblocks = (n + smallest_tile - 1) / smallest_tile;
    const int64_t tn = nblocks * RADIX;
    return (size_t)((tn + 3) / 4 * 4 + (int64_t)scan_tmp_elems_u32(tn) + 4) + sort_redo_elems(n) + SORT_STATUS_WORDS;
}

} // namespace
} // namespace vexhip

using namespace vexhip;

extern "C" {

int vexhip_sort_set_rank(int mode) { g_sort_rank = mode; return 0; }

size_t vexhip_sort_tmp_bytes(int key_dtype, int64_t n) {
    (void)key_dtype;
    return sizeof(unsigned) * sort_tmp_elems(n);
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
    const int ar = g_sort_rank < 0 ? 6 : g_sort_rank;
    unsigned *status = static_cast<unsigned *>(tmp) + sort_tmp_elems(n) - SORT_STATUS_WORDS;
    unsigned *redo = status - sort_redo_elems(n);
    VEXHIP_TRY(hipMemsetAsync(status, 0, SORT_STATUS_WORDS * sizeof(unsigned), s));
    switch (key_dtype) {
        case VEXHIP_U32: return sort_dispatch<unsigned, KEY_UNSIGNED>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, status, redo, ar);
        case VEXHIP_I32: return sort_dispatch<unsigned, KEY_SIGNED>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, status, redo, ar);
        case VEXHIP_F32: return sort_dispatch<unsigned, KEY_FLOAT>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, status, redo, ar);
        case VEXHIP_U64: return sort_dispatch<unsigned long long, KEY_UNSIGNED>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, status, redo, ar);
        case VEXHIP_I64: return sort_dispatch<unsigned long long, KEY_SIGNED>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, status, redo, ar);
        case VEXHIP_F64: return sort_dispatch<unsigned long long, KEY_FLOAT>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, status, redo, ar);
    }
    return fail(__FILE__, __LINE__, "unknown key dtype");
}

// What the last vexhip_sort on this workspace met (waits for the stream): tiles whose ranks were taken a second time because a lane of an
// LDS atomic had been served out of lane order (the result is correct all the same), and tiles dropped because their keys no longer
// matched the pass's histogram (the caller changed the input while the sort ran: the result is NOT sorted -- an error is returned).
int vexhip_sort_status(int dev, void *stream, int64_t n, const void *tmp, int64_t *reranked_tiles, int64_t *dropped_tiles)
{
    VEXHIP_REQUIRE(tmp && n >= 0, "bad argument");
    if (reranked_tiles) *reranked_tiles = 0;
    if (dropped_tiles) *dropped_tiles = 0;
    if (n <= 1) return 0;
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    unsigned h[SORT_STATUS_WORDS] = {0, 0, 0, 0};
    VEXHIP_TRY(hipMemcpyAsync(h, static_cast<const unsigned *>(tmp) + sort_tmp_elems(n) - SORT_STATUS_WORDS, sizeof(h), hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    if (reranked_tiles) *reranked_tiles = h[SORT_STATUS_REDONE];
    if (dropped_tiles) *dropped_tiles = h[SORT_STATUS_TABLE];
    if (h[SORT_STATUS_TABLE]) return fail(__FILE__, __LINE__, "vexhip_sort: the keys changed while the sort ran (a pass's histogram and its scatter disagree): the result is not sorted");
    return 0;
}

} // extern "C"

VEXHIP_WARM_TU(sort)
