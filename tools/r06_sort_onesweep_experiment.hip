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
// Where a tile ranked by match words finds its global bases: the scanned table of a table pass, or -- in a chained pass (below) -- the
// bases of its eighth + the counts of the tiles before it in the chain (every word INCLUSIVE once the chained kernel has finished); a
// chained pass also wants the tile's keys counted, by the eighth they go to and their NEXT digit, for the pass that follows.
enum : unsigned { CH_AGG = 1u << 30, CH_INCL = 2u << 30, CH_MASK = (1u << 30) - 1u };
constexpr int CHAINS = 8;                    // one chain of tiles per XCD
constexpr int ROWS = CHAINS + 1;             // + the ragged last tile
struct base_src {
    const unsigned *table; unsigned nblocks;         // table pass: digit-major, one column per tile
    const unsigned *base, *chain; unsigned per, nfull;  // chained pass: base[ROWS][RADIX], chain[tile][RADIX]
    unsigned *next; int next_shift; unsigned span, full; // counts of the next pass [ROWS][RADIX] (NULL: none wanted), positions per eighth, positions in complete tiles
};
__device__ __forceinline__ unsigned row_of(unsigned g, unsigned span, unsigned full) { return g >= full ? (unsigned)CHAINS : g / span; }

template <typename K, int MODE, bool DESC, int VB, int KPT, int TILE_KEYS, bool FULL>
__device__ __forceinline__ void scatter_tile(scatter_lds<K, VB, KPT> &L, const unsigned tile,
        const K *__restrict__ keys_in, K *__restrict__ keys_out,
        const typename valtype<VB>::type *__restrict__ vals_in, typename valtype<VB>::type *__restrict__ vals_out,
        long long n, int shift, const base_src &B)
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
        unsigned b;
        if (B.table) b = B.table[(size_t)t * B.nblocks + tile];
        else if (tile >= B.nfull) b = B.base[CHAINS * RADIX + t];
        else b = B.base[(tile / B.per) * RADIX + t] + (B.chain[(size_t)tile * RADIX + t] & CH_MASK) - count;
        L.gbase[t] = b - dstart;
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
            if (B.next) atomicAdd(&B.next[row_of(g, B.span, B.full) * RADIX + ((unsigned)(to_ordered<K, MODE, DESC>(kk) >> (B.next_shift & (8 * (int)sizeof(K) - 1))) & (RADIX - 1))], 1u);
        }
    } else {
        for (int i = t; i < nvalid; i += RB) {
            K kk = s_keys[i];
            unsigned d = (unsigned)(to_ordered<K, MODE, DESC>(kk) >> shift) & (RADIX - 1);
            unsigned g = L.gbase[d] + (unsigned)i;
            keys_out[g] = kk;
            if constexpr (VB != 0) vals_out[g] = s_vals[i];
            if (B.next) atomicAdd(&B.next[row_of(g, B.span, B.full) * RADIX + ((unsigned)(to_ordered<K, MODE, DESC>(kk) >> (B.next_shift & (8 * (int)sizeof(K) - 1))) & (RADIX - 1))], 1u);
        }
    }
}

// FULL = true: launched over the complete tiles (first_tile = their number; rank mode 0: the whole sort by match words, A/B and
// tests); FULL = false: one workgroup for the ragged last tile (first_tile = its index).
template <typename K, int MODE, bool DESC, int VB, int KPT, int TILE_KEYS, bool FULL>
__global__ __launch_bounds__(RB, 8)
void radix_scatter_kernel(const K *__restrict__ keys_in, K *__restrict__ keys_out,
        const void *__restrict__ vals_in_, void *__restrict__ vals_out_,
        long long n, int shift, unsigned first_tile, const base_src B)
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
            reinterpret_cast<const VT *>(vals_in_), reinterpret_cast<VT *>(vals_out_), n, shift, B);
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
enum { SORT_STATUS_ORDER = 0, SORT_STATUS_TABLE = 1, SORT_STATUS_TICKET = 2, SORT_STATUS_REDONE = 3, SORT_STATUS_CHAIN = 4, SORT_STATUS_WORDS = 8 };

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

// ---------------------------------------------------------------------------------------------------------------------------
// Round 6, the CHAINED pass (36 bytes per 4-byte key and sort instead of 48): a pass no longer reads the keys a second time for the
// tile-by-digit table.  A tile's digit counts come out of its ranking, and where its keys of digit d go is
//      base[e][d]     where the keys of digit d of the tile's EIGHTH e start (an eighth = one contiguous range of tiles = one chain)
//    + the counts of digit d of the tiles before it in its chain        -- DECOUPLED LOOK-BACK along the chain.
// Why eight chains and not one: tiles start every ~22 ns, a word takes 1-3 us from CU to CU under load, so that a look-back along
// ONE chain meets >100 predecessors x 256 words -- more bytes than the tile (measured with rocPRIM's onesweep: 18.6 ms,
// profiles/r06_sort_rocprim.json).  A chain per XCD starts a tile every ~180 ns and its look-backs meet 6-9 predecessors
// (profiles/r06_sort_chain.json), and its words never leave the XCD's L2: a chain is served by the workgroups of ONE XCD (they read
// HW_REG_XCC_ID and take the chain's tiles by ticket), the words are stored workgroup-scope (they stay in that L2) and read
// agent-scope (past the CU's L1).  With words stored agent-scope (write-through) the same kernel is 0.3 ms per pass slower.
// A chain belongs to the XCD that claims it first (owner[]); an XCD prefers the chain of its own number and takes another one only
// when nothing is left of its own and the other was never claimed: whatever the placement of the workgroups, every tile is served.
// A word [state:2 | count:30] is one 4-byte atomic: state and count travel together, no fence.  AGGREGATE = the tile's count of the
// digit, INCLUSIVE = the count of the tile and all before it in the chain.
// base[e][d] of the NEXT pass needs the digit counts of its eighths: every key is counted once more, in LDS, by the eighth it is
// written to and its next digit (a workgroup keeps the counts over all its tiles and adds them to the global table at its end);
// only the first pass reads the keys beforehand (radix_count_kernel).
constexpr unsigned CH_SPIN_LIMIT = 1u << 22; // polls of one word (~0.5 us each: seconds) before the sort gives up with an error (never a trap)
constexpr int CH_WINDOW = 4;                 // predecessors read per round of the look-back
__device__ __forceinline__ unsigned ld_chain(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_chain(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }    // orders LDS traffic only: loads from memory stay in flight across it
__device__ __forceinline__ unsigned xcc_id() { return (unsigned)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u; }   // HW_REG_XCC_ID[3:0]
#if !defined(__gfx950__) && !defined(__gfx942__) && defined(__HIP_DEVICE_COMPILE__)
#error "the chained sort orders LDS traffic with `s_waitcnt lgkmcnt(0)` + `s_barrier` and reads HW_REG_XCC_ID: gfx942 / gfx950 only (ARCH in csrc/Makefile)"
#endif

// the digit counts of pass 0 by eighth: the only extra read of the keys in a chained sort
template <typename K, int MODE, bool DESC, int TILE, int UN = 6>
__global__ __launch_bounds__(HB)
void radix_count_kernel(const K *__restrict__ keys, long long n, int shift, unsigned ntiles, unsigned nfull, unsigned per, unsigned *__restrict__ count, int vec_ok)
{
    constexpr int VN = 16 / (int)sizeof(K);
    typedef K vtype __attribute__((ext_vector_type(16 / sizeof(K))));
    __shared__ unsigned s_h[ROWS][RADIX];
    for (int i = threadIdx.x; i < ROWS * RADIX; i += HB) (&s_h[0][0])[i] = 0;
    __syncthreads();
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        unsigned *h = s_h[tile >= nfull ? (unsigned)CHAINS : tile / per];
        const long long base = (long long)tile * TILE;
        const int cnt = (int)((n - base < TILE) ? (n - base) : TILE);
        int done = 0;
        if (vec_ok) {
            const int nv = cnt / VN;
            const vtype *kv = reinterpret_cast<const vtype *>(keys + base);
            int v = threadIdx.x;
            for (; v + (UN - 1) * HB < nv; v += UN * HB) {
                vtype q[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) q[u] = __builtin_nontemporal_load(kv + v + u * HB);
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int j = 0; j < VN; ++j) count_digit(h, (unsigned)(to_ordered<K, MODE, DESC>(q[u][j]) >> shift) & (RADIX - 1));
            }
            for (; v < nv; v += HB) {
                vtype q = __builtin_nontemporal_load(kv + v);
#pragma unroll
                for (int j = 0; j < VN; ++j) count_digit(h, (unsigned)(to_ordered<K, MODE, DESC>(q[j]) >> shift) & (RADIX - 1));
            }
            done = nv * VN;
        }
        for (int i = done + threadIdx.x; i < cnt; i += HB) count_digit(h, (unsigned)(to_ordered<K, MODE, DESC>(keys[base + i]) >> shift) & (RADIX - 1));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ROWS * RADIX; i += HB) { const unsigned v = (&s_h[0][0])[i]; if (v) atomicAdd(&count[i], v); }
}

// counts by eighth -> where the keys of digit d of eighth e start; clears the counts of the following pass, the tickets and the owners
__global__ __launch_bounds__(RADIX)
void radix_bases_kernel(const unsigned *__restrict__ count, unsigned *__restrict__ base, unsigned *__restrict__ next, unsigned *__restrict__ ticket_owner)
{
    __shared__ unsigned s_w[RADIX / kWave];
    const int t = threadIdx.x, wave = t / kWave, lane = t % kWave;
    unsigned c[ROWS], total = 0;
#pragma unroll
    for (int e = 0; e < ROWS; ++e) { c[e] = count[e * RADIX + t]; total += c[e]; }
    unsigned inc = total;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) { const unsigned u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane == kWave - 1) s_w[wave] = inc;
    __syncthreads();
    unsigned run = inc - total;
    for (int w = 0; w < wave; ++w) run += s_w[w];
#pragma unroll
    for (int e = 0; e < ROWS; ++e) { base[e * RADIX + t] = run; run += c[e]; next[e * RADIX + t] = 0u; }
    if (t < 2 * CHAINS) ticket_owner[t] = 0u;
}

template <typename K, int VB, int KPT>
struct sweep_lds {
    static constexpr int TILE = UB * KPT;
    static constexpr int TILE_BYTES = TILE * ((int)sizeof(K) + VB);
    static constexpr int WORD_BYTES = UU * RADIX * 8;
    static constexpr int RAW_WORDS = ((TILE_BYTES > WORD_BYTES ? TILE_BYTES : WORD_BYTES) + 15) / 16 * 2;
    unsigned long long raw[RAW_WORDS];      // the units' counter words while the keys are ranked, then the re-ordered tile
    unsigned short off[UU][RADIX];          // a unit's first key of a digit, counted from the first key of its PART (8 units) of that digit
    unsigned gbase[RADIX];                  // global position of tile position 0 of a digit's run (+ TILE)
    unsigned nrec[RADIX];                   // next pass: [tile position from which the run lies in the following eighth:16 | eighth:4 | following eighth:4]
    unsigned short ptot[UU / 8][RADIX];     // keys of a digit in a part
    unsigned short pbase[UU / 8][RADIX];    // tile position of a part's first key of a digit
    unsigned short cnt[RADIX], dstart[RADIX];   // the tile's count of a digit, tile position of its first key
    unsigned flags;                         // 1: a lane of this tile was served out of lane order
    unsigned next_chain, next_j;            // the tile after this one (next_j = ~0u: none)
    unsigned cur_c, have_pending;           // the finder lane's: the chain it takes tickets from, whether one is asked for
    unsigned nexth[ROWS][RADIX];            // next pass: keys by eighth and next digit, over all tiles of this workgroup
};                                          // (everything before nexth lies below 64 KiB: one address register + immediate offsets)

typedef unsigned uint4v __attribute__((ext_vector_type(4)));
// four chain words of one tile (a lane's four digits) past the CU's L1; the asm is not hoisted out of a polling loop, and it waits itself
__device__ __forceinline__ uint4v ld_chain4(const uint4v *p) {
    uint4v v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// one round of the look-back: the rows of the eight tiles before `p`'s (1 KiB apart) in flight together, then one wait
__device__ __forceinline__ void ld_chain4_round(const uint4v *p, uint4v (&v)[8]) {
    const uint4v *q = p - 4 * (RADIX / 4);
    asm volatile("global_load_dwordx4 %0, %8, off offset:-1024 sc1\n\tglobal_load_dwordx4 %1, %8, off offset:-2048 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:-3072 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:-4096 sc1\n\t"
                 "global_load_dwordx4 %4, %9, off offset:-1024 sc1\n\tglobal_load_dwordx4 %5, %9, off offset:-2048 sc1\n\t"
                 "global_load_dwordx4 %6, %9, off offset:-3072 sc1\n\tglobal_load_dwordx4 %7, %9, off offset:-4096 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(p), "v"(q) : "memory");
}
constexpr int CH_PAD_ROWS = 8;               // rows in front of the chain words: a round may read that far before a chain's first tile (and ignores what it finds)

// One trip of a workgroup = one tile:   zero the counter words | rank | counts by part (all lanes; meanwhile ONE lane finds out which tile
// comes next) | wave 0: the tile's counts, published; places of the parts | ranks -> tile positions | re-order in LDS; the other waves
// ask for the next tile's keys, wave 0 looks back along the chain (four digits per lane, 16-byte loads) and publishes the inclusive
// counts | write-out (+ the counts of the next pass) |.  Seven barriers, all of them LDS-only (loads stay in flight across them).
template <typename K, int MODE, bool DESC, int VB, int KPT, bool WIDE, bool NEXT, bool DISTRUST = false>
__global__ __launch_bounds__(UB, 6)
void radix_onesweep_kernel(const K *__restrict__ keys_in, K *__restrict__ keys_out,
        const void *__restrict__ vals_in_, void *__restrict__ vals_out_,
        int shift, unsigned nfull, unsigned per, const unsigned *__restrict__ base, unsigned *__restrict__ next,
        unsigned *__restrict__ chain, unsigned *__restrict__ ticket, unsigned *__restrict__ owner,
        unsigned *__restrict__ status, unsigned *__restrict__ redo)
{
    typedef typename valtype<VB>::type VT;
    constexpr int TILE = UB * KPT;
    constexpr int W = 8;                                        // predecessors per round of the look-back
    static_assert(TILE == tile_keys<K, VB>(), "the tile of the table passes");
    static_assert(UB == 3 * RADIX && UU == 24, "three parts of eight units: a lane per (digit, part)");
    __shared__ __attribute__((aligned(16))) sweep_lds<K, VB, KPT> L;
    const VT *__restrict__ vals_in = reinterpret_cast<const VT *>(vals_in_);
    VT *__restrict__ vals_out = reinterpret_cast<VT *>(vals_out_);
    K *s_keys = reinterpret_cast<K *>(L.raw);
    VT *s_vals = reinterpret_cast<VT *>(reinterpret_cast<char *>(L.raw) + (size_t)TILE * sizeof(K));
    unsigned long long *s_word = L.raw;

    const int t0 = threadIdx.x;
    const int wave_s = __builtin_amdgcn_readfirstlane(t0 / kWave);      // (a scalar: the lane number is put together again in every trip)
    const unsigned span = per * (unsigned)TILE, full = nfull * (unsigned)TILE;
    const unsigned me = xcc_id() + 1u;                          // this workgroup's XCD, as it stands in owner[]
    constexpr int FINDER = UB - kWave;                          // the lane that takes the tickets: lane 0 of the last wave (wave 0 has the look-back)

    // the chain words through a buffer descriptor: byte offset of (tile, digit) = tile * 1024 (scalar) + 4 * digit (at most 2^31 / 3072 tiles:
    // 0.7 GB); loads bypass the CU's L1 (sc1: served by the XCD's L2), stores stay in that L2 (sc0).  Only where the offset is uniform and
    // the load stands outside every polling loop (the builtin is no atomic: a poll built on it is hoisted out of its loop).
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(chain, 0, -1, 0x00020000);

    __amdgpu_buffer_rsrc_t rk, rv;
    if constexpr (!WIDE) {
        rk = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(keys_out) - (long long)TILE * (long long)sizeof(K), 0, -1, 0x00020000);
        if constexpr (VB != 0) rv = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(vals_out) - (long long)TILE * (long long)VB, 0, -1, 0x00020000);
    }

    K key[KPT];
    VT val[VB ? KPT : 1];
    // (the keys through a descriptor of the TILE -- scalar arithmetic -- and a 32-bit lane offset: no 64-bit address per lane lives across the trip)
    auto fetch = [&](int t, unsigned tile_) {
        const int lane = t % kWave, unit = 2 * (t / kWave) + (lane >> 5);
        const int upos = unit * (32 * KPT) + (lane & 31);
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<K *>(keys_in) + (size_t)tile_ * TILE, 0, TILE * (int)sizeof(K), 0x00020000);
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            if constexpr (sizeof(K) == 4) key[k] = (K)__builtin_amdgcn_raw_buffer_load_b32(rt, (upos + k * 32) * 4, 0, 2);
            else { typedef unsigned u2 __attribute__((ext_vector_type(2))); key[k] = __builtin_bit_cast(K, (u2)__builtin_amdgcn_raw_buffer_load_b64(rt, (upos + k * 32) * 8, 0, 2)); }
        }
        if constexpr (VB != 0) {
            const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<VT *>(vals_in) + (size_t)tile_ * TILE, 0, TILE * VB, 0x00020000);
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                if constexpr (VB == 4) val[k] = (VT)__builtin_amdgcn_raw_buffer_load_b32(rq, (upos + k * 32) * 4, 0, 2);
                else { typedef unsigned u2 __attribute__((ext_vector_type(2))); val[k] = __builtin_bit_cast(VT, (u2)__builtin_amdgcn_raw_buffer_load_b64(rq, (upos + k * 32) * 8, 0, 2)); }
            }
        }
    };
    auto chain_len = [&](unsigned c) { const unsigned first = c * per; return first >= nfull ? 0u : (nfull - first < per ? nfull - first : per); };

#ifdef VEXHIP_SORT_PROFILE
    unsigned long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pc = __builtin_readcyclecounter();
#define PSTAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); pt[i] += now_ - pc; pc = now_; } while (0)
#else
#define PSTAMP(i) do {} while (0)
#endif

    // ---- which tile next (the finder lane): a ticket of a chain this XCD owns -- the chain it already serves; else the chain of its own
    // number, else any that nobody has claimed.  `pending` is the ticket asked for one trip ahead (an atomic's answer takes 1-2 us).
    unsigned pending = 0;                                       // (the finder's only register across trips; the rest of its state lives in LDS)
    unsigned &cur_c = L.cur_c, &have_pending = L.have_pending;
    if (t0 == FINDER) { cur_c = me - 1u; have_pending = 0u; }
    auto find_next = [&]() {
        unsigned c = cur_c, nj = ~0u;
#pragma nounroll
        for (int tries = 0; tries <= CHAINS && nj == ~0u; ++tries) {
            if (have_pending) { have_pending = 0u; if (pending < chain_len(c)) { nj = pending; break; } }
            else if (chain_len(c)) {
                unsigned o = __hip_atomic_load(&owner[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (o == 0u) { unsigned expect = 0u; o = __hip_atomic_compare_exchange_strong(&owner[c], &expect, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? me : expect; }
                if (o == me) { const unsigned q = atomicAdd(&ticket[c], 1u); if (q < chain_len(c)) { nj = q; break; } }
            }
            c = (c + 1u) % CHAINS;
        }
        cur_c = c; L.next_chain = c; L.next_j = nj;
        if (nj != ~0u) { pending = atomicAdd(&ticket[c], 1u); have_pending = 1u; }
    };

    for (int i = t0; i < ROWS * RADIX; i += UB) (&L.nexth[0][0])[i] = 0u;
    if (t0 == FINDER) find_next();
    lds_barrier();
    unsigned e = L.next_chain, j = L.next_j;
    if (j != ~0u) fetch(t0, e * per + j);
    lds_barrier();                                                  // (everybody has read the first tile before the finder names the second)

    while (j != ~0u) {
        // (the lane number is opaque in every trip: what derives from it -- LDS addresses, offsets -- is computed again instead of being
        //  kept in registers across the trip, where it would spill: a scratch reload waits for every load in flight)
        int t = wave_s * kWave + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(t));
        const int lane = t % kWave, wave = wave_s, unit = 2 * wave + (lane >> 5);
        const unsigned tile = e * per + j;

        {   // zero the counter words
            uint4v z = {0u, 0u, 0u, 0u};
            asm volatile("" : "+v"(z));                           // (made here: a zero kept across the trip is spilled and reloaded behind a wait for every load)
#pragma unroll
            for (int q = 0; q < UU * RADIX * 8 / 16 / UB; ++q) reinterpret_cast<uint4v *>(s_word)[t + q * UB] = z;
            if (t == 0) L.flags = 0u;
        }
        lds_barrier();
        PSTAMP(0);

        unsigned rr[KPT];
        {
            unsigned long long *uw = s_word + unit * RADIX;
            const unsigned mybit = 1u << (lane & 31);
            const unsigned long long add = (1ull << 32) | mybit;
            unsigned seen = 0;
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const unsigned d = (unsigned)(to_ordered<K, MODE, DESC>(key[k]) >> shift) & (RADIX - 1);
                const unsigned long long old = atomicAdd(&uw[d], add);
                atomicAdd(&uw[d], (unsigned long long)(0ull - (unsigned long long)mybit));
                rr[k] = (unsigned)(old >> 32);
                seen |= (unsigned)old;
            }
            bool out_of_order = (seen >> (lane & 31)) != 0;      // served before a lower lane of its unit that hit the same word
            if constexpr (DISTRUST) out_of_order = true;
            if (__any(out_of_order) && lane == 0) atomicOr(&L.flags, 1u);
        }
        PSTAMP(1);
        lds_barrier();
        const bool declined = (L.flags & 1u) != 0;               // (the counts below are right whatever the order of service)

        {   // a lane per (digit, part): the part's eight units, first to last
            const int dg = t & (RADIX - 1), part = t >> 8;
            unsigned run = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned c = (unsigned)(s_word[(part * 8 + u) * RADIX + dg] >> 32);
                L.off[part * 8 + u][dg] = (unsigned short)run;
                run += c;
            }
            L.ptot[part][dg] = (unsigned short)run;
        }
        if (t == FINDER) find_next();                             // which tile after this one; the ticket after that is asked for now
        PSTAMP(2);
        lds_barrier();
        const unsigned ne = L.next_chain, nj = L.next_j;

        if (wave == 0) {
            // lane l: digits 4l .. 4l+3 -- the tile's counts (published at once, one 16-byte store), where each digit and each of its parts starts
            typedef unsigned short us4 __attribute__((ext_vector_type(4)));
            const us4 p0 = *reinterpret_cast<const us4 *>(&L.ptot[0][4 * lane]), p1 = *reinterpret_cast<const us4 *>(&L.ptot[1][4 * lane]), p2 = *reinterpret_cast<const us4 *>(&L.ptot[2][4 * lane]);
            uint4v cnt;
#pragma unroll
            for (int c = 0; c < 4; ++c) cnt[c] = (unsigned)p0[c] + p1[c] + p2[c];
            const unsigned st = j == 0 ? CH_INCL : CH_AGG;
            const uint4v pub = {st | cnt[0], st | cnt[1], st | cnt[2], st | cnt[3]};
            __builtin_amdgcn_raw_buffer_store_b128(pub, rc, lane * 16, (int)(tile * (RADIX * 4u)), 1);
            unsigned inc = cnt[0] + cnt[1] + cnt[2] + cnt[3];
            const unsigned mine_total = inc;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {                 // (by the trip's own lane number: __shfl_up's is hoisted out of the loop and spills)
                const unsigned u = (unsigned)__builtin_amdgcn_ds_bpermute(((lane - o) & (kWave - 1)) << 2, (int)inc);
                if (lane >= o) inc += u;
            }
            unsigned ds = inc - mine_total;
            us4 c16, d16, b1, b2;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                c16[c] = (unsigned short)cnt[c]; d16[c] = (unsigned short)ds;
                b1[c] = (unsigned short)(ds + p0[c]); b2[c] = (unsigned short)(ds + p0[c] + p1[c]);
                ds += cnt[c];
            }
            *reinterpret_cast<us4 *>(&L.cnt[4 * lane]) = c16;
            *reinterpret_cast<us4 *>(&L.dstart[4 * lane]) = d16;
            *reinterpret_cast<us4 *>(&L.pbase[0][4 * lane]) = d16;
            *reinterpret_cast<us4 *>(&L.pbase[1][4 * lane]) = b1;
            *reinterpret_cast<us4 *>(&L.pbase[2][4 * lane]) = b2;
        }
        PSTAMP(3);
        lds_barrier();
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            asm volatile("" : "+v"(key[k]));                      // (the digit is computed again: kept from the ranking it would cost a register per key)
            const unsigned d = (unsigned)(to_ordered<K, MODE, DESC>(key[k]) >> shift) & (RADIX - 1);
            rr[k] += (unsigned)L.off[unit][d] + (unsigned)L.pbase[unit >> 3][d];
        }
        PSTAMP(4);
        // (the tile goes where the counter words are: nobody has read those since the barrier before last)
        if (!declined) {
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                s_keys[rr[k]] = key[k];
                if constexpr (VB != 0) s_vals[rr[k]] = val[k];
            }
        }
        if (wave != 0) {
            if (nj != ~0u) fetch(t, ne * per + nj);                // the next tile's keys travel during the look-back and the write-out
        } else {
            // ---- the look-back: the counts of this lane's four digits in the tiles before this one, each back to the nearest tile that knows
            // its inclusive count
            unsigned excl[4] = {0u, 0u, 0u, 0u}, done = j == 0 ? 15u : 0u, back = 0;
#ifdef VEXHIP_SORT_NOLOOKBACK
            done = 15u;                 // (timing experiment: no look-back; the keys go where they would go if every tile had this tile's counts)
#endif
            bool lost = false;
            const uint4v *row = reinterpret_cast<const uint4v *>(chain + (size_t)tile * RADIX) + lane;
            const uint4v nothing = {CH_INCL, CH_INCL, CH_INCL, CH_INCL};      // before the chain's first tile
            static_assert(W == 8 && CH_PAD_ROWS >= W, "a round is eight loads; the rows in front of the first tile exist");
            while (done != 15u) {
                uint4v wv[W];
#ifdef VEXHIP_SORT_PROFILE
                if (lane == 0) pt[8] += 16;
#endif
                ld_chain4_round(row - (size_t)back * (RADIX / 4), wv);
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    if (done != 15u) {
                        uint4v v = (j > back + (unsigned)i) ? wv[i] : nothing;
                        unsigned spins = 0;
                        for (;;) {
                            unsigned missing = 0;
#pragma unroll
                            for (int c = 0; c < 4; ++c) missing |= (!((done >> c) & 1u) && (v[c] >> 30) == 0u) ? 1u : 0u;
                            if (!missing) break;
                            __builtin_amdgcn_s_sleep(1);
#ifdef VEXHIP_SORT_PROFILE
                            if (lane == 0) { pt[9] += 16; if (spins == 0) pt[10] += 16; }
#endif
                            v = ld_chain4(row - (size_t)(back + i + 1) * (RADIX / 4));
                            if (++spins > CH_SPIN_LIMIT) { lost = true; v = nothing; }
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (!((done >> c) & 1u)) { excl[c] += v[c] & CH_MASK; if ((v[c] >> 30) == 2u) done |= 1u << c; }
                        }
                    }
                }
                back += W;
            }
            typedef unsigned short us4 __attribute__((ext_vector_type(4)));
            const us4 c16 = *reinterpret_cast<const us4 *>(&L.cnt[4 * lane]), d16 = *reinterpret_cast<const us4 *>(&L.dstart[4 * lane]);
            if (j != 0) {
                const uint4v pub = {CH_INCL | (excl[0] + c16[0]), CH_INCL | (excl[1] + c16[1]), CH_INCL | (excl[2] + c16[2]), CH_INCL | (excl[3] + c16[3])};
                __builtin_amdgcn_raw_buffer_store_b128(pub, rc, lane * 16, (int)(tile * (RADIX * 4u)), 1);
            }
            if (lost) atomicAdd(&status[SORT_STATUS_CHAIN], 1u);
            const uint4v b4 = *reinterpret_cast<const uint4v *>(base + e * RADIX + 4 * lane);
            uint4v gb, nr;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned cnt = c16[c], ds = d16[c];
#ifdef VEXHIP_SORT_NOLOOKBACK
                unsigned g0 = b4[c] + j * cnt; if (g0 > full - (unsigned)TILE) g0 = full - (unsigned)TILE;
#else
                const unsigned g0 = b4[c] + excl[c];                  // where the tile's keys of this digit go
#endif
                gb[c] = g0 - ds + (unsigned)TILE;                     // biased by TILE: base + position stays a plain 32-bit sum
                if constexpr (NEXT) {
                    unsigned r0 = 0;
#pragma unroll
                    for (unsigned q = 1; q < (unsigned)CHAINS; ++q) r0 += g0 >= q * span ? 1u : 0u;
                    if (g0 >= full) r0 = (unsigned)CHAINS;
                    const unsigned nb = r0 == (unsigned)CHAINS ? ~0u : ((r0 + 1u) * span < full ? (r0 + 1u) * span : full);   // first position of the following eighth
                    const unsigned r1 = nb >= full ? (unsigned)CHAINS : r0 + 1u;
                    const unsigned split = (nb - g0 < cnt) ? ds + (nb - g0) : 0xffffu;
                    nr[c] = split | (r0 << 16) | (r1 << 20);
                }
            }
            *reinterpret_cast<uint4v *>(&L.gbase[4 * lane]) = gb;
            if constexpr (NEXT) *reinterpret_cast<uint4v *>(&L.nrec[4 * lane]) = nr;
            if (nj != ~0u) fetch(t, ne * per + nj);
        }
        PSTAMP(5);
        lds_barrier();
        PSTAMP(6);

        if (declined) {
            // handed to the kernel behind this launch (radix_redo_kernel), which ranks the tile by match words and finds its bases in the chain
            if (t == 0) redo[atomicAdd(&status[SORT_STATUS_ORDER], 1u)] = tile;
        } else {
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const unsigned i = (unsigned)(t + k * UB);
                const K kk = s_keys[i];
                const K ko = to_ordered<K, MODE, DESC>(kk);
                const unsigned d = (unsigned)(ko >> shift) & (RADIX - 1);
                const unsigned eo = L.gbase[d] + (unsigned)t;
                if constexpr (WIDE) {
                    const unsigned g = eo + (unsigned)(k * UB) - (unsigned)TILE;
                    keys_out[(size_t)g] = kk;
                    if constexpr (VB != 0) vals_out[(size_t)g] = s_vals[i];
                } else {
                    store_elem<K>(kk, rk, eo * (unsigned)sizeof(K), (unsigned)(k * UB) * (unsigned)sizeof(K));
                    if constexpr (VB != 0) store_elem<VT>(s_vals[i], rv, eo * (unsigned)VB, (unsigned)(k * UB) * (unsigned)VB);
                }
                if constexpr (NEXT) {
                    const unsigned nr = L.nrec[d];
                    const unsigned row = (i >= (nr & 0xffffu) ? (nr >> 20) : (nr >> 16)) & 15u;
                    atomicAdd(&L.nexth[row][(unsigned)(ko >> ((shift + 8) & (8 * (int)sizeof(K) - 1))) & (RADIX - 1)], 1u);
                }
            }
        }
        PSTAMP(7);
        lds_barrier();                                            // (the next tile's counter words go where this tile lies)
        e = ne; j = nj;
    }
#ifdef VEXHIP_SORT_PROFILE
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) for (int i = 0; i < 12; ++i) redo[100000 + (blockIdx.x * (UB / kWave) + wave_s) * 12 + i] = (unsigned)(pt[i] >> 4);
#endif
#undef PSTAMP
    if constexpr (NEXT) {
        lds_barrier();
        for (int i = wave_s * kWave + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); i < ROWS * RADIX; i += UB) { const unsigned v = (&L.nexth[0][0])[i]; if (v) atomicAdd(&next[i], v); }
    }
}

// Behind every scatter of the unit kernel: the tiles it declined (a lane served out of order; tests: all of them), ranked by match words.
template <typename K, int MODE, bool DESC, int VB, int KPT, int TILE_KEYS>
__global__ __launch_bounds__(RB, 8)
void radix_redo_kernel(const K *__restrict__ keys_in, K *__restrict__ keys_out,
        const void *__restrict__ vals_in_, void *__restrict__ vals_out_,
        long long n, int shift, const base_src B, const unsigned *__restrict__ status, const unsigned *__restrict__ redo,
        unsigned *__restrict__ total)
{
    typedef typename valtype<VB>::type VT;
    __shared__ scatter_lds<K, VB, KPT> L;
    const unsigned count = status[SORT_STATUS_ORDER];
    for (unsigned i = blockIdx.x; i < count; i += gridDim.x) {
        scatter_tile<K, MODE, DESC, VB, KPT, TILE_KEYS, TILE_KEYS == RB * KPT>(L, redo[i], keys_in, keys_out,
                reinterpret_cast<const VT *>(vals_in_), reinterpret_cast<VT *>(vals_out_), n, shift, B);
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
int sort_passes(hipStream_t s, K *keys, K *keys_tmp, void *vals, void *vals_tmp, int64_t n, unsigned *tmp, unsigned *status, unsigned *redo, unsigned *chain, int rank, int cus) {
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
    const unsigned nfull_all = (unsigned)(n / TILE);
    if ((rank == 8 || rank == 9) && nfull_all >= 1) {
        // ---- chained passes (radix_onesweep_kernel): one read of the keys for the counts of pass 0, then 1 read + 1 write per pass
        const unsigned nfull = nfull_all, per = (nfull + CHAINS - 1) / CHAINS;
        chain += CH_PAD_ROWS * RADIX;                             // (rows in front of the first tile's: read, never used, by a look-back near a chain's start)
        unsigned *ctl = chain + (size_t)nfull * RADIX;           // [ counts of this pass | counts of the next | bases | tickets, owners ]
        unsigned *count[2] = {ctl, ctl + ROWS * RADIX}, *base = ctl + 2 * ROWS * RADIX, *tick = ctl + 3 * ROWS * RADIX;
        VEXHIP_TRY(hipMemsetAsync(count[0], 0, ROWS * RADIX * sizeof(unsigned), s));
        radix_count_kernel<K, MODE, DESC, TILE><<<std::min(nblocks, (unsigned)(8 * cus)), HB, 0, s>>>(src, n, 0, nblocks, nfull, per, count[0], vec_ok);
        VEXHIP_LAUNCH_CHECK();
        const unsigned grid = (unsigned)std::min<int64_t>(2 * (int64_t)cus, (int64_t)nfull);
        for (int p = 0; p < passes; ++p) {
            const int shift = 8 * p;
            const bool more = p + 1 < passes;
            unsigned *cur = count[p & 1], *nxt = count[(p + 1) & 1];
            radix_bases_kernel<<<1, RADIX, 0, s>>>(cur, base, nxt, tick);
            VEXHIP_LAUNCH_CHECK();
            VEXHIP_TRY(hipMemsetAsync(chain, 0, (size_t)nfull * RADIX * sizeof(unsigned), s));
#define SWEEP(W, NX, DIS) radix_onesweep_kernel<K, MODE, DESC, VB, UKPT, W, NX, DIS><<<grid, UB, 0, s>>>(src, dst, vsrc, vdst, shift, nfull, per, base, nxt, chain, tick, tick + CHAINS, status, redo)
            if (rank == 9) { if (more) SWEEP(true, true, true); else SWEEP(true, false, true); }
            else if (wide) { if (more) SWEEP(true, true, false); else SWEEP(true, false, false); }
            else           { if (more) SWEEP(false, true, false); else SWEEP(false, false, false); }
#undef SWEEP
            VEXHIP_LAUNCH_CHECK();
            const base_src B = {nullptr, 0u, base, chain, per, nfull, more ? nxt : nullptr, shift + 8, per * (unsigned)TILE, nfull * (unsigned)TILE};
            radix_redo_kernel<K, MODE, DESC, VB, SLOTS, TILE><<<std::min(nfull, 512u), RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, B, status, redo, status + SORT_STATUS_REDONE);
            VEXHIP_LAUNCH_CHECK();
            if (nfull < nblocks) {
                radix_scatter_kernel<K, MODE, DESC, VB, SLOTS, TILE, false><<<1, RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nfull, B);
                VEXHIP_LAUNCH_CHECK();
            }
            std::swap(src, dst);
            std::swap(vsrc, vdst);
        }
        return 0;
    }
    const base_src B = {table, nblocks, nullptr, nullptr, 0u, 0u, nullptr, 0, 1u, 0u};
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
            radix_redo_kernel<K, MODE, DESC, VB, SLOTS, TILE><<<std::min(nfull, 512u), RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, B, status, redo, status + SORT_STATUS_REDONE);
            VEXHIP_LAUNCH_CHECK();
        } else if (nfull) {
            // match words in every tile: complete tiles that fill the kernel's slots take the mask-free form
            if constexpr (fills) radix_scatter_kernel<K, MODE, DESC, VB, SLOTS, TILE, true><<<grid, RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nfull, B);
            else radix_scatter_kernel<K, MODE, DESC, VB, SLOTS, TILE, false><<<nfull, RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, 0u, B);
            VEXHIP_LAUNCH_CHECK();
        }
        if (nfull < nblocks) {
            // the ragged last tile: one workgroup, ranked by the match words
            radix_scatter_kernel<K, MODE, DESC, VB, SLOTS, TILE, false><<<1, RB, 0, s>>>(src, dst, vsrc, vdst, n, shift, nfull, B);
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
int sort_dispatch(hipStream_t s, int desc, int vb, void *keys, void *keys_tmp, void *vals, void *vals_tmp, int64_t n, void *tmp, unsigned *status, unsigned *redo, unsigned *chain, int rank, int cus) {
#define GO(DESC, VB) return sort_passes<K, MODE, DESC, VB>(s, (K *)keys, (K *)keys_tmp, vals, vals_tmp, n, (unsigned *)tmp, status, redo, chain, rank, cus)
    if (desc) { if (vb == 0) GO(true, 0); if (vb == 4) GO(true, 4); if (vb == 8) GO(true, 8); }
    else      { if (vb == 0) GO(false, 0); if (vb == 4) GO(false, 4); if (vb == 8) GO(false, 8); }
#undef GO
    return fail(__FILE__, __LINE__, "value_bytes must be 0, 4 or 8");
}

int g_sort_rank = -1;               // -1: the default (6: the unit scatter); 0: match words in every tile; 7: the unit scatter, every tile ranked again by ballots (tests)

// the status words of a sort live at the end of its workspace (vexhip_sort_tmp_bytes reserves them)
// [ table | scan workspace | chain words and control | redo list (one entry per tile) | status words ]
inline size_t sort_redo_elems(int64_t n) { const int64_t smallest_tile = UB * 4; return (size_t)((n + smallest_tile - 1) / smallest_tile + 4); }     // (8-byte key, 8-byte value): 3072 elements
// a chained sort's words: one per (tile, digit), then the counts of two passes, the bases, the tickets and the owners
inline size_t sort_chain_elems(int64_t n) { const int64_t smallest_tile = UB * 4; return (size_t)(((n + smallest_tile - 1) / smallest_tile + CH_PAD_ROWS) * RADIX) + 3 * ROWS * RADIX + 2 * CHAINS; }
inline size_t sort_tmp_elems(int64_t n) {
    const int64_t smallest_tile = UB * 4;
    const int64_t nblocks = (n + smallest_tile - 1) / smallest_tile;
    const int64_t tn = nblocks * RADIX;
    return (size_t)((tn + 3) / 4 * 4 + (int64_t)scan_tmp_elems_u32(tn) + 4) + sort_chain_elems(n) + sort_redo_elems(n) + SORT_STATUS_WORDS;
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
    unsigned *chain = redo - sort_chain_elems(n);
    const int cus = info(dev).cus > 0 ? info(dev).cus : 256;
    VEXHIP_TRY(hipMemsetAsync(status, 0, SORT_STATUS_WORDS * sizeof(unsigned), s));
    switch (key_dtype) {
        case VEXHIP_U32: return sort_dispatch<unsigned, KEY_UNSIGNED>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, status, redo, chain, ar, cus);
        case VEXHIP_I32: return sort_dispatch<unsigned, KEY_SIGNED>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, status, redo, chain, ar, cus);
        case VEXHIP_F32: return sort_dispatch<unsigned, KEY_FLOAT>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, status, redo, chain, ar, cus);
        case VEXHIP_U64: return sort_dispatch<unsigned long long, KEY_UNSIGNED>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, status, redo, chain, ar, cus);
        case VEXHIP_I64: return sort_dispatch<unsigned long long, KEY_SIGNED>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, status, redo, chain, ar, cus);
        case VEXHIP_F64: return sort_dispatch<unsigned long long, KEY_FLOAT>(s, descending, value_bytes, keys, keys_tmp, vals, vals_tmp, n, tmp, status, redo, chain, ar, cus);
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
    unsigned h[SORT_STATUS_WORDS] = {0, 0, 0, 0, 0, 0, 0, 0};
    VEXHIP_TRY(hipMemcpyAsync(h, static_cast<const unsigned *>(tmp) + sort_tmp_elems(n) - SORT_STATUS_WORDS, sizeof(h), hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    if (reranked_tiles) *reranked_tiles = h[SORT_STATUS_REDONE];
    if (dropped_tiles) *dropped_tiles = h[SORT_STATUS_TABLE];
    if (h[SORT_STATUS_CHAIN]) return fail(__FILE__, __LINE__, "vexhip_sort: a look-back along a chain of tiles gave up waiting for a predecessor's count: the result is not sorted");
    if (h[SORT_STATUS_TABLE]) return fail(__FILE__, __LINE__, "vexhip_sort: the keys changed while the sort ran (a pass's histogram and its scatter disagree): the result is not sorted");
    return 0;
}

} // extern "C"

VEXHIP_WARM_TU(sort)
