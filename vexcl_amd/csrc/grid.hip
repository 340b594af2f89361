// The GRID product (round 4): the plane product (plane.hip) for grids of ANY line length.
//
// y (=|+=) alpha * A * x for value-coded SELL-512 storage whose diagonals are those of a 7-point operator on an nx x ny x nz
// grid -- {0, +-1, +-nx, +-nx*ny} -- with nx, ny arbitrary (odd, not a divisor or multiple of 512: 384^3, 500^3, 127^3).
// Semantics are the reference's ELL product (/root/reference/vexcl/spmat/hybrid_ell.inl:238-269: the row's entries in storage
// order, products rounded before they are added, the scale applied to the sum), results bit-identical to the CSR loop
// (spmat/csr.inl:163-170).  The reference's own size-agnostic answer to stencil matrices is SpMatCCSR (spmat/ccsr.hpp:55-113:
// rows -> one of a few (offset, value) stencils); this is what the plan below recovers from the stored matrix.
//
// Why the plane product does not cover these grids: it reads the matrix from the slice dictionary of the SELL-512 storage -- a
// slice IS a grid line when nx = 512.  For other nx the lines drift through the slices: 384^3 still has a dictionary, 500^3
// has > 128 distinct slices and streams 14 bytes of codes per row next to the 16 bytes of x and y.  The grid does not care:
//   * the plan re-expresses the matrix by GRID LINE.  A device pass turns every row's codes into seven bytes -- the value code
//     at each of the positions {-P, -nx, -1, 0, +1, +nx, +P}, 255 where the row has no entry -- checks that every row keeps
//     its entries in ascending position (position order = storage order) and hashes each line's rows.  Lines with equal rows
//     form a CLASS (a handful: interior lines, boundary lines, lines of boundary planes); the matrix becomes 4 bytes per line
//     (its class) + a table of 7 * nx bytes per class, and a second pass verifies every line against its class byte by byte.
//   * the kernel is the plane kernel with the line length as a number: a workgroup owns TWO adjacent grid lines (one SEGMENT of
//     <= 512 rows of them when nx > 512) and walks through the planes; lane t owns rows 2t, 2t + 1 of its segment in every
//     plane, so the +-nx and +-P neighbours are pairs the same lane loaded (registers), the +-1 neighbours one DPP wave shift
//     (the element beyond either end of a wave's 128 rows: a scalar-like 8-byte load).  All x addressing is LINEAR (line *
//     nx + row), so lines need not be 16-byte aligned (odd nx: 16-byte requests at 8-byte addresses, which gfx950 serves),
//     the last tile of a plane with an odd number of lines simply does not store its second line, and a lane beyond the end
//     of its line computes on whatever lies there and stores nothing (the fast loop runs only where such a lane's requests
//     still lie inside the arrays; the last planes take clamped, element-wise requests).
// The 512-point kernel stays the default where it applies (the headline): same walk, fewer scalar operands.
// Compiled with -ffp-contract=off.
#include "common.hpp"
#include "halo.hpp"
#include "lanes.hpp"
#include "grid.hpp"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace vexhip {
namespace {

constexpr unsigned GR_PAD_FIRST = 254;      // codes 254 / 255 are padding (sell8.hip)
constexpr int GR_ABSENT = 255;              // table byte of a position without an entry (value codes are < 255)

struct codes_dev {           // where the value-coded slices are (sell8.hip: ceil(w/2) KiB diagonal codes, then as many value codes)
    const char *buf; const int *blocks; int w, ndeltas;
    signed char pos[8];      // diagonal code -> position 0..6
};


// ---- set-up: a row's codes -> seven bytes (value code per position, 255 = no entry); false: not a matrix for this product ----
__device__ __forceinline__ bool row_signature(const codes_dev &cd, long long i, unsigned char (&sig)[7]) {
    const long long s = i >> 9;
    const int t2 = (int)(i & 511) >> 1, q = (int)(i & 1);
    const int wp = (cd.w + 1) >> 1;
    const long long sb = cd.blocks ? (long long)cd.blocks[s] : s;
    const unsigned *cw = reinterpret_cast<const unsigned *>(cd.buf + sb * ((long long)wp * 2048)) + t2;
    const unsigned *vw = cw + wp * 256;
#pragma unroll
    for (int p = 0; p < 7; ++p) sig[p] = GR_ABSENT;
    int last = -1; bool ok = true;
    for (int j = 0; j < cd.w; ++j) {
        const int sh = 8 * ((j & 1) * 2 + q);
        const unsigned code = (cw[(j >> 1) * 256] >> sh) & 255u;
        if (code >= GR_PAD_FIRST) continue;
        if ((int)code >= cd.ndeltas) { ok = false; continue; }
        const int p = cd.pos[code];
        const unsigned vc = (vw[(j >> 1) * 256] >> sh) & 255u;
        if (p <= last || vc >= (unsigned)GR_ABSENT) { ok = false; continue; }
        last = p;
#pragma unroll
        for (int k = 0; k < 7; ++k) if (k == p) sig[k] = (unsigned char)vc;
    }
    return ok;
}
__device__ __forceinline__ unsigned long long mix64(unsigned long long v) {           // splitmix64 finaliser
    v ^= v >> 30; v *= 0xbf58476d1ce4e5b9ull; v ^= v >> 27; v *= 0x94d049bb133111ebull; v ^= v >> 31;
    return v;
}
// one wave per grid line: hash[line] = sum over its rows of mix(signature, row) (commutative: any lane order)
__global__ __launch_bounds__(256)
void grid_line_hash_kernel(codes_dev cd, long long lines, int nx, unsigned long long *__restrict__ hash, int *__restrict__ bad)
{
    const long long line = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= lines) return;
    const int lane = threadIdx.x & 63;
    unsigned long long h = 0; bool ok = true;
    for (int r = lane; r < nx; r += 64) {
        unsigned char sig[7];
        ok = row_signature(cd, line * nx + r, sig) && ok;
        unsigned long long v = 0;
#pragma unroll
        for (int p = 0; p < 7; ++p) v |= (unsigned long long)sig[p] << (8 * p);
        h += mix64(v + 0x9e3779b97f4a7c15ull * (unsigned long long)(r + 1));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
    if (lane == 0) hash[line] = h;
    if (!ok) atomicOr(bad, 1);
}
// table[c][p][r] of class c from its representative line (rows beyond nx: 255)
__global__ __launch_bounds__(256)
void grid_line_table_kernel(codes_dev cd, const long long *__restrict__ rep, int nx, int pitch, unsigned char *__restrict__ table)
{
    const int c = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= pitch) return;
    unsigned char sig[7];
#pragma unroll
    for (int p = 0; p < 7; ++p) sig[p] = GR_ABSENT;
    if (r < nx) (void)row_signature(cd, rep[c] * nx + r, sig);
#pragma unroll
    for (int p = 0; p < 7; ++p) table[((long long)c * 7 + p) * pitch + r] = sig[p];
}
// every row of every line against the table of the line's class
__global__ __launch_bounds__(256)
void grid_line_verify_kernel(codes_dev cd, long long lines, int nx, int pitch, const int *__restrict__ line_class,
        const unsigned char *__restrict__ table, int *__restrict__ bad)
{
    const long long line = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= lines) return;
    const int lane = threadIdx.x & 63;
    const unsigned char *tb = table + (long long)line_class[line] * 7 * pitch;
    bool ok = true;
    for (int r = lane; r < nx; r += 64) {
        unsigned char sig[7];
        ok = row_signature(cd, line * nx + r, sig) && ok;
#pragma unroll
        for (int p = 0; p < 7; ++p) ok = ok && sig[p] == tb[(long long)p * pitch + r];
    }
    if (!ok) atomicOr(bad, 2);
}


// ---- the matrix by grid line straight from the CSR arrays (round 4: the set-up of grid matrices in ONE pass over them) ----
// A probe of a few thousand rows in the middle of the matrix names the diagonals; if they are {0, +-1, +-nx, +-P}, one kernel
// reads the CSR arrays once and leaves what the grid / plane products read.  A workgroup takes one grid line at a time, 512 rows
// of it at a time: the rows' entries are staged in LDS with coalesced loads -- a value becomes its code there: an LDS hash per workgroup in front of a device-wide
// table that numbers the distinct values in the order they are first met -- every lane turns its two rows into seven bytes
// each (code per position, 255 = no entry; the entries must ascend by position), the line's rows are hashed and the line is
// looked up in a device-wide table of classes: the first workgroup to bring a hash writes the class table and publishes its
// number, every later one compares its rows with that table byte by byte (a different line with the same hash ends the build:
// the classic set-up takes over, as it does for rows that do not fit -- a diagonal outside the set, descending positions, more
// than 254 values, more than 128 classes).  No per-slice codes are written (2.1 GB at 512^3), no second pass over the arrays
// (the analysis of the classic set-up reads them once to find the tables, the fill a second time), nothing but a few counters
// and the value table returns to the host.
#ifndef VEXHIP_GB_PIECE
#define VEXHIP_GB_PIECE 512
#endif
constexpr int GB_PIECE = VEXHIP_GB_PIECE;     // rows staged at a time
constexpr int GB_CAP = 8 * GB_PIECE;          // entries of a piece (rows with more than 8 entries have no place in this storage)
constexpr int GB_CHUNK = GB_CAP / 256;        // entries a lane requests together: a whole piece in one round (16)
constexpr int GB_VSLOTS = 512;                // hash of the values: LDS per workgroup, and the device-wide table behind it
constexpr int GB_KEYS = 1024;                 // device-wide table of line hashes
constexpr int GB_KNOWN = 128;                 // classes a workgroup remembers
constexpr int GB_MAX_CLASSES = 128;
constexpr int GB_MAX_VALUES = 255;            // codes 0 .. 254 (255 = no entry)
enum { GB_BAD_ROW = 1, GB_OVERFLOW = 2, GB_COLLISION = 4, GB_TIMEOUT = 8 };
enum { GBI_COUNT = 0, GBI_FLAGS, GBI_VCOUNT, GBI_POSMASK, GBI_MAXCOL, GBI_MAXLEN, GBI_USES = 8, GBI_INTS = 8 + GB_MAX_CLASSES };

struct gb_dev {
    long long n, lines, far;
    int nx, pitch, pieces;
    unsigned long long *keys; int *ids;                              // classes: hash of a line -> number
    unsigned long long *vkeys; int *vstate; int *vcodes;             // values: bits -> code (vstate 0 empty, 1 being written, 2 ready)
    int *ints;                                                       // GBI_*
    unsigned char *table; int *line_class;
};

// (two 32-bit multiplies: the 64-bit golden-ratio multiply was a dozen instructions per entry in a kernel bound by instruction issue)
__device__ __forceinline__ unsigned gb_vhash(unsigned long long bits) { return ((((unsigned)bits ^ ((unsigned)(bits >> 32) * 0x9e3779b9u)) * 0x85ebca6bu) >> 23) & (GB_VSLOTS - 1); }

// the device-wide code of a value: found, or numbered now (-1: the table is full / a writer never finished)
__device__ int gb_global_value_code(unsigned long long bits, const gb_dev &g)
{
    // the build is over (a flag is set) or the table is full: no walk through 512 device-wide slots per value -- a matrix with a
    // coefficient per face kept every wave of the kernel in such walks for seconds before this check (5.7 s at 512^3)
    if (__hip_atomic_load(&g.ints[GBI_FLAGS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return -1;
    if (__hip_atomic_load(&g.ints[GBI_VCOUNT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= GB_MAX_VALUES) { atomicOr(&g.ints[GBI_FLAGS], GB_OVERFLOW); return -1; }
    unsigned h = gb_vhash(bits);
    for (int probe = 0; probe < GB_VSLOTS; ++probe) {
        int st = __hip_atomic_load(&g.vstate[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (st == 0) {
            st = atomicCAS(&g.vstate[h], 0, 1);
            if (st == 0) {                                           // this slot is ours: number the value, publish
                int c = atomicAdd(&g.ints[GBI_VCOUNT], 1);
                if (c >= GB_MAX_VALUES) { atomicOr(&g.ints[GBI_FLAGS], GB_OVERFLOW); c = -1; }
                __hip_atomic_store(&g.vkeys[h], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&g.vcodes[h], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&g.vstate[h], 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                return c;
            }
        }
        for (int spin = 0; st != 2 && spin < (1 << 18); ++spin) {    // the writer is running: a few hundred cycles
            __builtin_amdgcn_s_sleep(2);
            st = __hip_atomic_load(&g.vstate[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (st != 2) { atomicOr(&g.ints[GBI_FLAGS], GB_TIMEOUT); return -1; }
        if (__hip_atomic_load(&g.vkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == bits)
            return __hip_atomic_load(&g.vcodes[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        h = (h + 1) & (GB_VSLOTS - 1);
    }
    atomicOr(&g.ints[GBI_FLAGS], GB_OVERFLOW);
    return -1;
}

// V: the matrix values (double / float; a float is coded by the bits of the double it converts to -- exactly, both ways)
template <typename P, typename V>
__global__ __launch_bounds__(256, 4)                // four workgroups per CU (the LDS allows no fifth): 128 registers
void grid_build_kernel(const P *__restrict__ ptr, const int *__restrict__ col, const V *__restrict__ val, gb_dev g)
{
    extern __shared__ unsigned char gb_lds[];
    // [value hash keys 512 x 8][value hash codes 512][row bounds 520 x 8][staged columns GB_CAP x 4][staged value codes GB_CAP][line 7 x pitch][scratch]
    unsigned long long *s_vkey = reinterpret_cast<unsigned long long *>(gb_lds);
    unsigned char *s_vcode = gb_lds + GB_VSLOTS * 8;
    long long *s_ptr = reinterpret_cast<long long *>(s_vcode + GB_VSLOTS);
    int *s_c = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(s_ptr) + 520 * 8);
    unsigned char *s_vc = reinterpret_cast<unsigned char *>(s_c + GB_CAP);
    unsigned char *s_sig = s_vc + GB_CAP;
    unsigned long long *s_red = reinterpret_cast<unsigned long long *>(s_sig + ((7 * g.pitch + 15) / 16) * 16);     // [0..3] wave sums, [4] id, [5] winner, [6] slot, [8..11] bad rows
    __shared__ unsigned long long s_kkey[GB_KNOWN];
    __shared__ int s_kid[GB_KNOWN];
    __shared__ int s_uses[GB_MAX_CLASSES];
    int known = 0;                                       // (lane 0's)
    const int t = threadIdx.x, lane = t & 63;
    if (t < GB_MAX_CLASSES) s_uses[t] = 0;
    // the workgroup's view of the value table: bits -> code (all ones = empty slot: a value with those bits -- a NaN -- declines)
    for (int i = t; i < GB_VSLOTS; i += 256) { s_vkey[i] = ~0ull; s_vcode[i] = 255; }
    __syncthreads();

    unsigned posmask = 0; int maxcol = -1, maxlen = 0;
    long long line = blockIdx.x; int pc = 0;
    unsigned long long hsum = 0;
    bool bad = false;

    // Round 5: the pass is bound by the LATENCY of its dependent steps per line (row bounds -> entries -> codes -> rows -> class ->
    // table), four workgroups per CU to hide it (2 / 3 / 4 per CU: 6.3 / 4.6 / 4.0 ms at 512^3; the float arrays, a third fewer
    // bytes, take no less).  So the entries of the NEXT piece are requested as soon as the staging has emptied the registers of
    // this one -- all of a piece in one round, GB_CHUNK = 16 entries per lane -- and its row bounds one step earlier; they arrive
    // while the rows of this piece are coded and the line is classed.  The barriers in between order LDS traffic only (gb_barrier):
    // a __syncthreads() waits for every outstanding load of the wave.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "grid_build_kernel orders its LDS traffic with `s_waitcnt lgkmcnt(0)` + `s_barrier`: gfx9-family targets only (ARCH in csrc/Makefile)"
#endif
    auto gb_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto load_bounds = [&](long long ln, int p, long long &e0_, long long &cnt64_, long long (&mp)[3]) {
        const int r0_ = p * GB_PIECE;
        const int rows_ = g.nx - r0_ < GB_PIECE ? g.nx - r0_ : GB_PIECE;
        const long long at = ln * g.nx + r0_;
        e0_ = (long long)ptr[at];
        cnt64_ = (long long)ptr[at + rows_] - e0_;
#pragma unroll
        for (int u = 0; u < 3; ++u) { const int i = t + 256 * u; mp[u] = i <= rows_ ? (long long)ptr[at + i] : 0; }
    };
    int n_c[GB_CHUNK]; V n_v[GB_CHUNK];
    // (buffer requests: the piece's first entry is the base -- scalar --, the lane's offset ONE register for all sixteen requests,
    //  the range check gives the lanes beyond the piece's last entry zeros.  Flat addressing took two registers per request: 160
    //  in all, one workgroup per CU fewer.)
    auto load_entries = [&](long long e0_, int cnt_) {
        const long long eu = ((long long)__builtin_amdgcn_readfirstlane((int)(e0_ >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)e0_);
        const int cu = __builtin_amdgcn_readfirstlane(cnt_ > 0 ? cnt_ : 0);
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(col + eu), 0, cu * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<V *>(val + eu), 0, cu * (int)sizeof(V), 0x00020000);
#pragma unroll
        for (int u = 0; u < GB_CHUNK; ++u) {
            n_c[u] = __builtin_amdgcn_raw_buffer_load_b32(rc, 4 * t, 1024 * u, 0);
            if constexpr (sizeof(V) == 8) n_v[u] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b64(rv, 8 * t, 2048 * u, 0));
            else n_v[u] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b32(rv, 4 * t, 1024 * u, 0));
        }
    };
    long long e0 = 0, cnt64 = 0, my_ptr[3] = {0, 0, 0};
    int cnt = 0;
    if (line < g.lines) {
        load_bounds(line, pc, e0, cnt64, my_ptr);
        cnt = (cnt64 < 0 || cnt64 > GB_CAP) ? -1 : (int)cnt64;
        load_entries(e0, cnt);
    }
    while (line < g.lines) {
        const int r0 = pc * GB_PIECE;
        const int rows = g.nx - r0 < GB_PIECE ? g.nx - r0 : GB_PIECE;
        const long long row_l = line * g.nx;
        long long nline = line; int npc = pc + 1;
        if (npc == g.pieces) { nline = line + gridDim.x; npc = 0; }
        // the NEXT piece's bounds (uniform) and its rows' (one to three per lane): requested now, used behind the staging
        long long e0n = 0, cnt64n = 0, my_ptr_n[3] = {0, 0, 0};
        const bool more = nline < g.lines;
        if (more) load_bounds(nline, npc, e0n, cnt64n, my_ptr_n);

        // another workgroup gave up?  ONE lane looks and the workgroup leaves together (every lane loading the flag for itself could
        // split the workgroup: lanes that leave while the others wait at the barrier below and go on with a stale s_red)
        if (pc == 0 && t == 0) s_red[7] = __hip_atomic_load(&g.ints[GBI_FLAGS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 1ull : 0ull;
        gb_barrier();                                           // the previous trip is done with the staged entries and with the line's rows (compared / copied)
        if (pc == 0 && s_red[7]) return;                        // uniform
        if (pc == 0) {
            for (int i = t; i < 7 * g.pitch; i += 256) s_sig[i] = GR_ABSENT;       // (the rows below write into it behind the next barrier)
            hsum = 0; bad = false;
        }
        if (cnt < 0) bad = true;                                // uniform
#pragma unroll
        for (int u = 0; u < 3; ++u) { const int i = t + 256 * u; if (i <= rows) s_ptr[i] = my_ptr[u]; }
        // the piece's entries (requested one trip ago): coded and staged.  The first probe of the workgroup's value hash almost
        // always decides: that much is straight-line code for all sixteen entries (an empty slot holds code 255, so does a slot with
        // another value's bits); whatever is still uncoded -- a collision in the hash, a value this workgroup has not met, the
        // bits that mean "empty" -- is settled behind ONE vote per piece.
        bool open_ = false;
#pragma unroll
        for (int u = 0; u < GB_CHUNK; ++u) {
            const int k = t + 256 * u;
            const unsigned long long bits = (unsigned long long)__double_as_longlong((double)n_v[u]);
            const unsigned h = gb_vhash(bits);
            const unsigned code = s_vkey[h] == bits ? (unsigned)s_vcode[h] : 255u;
            if (k < cnt) {
                s_c[k] = n_c[u]; s_vc[k] = (unsigned char)code;      // (255: settled below, the lane re-reads its own byte)
                open_ = open_ || code == 255u;
                maxcol = n_c[u] > maxcol ? n_c[u] : maxcol;
            }
        }
        if (__ballot(open_)) {                                  // (per wave)
#pragma unroll 1
            for (int u = 0; u < GB_CHUNK; ++u) {
                const int k = t + 256 * u;
                const bool active = k < cnt;
                // (dynamic index into the register array: a select chain, this path runs a few times per workgroup)
                unsigned long long bits = 0;
#pragma unroll
                for (int v = 0; v < GB_CHUNK; ++v) if (v == u) bits = (unsigned long long)__double_as_longlong((double)n_v[v]);
                unsigned cd = active ? (unsigned)s_vc[k] : 0u;
                if (active && cd == 255u) {
                    if (bits == ~0ull) { bad = true; cd = 254; }
                    else {
                        unsigned h = gb_vhash(bits);
                        for (int probe = 0; probe < GB_VSLOTS; ++probe) {
                            const unsigned long long key = s_vkey[h];
                            if (key == bits) { cd = s_vcode[h]; break; }
                            if (key == ~0ull) break;
                            h = (h + 1) & (GB_VSLOTS - 1);
                        }
                    }
                }
                // values this workgroup has not met: one look-up in the device-wide table per distinct value and wave
                unsigned long long miss = __ballot(active && cd == 255u);
                while (miss) {
                    const int leader = __ffsll((long long)miss) - 1;
                    const unsigned long long lb = __shfl(bits, leader, 64);
                    int cglob = 0;
                    if (lane == leader) {
                        cglob = gb_global_value_code(lb, g);
                        if (cglob >= 0) {                           // into the workgroup's hash
                            unsigned h = gb_vhash(lb);
                            for (int probe = 0; probe < GB_VSLOTS; ++probe) {
                                const unsigned long long old = atomicCAS(&s_vkey[h], ~0ull, lb);
                                if (old == ~0ull || old == lb) { s_vcode[h] = (unsigned char)cglob; break; }
                                h = (h + 1) & (GB_VSLOTS - 1);
                            }
                        }
                    }
                    cglob = __shfl(cglob, leader, 64);
                    if (cglob < 0) { if (active && cd == 255u) { bad = true; cd = 254; } }              // the build is over: no further look-ups
                    else if (active && cd == 255u && bits == lb) cd = (unsigned)cglob;
                    miss = __ballot(active && cd == 255u);
                }
                if (active) s_vc[k] = (unsigned char)cd;
            }
        }
        // the next piece's entries, into the registers just emptied
        const int cntn = (cnt64n < 0 || cnt64n > GB_CAP) ? -1 : (int)cnt64n;
        if (more) load_entries(e0n, cntn);
        gb_barrier();
        if (cnt >= 0) {
            // the lane's two rows, straight-line (selects instead of branches: per-lane trip counts cost a scalar instruction per
            // vector one); the eighth trip only where a row of the wave has eight entries (a 7-point operator never does)
            int rb[2], rl[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = 2 * t + q;
                const bool in = r < rows;
                const int b = in ? (int)(s_ptr[r] - e0) : 0, e = in ? (int)(s_ptr[r + 1] - e0) : 0;
                const bool fits = !(e - b > 8 || e < b || b < 0 || e > cnt);
                if (!fits) bad = true;
                rb[q] = fits ? b : 0; rl[q] = fits ? e - b : 0;
                maxlen = rl[q] > maxlen ? rl[q] : maxlen;
            }
            const bool eighth = __ballot(rl[0] > 7 || rl[1] > 7) != 0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = 2 * t + q;
                const int i = (int)(row_l + r0 + r);                                   // (rows are below 2^31)
                unsigned long long sig = 0x00ffffffffffffffull;           // seven bytes of 255
                int last = -1;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (u == 7 && !eighth) break;                        // uniform
                    const bool valid = u < rl[q];
                    const int j = valid ? rb[q] + u : 0;
                    const int d = s_c[j] - i;                            // (a difference that wraps is no diagonal of the set)
                    const unsigned vc = s_vc[j];
                    const int p = d == 0 ? 3 : d == -1 ? 2 : d == 1 ? 4 : d == -g.nx ? 1 : d == g.nx ? 5 : d == -(int)g.far ? 0 : d == (int)g.far ? 6 : -1;
                    const bool ok = valid && p > last && vc < 254u;
                    if (valid && !ok) bad = true;
                    last = ok ? p : last;
                    posmask |= ok ? 1u << (p & 7) : 0u;
                    sig ^= ok ? (unsigned long long)(255u ^ vc) << (8 * (p & 7)) : 0ull;          // (the byte at p still holds 255: positions ascend)
                }
                if (r < rows) {
#pragma unroll
                    for (int p = 0; p < 7; ++p) s_sig[p * g.pitch + r0 + r] = (unsigned char)(sig >> (8 * p));
                    hsum += mix64(sig + 0x9e3779b97f4a7c15ull * (unsigned long long)(r0 + r + 1));
                }
            }
        }
        if (pc + 1 < g.pieces) { e0 = e0n; cnt64 = cnt64n; cnt = cntn; my_ptr[0] = my_ptr_n[0]; my_ptr[1] = my_ptr_n[1]; my_ptr[2] = my_ptr_n[2]; line = nline; pc = npc; continue; }        // (same line, next piece)

        // ---- the line is complete: its hash, whether any lane met a row that does not fit, its class ----
        unsigned long long hs = hsum;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hs += __shfl_xor(hs, o, 64);
        const unsigned long long anybad = __ballot(bad);
        gb_barrier();
        if (lane == 0) { s_red[t >> 6] = hs; s_red[8 + (t >> 6)] = anybad ? 1 : 0; }
        gb_barrier();
        if (t == 0) {
            int id = -2; int winner = 0; unsigned slot = 0;
            int asked = 0;                                      // the device-wide table was consulted (first meeting of a class: a handful per workgroup)
            if (s_red[8] | s_red[9] | s_red[10] | s_red[11]) atomicOr(&g.ints[GBI_FLAGS], GB_BAD_ROW);
            else {
                const unsigned long long key = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) | 1ull;      // never 0 (= empty)
                // the classes this workgroup has met before (a handful): no device-wide traffic for them
                for (int k = 0; k < known; ++k) if (s_kkey[k] == key) { id = s_kid[k]; break; }
                if (id < 0) {
                    asked = 1;
                    slot = (unsigned)(key >> 20) & (GB_KEYS - 1);
                    for (int probe = 0; probe < GB_KEYS; ++probe) {
                        unsigned long long old = __hip_atomic_load(&g.keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (old == 0ull) old = atomicCAS(&g.keys[slot], 0ull, key);
                        if (old == 0ull) {                                 // a new class: number it, write its table, publish
                            id = atomicAdd(&g.ints[GBI_COUNT], 1);
                            if (id >= GB_MAX_CLASSES) { atomicOr(&g.ints[GBI_FLAGS], GB_OVERFLOW); __hip_atomic_store(&g.ids[slot], -2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); id = -2; }
                            else winner = 1;
                            break;
                        }
                        if (old == key) {                                  // known: wait for its number (the writer is running)
                            int v = -1;
                            for (int spin = 0; spin < (1 << 18); ++spin) {
                                v = __hip_atomic_load(&g.ids[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                                if (v != -1) break;
                                __builtin_amdgcn_s_sleep(4);
                            }
                            if (v == -1) atomicOr(&g.ints[GBI_FLAGS], GB_TIMEOUT);
                            id = v < 0 ? -2 : v;
                            break;
                        }
                        slot = (slot + 1) & (GB_KEYS - 1);
                    }
                    if (id >= 0 && known < GB_KNOWN) { s_kkey[known] = key; s_kid[known] = id; ++known; }
                }
            }
            s_red[4] = (unsigned long long)(long long)id; s_red[5] = (unsigned long long)winner; s_red[6] = slot; s_red[12] = (unsigned long long)asked;
        }
        gb_barrier();
        // lane 0 has acquired a class number at agent scope: every wave orders its own loads of the class table behind that (rare)
        if (s_red[12]) __syncthreads();                                                 // uniform
        const int id = (int)(long long)s_red[4];
        const bool winner = s_red[5] != 0;
        if (id < 0) return;                                                              // the build is over (a flag is set); uniform
        unsigned *tb = reinterpret_cast<unsigned *>(g.table + (long long)id * 7 * g.pitch);
        const unsigned *mine = reinterpret_cast<const unsigned *>(s_sig);
        const int words = 7 * g.pitch / 4;                                              // pitch is a multiple of 16
        if (winner) {
            for (int i = t; i < words; i += 256) __hip_atomic_store(tb + i, mine[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();                                    // every lane's part of the table is visible device-wide ...
            __syncthreads();
            if (t == 0) __hip_atomic_store(&g.ids[(unsigned)s_red[6]], id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);    // ... before its number is
        } else {
            // (plain loads: lane 0's acquire of the class number -- when this workgroup first met the class -- has invalidated what this
            //  CU and its L2 held; the table does not change after it is published.  All loads of a trip are issued before the
            //  first comparison.)
            unsigned diff = 0;
            int i = t;
            for (; i + 3 * 256 < words; i += 4 * 256) {
                const unsigned a0 = tb[i], a1 = tb[i + 256], a2 = tb[i + 512], a3 = tb[i + 768];
                diff |= (a0 ^ mine[i]) | (a1 ^ mine[i + 256]) | (a2 ^ mine[i + 512]) | (a3 ^ mine[i + 768]);
            }
            for (; i < words; i += 256) diff |= tb[i] ^ mine[i];
            if (__ballot(diff != 0) && lane == 0) atomicOr(&g.ints[GBI_FLAGS], GB_COLLISION);
        }
        if (t == 0) { g.line_class[line] = id; ++s_uses[id]; }
        e0 = e0n; cnt64 = cnt64n; cnt = cntn; my_ptr[0] = my_ptr_n[0]; my_ptr[1] = my_ptr_n[1]; my_ptr[2] = my_ptr_n[2]; line = nline; pc = npc;
    }
    __syncthreads();
    for (int i = t; i < GB_MAX_CLASSES; i += 256) if (s_uses[i]) atomicAdd(&g.ints[GBI_USES + i], s_uses[i]);
    // what the product and the storage selection need to know about the matrix as a whole
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { posmask |= (unsigned)__shfl_xor((int)posmask, o, 64); const int mc = __shfl_xor(maxcol, o, 64), ml = __shfl_xor(maxlen, o, 64); maxcol = mc > maxcol ? mc : maxcol; maxlen = ml > maxlen ? ml : maxlen; }
    if (lane == 0) { atomicOr(&g.ints[GBI_POSMASK], (int)posmask); atomicMax(&g.ints[GBI_MAXCOL], maxcol); atomicMax(&g.ints[GBI_MAXLEN], maxlen); }
}

// the diagonals of a few thousand rows (the probe in front of the one-pass build): a set of at most 15, overflow flag in [15]
template <typename P, typename V>
__global__ __launch_bounds__(256)
void grid_probe_kernel(const P *__restrict__ ptr, const int *__restrict__ col, const V *__restrict__ val, long long first, long long rows,
        int *__restrict__ set /* 16, INT_MIN = empty */, unsigned long long *__restrict__ vset /* 512 slots, all ones = empty */, int *__restrict__ vinfo /* [0] distinct values, [1] more than 254 */)
{
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
        const long long i = first + r;
        const long long b = (long long)ptr[i], e = (long long)ptr[i + 1];
        // the values of these rows: a matrix with more than 254 of them in 8192 rows (a coefficient per face) is not one for this storage
        for (long long j = b; j < e && j < b + 16; ++j) {
            if (__hip_atomic_load(&vinfo[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            const unsigned long long bits = (unsigned long long)__double_as_longlong((double)val[j]);
            unsigned h = gb_vhash(bits);
            for (int probe = 0; probe < GB_VSLOTS; ++probe) {
                unsigned long long old = __hip_atomic_load(&vset[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == ~0ull) {
                    old = atomicCAS(&vset[h], ~0ull, bits);
                    if (old == ~0ull && atomicAdd(&vinfo[0], 1) >= GB_MAX_VALUES) atomicExch(&vinfo[1], 1);
                }
                if (old == ~0ull || old == bits) break;
                h = (h + 1) & (GB_VSLOTS - 1);
            }
        }
        for (long long j = b; j < e && j < b + 16; ++j) {
            const long long d64 = (long long)col[j] - i;
            const int d = (int)d64;
            bool placed = false;
            for (int k = 0; k < 15 && !placed; ++k) {
                int old = __hip_atomic_load(&set[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (int)0x80000000) old = atomicCAS(&set[k], (int)0x80000000, d);
                placed = old == (int)0x80000000 || old == d;
            }
            if (!placed || d64 != (long long)d) atomicExch(&set[15], 1);
        }
    }
}

// ---- the product ----
// MAXT: 256 lanes (segments of <= 512 rows, four workgroups per CU) or 512 (segments of <= 1024 rows, two per CU: lines of 513 ..
// 1024 points stay in one piece -- two 320-row segments of a 640-point line read x 1.87 times, profiles/r04_grid_pmc.json)
// HALO (round 6; halo.hpp, the PULL form only): the launch is one device's whole product step.  x and y are addressed in the numbering of
// the STORED grid, whose plane z0 - 1 / z1 (where the device has a neighbour) is a ghost plane: read from the neighbour's x in place
// (H.lo / H.hi), behind its "x is final" flag where flags order the launches.  The walks are the plan's; the two that touch a ghost
// plane are dispatched last and wait before they start.
// ZM: the addend of the result (plane.hip): 0 none, 1 beta times the array zs ('+=': zs = y, beta = 1), 2 beta times x itself (from the registers)
template <int ZM, int STORE_AUX, int MAXT, bool HALO>
__device__ __forceinline__
void grid_walk(const double *__restrict__ x, double *__restrict__ y, double alpha, const double *__restrict__ zs, double beta,
        const int *__restrict__ line_class, const unsigned char *__restrict__ table, const double *__restrict__ values, const grid_dev &gd,
        const halo_dev &H, [[maybe_unused]] const unsigned long long step)
{
    constexpr int TY = 2;
    // LDS: the value table and the decoded values of the OTHER class, lane-private ([position * 2 + row][lane]: conflict-free)
    __shared__ double s_value[256];
    __shared__ double s_other[14][MAXT];

    const int t = threadIdx.x;
    const unsigned b = blockIdx.x, xcd = b & 7u, q = b >> 3;
    int zc = (int)(q / (unsigned)gd.tpx);
    int tile = (int)xcd * gd.tpx + (int)(q - (unsigned)zc * (unsigned)gd.tpx);
    if (gd.cpx) { const int w = (int)(q / (unsigned)gd.tiles); tile = (int)(q - (unsigned)w * (unsigned)gd.tiles); zc = (int)xcd * gd.cpx + w; }      // (grid.hpp)
    if (tile >= gd.tiles) return;                                   // the whole workgroup
    const int ytile = tile / gd.segs, seg = tile - ytile * gd.segs;
    const int y0 = TY * ytile;
    const int nl = gd.ny - y0 < TY ? gd.ny - y0 : TY;               // lines of the tile inside a plane (odd ny: the last tile has one)
    const int row0 = seg * gd.seg_len;
    const int len = gd.nx - row0 < gd.seg_len ? gd.nx - row0 : gd.seg_len;
    int z, zend_;
    [[maybe_unused]] bool ghost_bad = false;
    [[maybe_unused]] __shared__ int s_flag;
    if constexpr (HALO) {
        // the walks of the planes [z0, z1), the first and the last of them -- which touch a ghost plane -- dispatched behind the others
        const int nch = (H.z1 - H.z0 + gd.depth - 1) / gd.depth;
        const int c = nch > 1 ? (zc + 1) % nch : 0;
        z = H.z0 + c * gd.depth; zend_ = z + gd.depth < H.z1 ? z + gd.depth : H.z1;
        if (zc >= nch || z >= zend_) return;
        ghost_bad = !halo_wait(H, step, H.lo && z == H.z0, H.hi && zend_ == H.z1, &s_flag);
    } else {
        z = zc * gd.depth;
        zend_ = z + gd.depth < gd.nz ? z + gd.depth : gd.nz;
        if (z >= zend_) return;
    }
    const int zend = zend_;
    const int nx = gd.nx, ny = gd.ny;
    // HALO: the elements of x and y that exist are those of the planes [z0, z1)
    const long long own_lo = HALO ? (long long)H.z0 * ny * nx : 0;
    const long long lines = HALO ? (long long)H.z1 * ny : gd.lines, x_last = HALO ? (long long)H.z1 * ny * nx - 1 : gd.x_last, n = HALO ? (long long)H.z1 * ny * nx : gd.n;
    const unsigned lane_b = 16u * (unsigned)t;
    const bool full = 2 * t + 1 < len, half = 2 * t + 1 == len;       // the lane stores a pair / its first row only / nothing
    // the element beyond either end of the wave's 128 rows of a segment: lane 63 reads the one behind them, every other lane the
    // one in front (lane 0 uses it; one cache line for the rest) -- byte offset from the start of the segment
    const int edge_b = (t >> 6) * 1024 + ((t & 63) == 63 ? 1024 : -8);

    for (int i = t; i < 256; i += (int)blockDim.x) s_value[i] = values[i];
    __syncthreads();

    // ---- a line class -> values (into s_other) and validity (returned) of this lane's rows at the seven positions ----
    auto decode = [&](int cls) -> unsigned {
        const unsigned char *tb = table + (long long)cls * 7 * gd.pitch + row0 + 2 * t;
        unsigned bits = 0;
#pragma unroll
        for (int p = 0; p < 7; ++p) {
            const unsigned c2 = *reinterpret_cast<const unsigned short *>(tb + (long long)p * gd.pitch);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned code = (c2 >> (8 * r)) & 255u;
                s_other[2 * p + r][t] = s_value[code];            // entry 255 of the value table is 0.0
                bits |= (code != (unsigned)GR_ABSENT ? 1u : 0u) << (2 * p + r);
            }
        }
        return bits;
    };

    const int hot = gd.hot;
    double aH[7][2];                            // the hot class: values ...
    unsigned long long mH[7][2];                // ... and the lanes with an entry, per position and row
    {
        const unsigned bitsH = decode(hot);
#pragma unroll
        for (int p = 0; p < 7; ++p) {
            aH[p][0] = s_other[2 * p][t]; aH[p][1] = s_other[2 * p + 1][t];
            mH[p][0] = __builtin_amdgcn_ballot_w64((bitsH >> (2 * p)) & 1u);
            mH[p][1] = __builtin_amdgcn_ballot_w64((bitsH >> (2 * p + 1)) & 1u);
        }
    }
    unsigned bitsO = 0;
    int other = -1;                             // what s_other holds now is the hot class's: never asked for

    // clamped requests (prologue, slow steps): line `l` of the tile's window (0 = the line above the tile, 1 .. TY = the tile,
    // TY + 1 = the line below) in plane zz, element by element.  What lies outside x is never referenced by an entry; what is
    // loaded in its place is multiplied by +0.0 behind a mask
    auto elem = [&](long long i) -> double {
        if constexpr (HALO) {
            // an element of the plane below / above the device's planes: the neighbour's boundary plane of x, where it lies
            const long long plane = (long long)ny * nx;
            if (i < own_lo) { const long long j = i - (own_lo - plane); if (H.lo && j >= 0) return ghost_bad ? __builtin_nan("") : H.lo[j]; }
            else if (i > x_last) { const long long j = i - (x_last + 1); if (H.hi && j < plane) return ghost_bad ? __builtin_nan("") : H.hi[j]; }
            i = i < own_lo ? own_lo : i;
        }
        i = i < 0 ? 0 : i; i = i > x_last ? x_last : i; return x[i];
    };
    auto ld = [&](int zz, int l) -> d2 {
        const long long i = ((long long)zz * ny + (y0 - 1 + l)) * nx + row0 + 2 * t;
        d2 r; r.x = elem(i); r.y = elem(i + 1);
        return r;
    };
    auto edge = [&](int zz, int l) -> double {
        return elem(((long long)zz * ny + (y0 - 1 + l)) * nx + row0 + (edge_b >> 3));
    };
    auto yold = [&](int zz, int l) -> d2 {
        long long i = ((long long)zz * ny + (y0 + l)) * nx + row0 + 2 * t, j = i + 1;
        i = i < own_lo ? own_lo : i; i = i > n - 1 ? n - 1 : i; j = j < own_lo ? own_lo : j; j = j > n - 1 ? n - 1 : j;
        d2 r; r.x = zs[i]; r.y = zs[j];
        return r;
    };

    // ---- state at the top of the step for plane z: as in plane.hip ----
    // Cs[0..3]: the tile's two centre lines in planes z-1, z, z+1, z+2;  Hs[0..1]: the halo lines (above, below) in planes z, z+1;
    // Es[0..1]: per centre line the edge element of this lane in planes z, z+1.  The fast loop runs GROUPS of four steps with
    // the names rotated; every request has two steps to arrive.
    d2 Cs[4][TY], Hs[2][2], Yo[TY];
    double Es[2][TY];
    const unsigned line_b = (unsigned)nx * 8u;                        // bytes from a line to the next
    const unsigned plane_b32 = (unsigned)ny * line_b;                 // ... to the same line of the next plane (the plan: (depth + 4) of them < 2^32)
    const int z_first = z;
    // buffer resources of the fast loop: x from the segment of the line above the tile in the workgroup's first plane, y from
    // the tile's first line in that plane.  No range check (the scalar offset that moves with the planes takes no part in it):
    // the fast loop only runs where every request of every lane lies inside the arrays (zh below)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(x + (((long long)z_first * ny + (y0 - 1)) * nx + row0)), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y + (((long long)z_first * ny + y0) * nx + row0), 0, -1, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(ZM == 1 ? zs : x) + (((long long)z_first * ny + y0) * nx + row0), 0, -1, 0x00020000);
#pragma unroll
    for (int l = 0; l < TY; ++l) { Cs[0][l] = ld(z - 1, l + 1); Cs[1][l] = ld(z, l + 1); Cs[2][l] = ld(z + 1, l + 1); Cs[3][l] = ld(z + 2, l + 1); }
    // gd.flat (round 6): no entry at +-nx anywhere -- a 5-point operator on a 2-D grid, its rows cut into virtual lines: the lines above
    // and below the tile are never requested (two of the six 16-byte requests of a step)
    const bool flat = gd.flat != 0;                                   // uniform
    const d2 dzero = {0.0, 0.0};
    Hs[0][0] = Hs[0][1] = Hs[1][0] = Hs[1][1] = dzero;
    if (!flat) { Hs[0][0] = ld(z, 0); Hs[0][1] = ld(z, TY + 1); Hs[1][0] = ld(z + 1, 0); Hs[1][1] = ld(z + 1, TY + 1); }
#pragma unroll
    for (int l = 0; l < TY; ++l) {
        Es[0][l] = edge(z, l + 1); Es[1][l] = edge(z + 1, l + 1);
        if (ZM == 1) Yo[l] = yold(z, l);
    }

    // the lane's sums for tile line l: x at the seven positions from the registers named above
#define GRID_XS(P, C, N, H, E, l)                                                                                                    \
        const d2 c = C[l], up = (l) == 0 ? H[0] : C[(l) > 0 ? (l) - 1 : 0], dn = (l) == TY - 1 ? H[1] : C[(l) < TY - 1 ? (l) + 1 : 0];                                                  \
        const double xs0[7] = {P[l].x, up.x, shift_from_lower_lane(c.y, E[l]), c.x, c.y, dn.x, N[l].x};                                \
        const double xs1[7] = {P[l].y, up.y, c.x, c.y, shift_from_upper_lane(c.x, E[l]), dn.y, N[l].y};
#define GRID_HOT_SUMS(s0, s1)                                                                                                        \
        _Pragma("unroll") for (int p = 0; p < 7; ++p) { s0 += aH[p][0] * keep_lanes(xs0[p], mH[p][0]); s1 += aH[p][1] * keep_lanes(xs1[p], mH[p][1]); }
#define GRID_OTHER_SUMS(s0, s1)                                                                                                      \
        _Pragma("unroll") for (int p = 0; p < 7; ++p) {                                                                               \
            s0 += s_other[2 * p][t] * keep_bit(xs0[p], bitsO, 2 * p); s1 += s_other[2 * p + 1][t] * keep_bit(xs1[p], bitsO, 2 * p + 1); }

    // fast steps: every request lies inside the arrays.  A lane may sit beyond the end of its line (it stores nothing, but it
    // requests): `over` = the furthest element, from the start of a line, that a lane of this workgroup asks for
    int zh = zend;
    {
        const long long over = row0 + 2 * (long long)blockDim.x + 1;
        const long long xl_in = x_last - over >= 0 ? (x_last - over) / nx : -1;       // largest line all of whose requests are inside x
        const long long yl_in = n - 1 - over >= 0 ? (n - 1 - over) / nx : -1;         // ... inside y ('+=' reads the old y one plane ahead)
        // largest z with (z + ahead) * ny + y0 + line <= limit, + 1
        auto end_for = [&](long long limit, int ahead, int line) -> long long { const long long v = limit - y0 - line; return v < 0 ? 0 : v / ny - ahead + 1; };
        long long e = end_for(xl_in, 3, TY);                                           // x: planes up to z + 3, lines up to the one below the tile
        e = std::min(e, end_for(lines - 1, 0, TY - 1));                               // y: both lines exist
        if (ZM == 1) e = std::min(e, end_for(yl_in, 1, TY - 1));
        zh = zh < e ? zh : (int)(e < 0 ? 0 : e);
    }

    while (z < zend) {
        // ---- how many of the next planes (<= 64) can take fast steps: both lines use the hot class or the other class ----
        unsigned long long use_hot[TY];          // bit k: line l of plane z + k uses the hot class (else: the other class)
        int run;
        {
            const int k = t & 63, zz = z + k;
            const bool in = zz < zh;
            bool ok = in;
#pragma unroll
            for (int l = 0; l < TY; ++l) {
                const int bk = (in && l < nl) ? line_class[(long long)zz * ny + (y0 + l)] : hot;
                ok = ok && (bk == hot || bk == other);
                use_hot[l] = __builtin_amdgcn_ballot_w64(bk == hot);
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
            run = ~m ? __builtin_ctzll(~m) : 64;
        }
        if (run >= 4) {
            // xo: plane z + 2, the line above the tile; yo: plane z, the tile's first line; both relative to the workgroup's first plane
            unsigned xo = (unsigned)((z + 2 - z_first) * plane_b32), yo = (unsigned)((z - z_first) * plane_b32);
            auto fast_step = [&](d2 (&P)[TY], d2 (&C)[TY], d2 (&N)[TY], d2 (&H)[2], double (&E)[TY]) {
                d2 o[TY];
#pragma unroll
                for (int l = 0; l < TY; ++l) {
                    GRID_XS(P, C, N, H, E, l)
                    double s0 = 0.0, s1 = 0.0;
                    if (use_hot[l] & 1ull) { GRID_HOT_SUMS(s0, s1) } else { GRID_OTHER_SUMS(s0, s1) }      // uniform
                    o[l].x = alpha * s0; o[l].y = alpha * s1;
                    if (ZM == 1) { o[l].x = beta * Yo[l].x + o[l].x; o[l].y = beta * Yo[l].y + o[l].y; }
                    if (ZM == 2) { o[l].x = beta * c.x + o[l].x; o[l].y = beta * c.y + o[l].y; }
                }
#pragma unroll
                for (int l = 0; l < TY; ++l) use_hot[l] >>= 1;
#pragma unroll
                for (int l = 0; l < TY; ++l)
                    if (l < nl) {                  // uniform; written once, not read again by this kernel
                        if (full) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o[l]), ry, (int)lane_b, (int)(yo + l * line_b), STORE_AUX);
                        else if (half) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, o[l].x), ry, (int)lane_b, (int)(yo + l * line_b), STORE_AUX);
                    }
                if (ZM == 1) {
#pragma unroll
                    for (int l = 0; l < TY; ++l) Yo[l] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rz, (int)lane_b, (int)(yo + plane_b32 + l * line_b), 0));
                }
                if (!flat) {
                    H[0] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)lane_b, (int)xo, 0));
                    H[1] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)lane_b, (int)(xo + (TY + 1) * line_b), 0));
                }
#pragma unroll
                for (int l = 0; l < TY; ++l) {
                    P[l] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)lane_b, (int)(xo + plane_b32 + (l + 1) * line_b), 0));
                    E[l] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx, edge_b + 8, (int)(xo + (l + 1) * line_b - 8u), 0));
                }
                xo += plane_b32; yo += plane_b32; ++z;
            };
            for (int g = run >> 2; g > 0; --g) {
                fast_step(Cs[0], Cs[1], Cs[2], Hs[0], Es[0]);
                fast_step(Cs[1], Cs[2], Cs[3], Hs[1], Es[1]);
                fast_step(Cs[2], Cs[3], Cs[0], Hs[0], Es[0]);
                fast_step(Cs[3], Cs[0], Cs[1], Hs[1], Es[1]);
            }
            if (run == 64) continue;                                      // look again: the run may go on
        }
        if (z >= zend) break;
        // ---- a slow step: a line needs another class decoded, the last planes (clamped requests), the ragged last plane, what
        // a run leaves over after its groups of four; names rotated by copies ----
#pragma unroll
        for (int l = 0; l < TY; ++l) {
            const long long li = (long long)z * ny + (y0 + l);
            if (l < nl && li < lines) {                                   // uniform
                GRID_XS(Cs[0], Cs[1], Cs[2], Hs[0], Es[0], l)
                double s0 = 0.0, s1 = 0.0;
                const int cls = __builtin_amdgcn_readfirstlane(line_class[li]);
                if (cls == hot) { GRID_HOT_SUMS(s0, s1) }
                else {
                    if (cls != other) { bitsO = decode(cls); other = cls; }
                    GRID_OTHER_SUMS(s0, s1)
                }
                d2 o; o.x = alpha * s0; o.y = alpha * s1;
                if (ZM == 1) { o.x = beta * Yo[l].x + o.x; o.y = beta * Yo[l].y + o.y; }
                if (ZM == 2) { o.x = beta * c.x + o.x; o.y = beta * c.y + o.y; }
                double *yr = y + li * nx + row0 + 2 * t;
                if (full || half) __builtin_nontemporal_store(o.x, yr);
                if (full) __builtin_nontemporal_store(o.y, yr + 1);
            }
        }
#pragma unroll
        for (int l = 0; l < TY; ++l) {
            Cs[0][l] = Cs[1][l]; Cs[1][l] = Cs[2][l]; Cs[2][l] = Cs[3][l]; Cs[3][l] = ld(z + 3, l + 1);
            Es[0][l] = Es[1][l]; Es[1][l] = edge(z + 2, l + 1);
            if (ZM == 1) Yo[l] = yold(z + 1, l);
        }
#pragma unroll
        for (int l = 0; l < 2; ++l) { Hs[0][l] = Hs[1][l]; if (!flat) Hs[1][l] = ld(z + 2, (TY + 1) * l); }
        ++z;
    }
#undef GRID_XS
#undef GRID_HOT_SUMS
#undef GRID_OTHER_SUMS
}

template <int ZM, int STORE_AUX, int MAXT, bool HALO = false>
__global__ __launch_bounds__(MAXT, MAXT == 256 ? 4 : 2)
void sell8_grid_kernel(const double *__restrict__ x, double *__restrict__ y, double alpha, const double *__restrict__ zs, double beta,
        const int *__restrict__ line_class, const unsigned char *__restrict__ table, const double *__restrict__ values, grid_dev gd, halo_dev H)
{
    if constexpr (!HALO) {
        grid_walk<ZM, STORE_AUX, MAXT, false>(x, y, alpha, zs, beta, line_class, table, values, gd, H, 0ull);
    } else {
        const unsigned long long step = *H.step;
        halo_announce(H, step);
        grid_walk<ZM, STORE_AUX, MAXT, true>(x, y, alpha, zs, beta, line_class, table, values, gd, H, step);
        halo_finish(H, step);
    }
}


// segments of <= 512 rows (even), lanes for one segment, bytes per position row of a class table, planes per workgroup
struct grid_geometry { long long segs, seg_len, pitch, depth; int threads; };
inline long long grid_pitch(long long nx) { return (((nx + 511) / 512) * 512 + 2 + 15) / 16 * 16; }

bool grid_geometry_with(long long cus, long long nx, long long ny, long long nz, grid_geometry *geo);
bool grid_geometry_for(int dev, long long nx, long long ny, long long nz, grid_geometry *geo)
{
    return grid_geometry_with(std::max(1, info(dev).cus), nx, ny, nz, geo);
}
// (host arithmetic only: vexhip_sell8_grid_geometry exposes it so that the plans of every line length can be checked without a device)
bool grid_geometry_with(long long cus, long long nx, long long ny, long long nz, grid_geometry *geo)
{
    // lines of up to 1024 points in ONE segment (workgroups of up to 512 lanes), longer ones in segments of <= 1024 rows.
    // (Round 4 cut 513 .. 768-point lines in two: 640^3 / 700^3 in one segment took 1.26 / 1.63 ms against 0.98 / 1.51 ms in two --
    //  but only because one segment was given ONE walk per tile; with the walks chosen below it takes 0.874 / 1.27 ms,
    //  profiles/r05_grid_long_lines.json.  VEXHIP_GRID_SEGMENT = 512 | 1024 overrides.)
    long long max_seg = nx > 512 ? 1024 : 512;
    if (const char *e = env(ENV_VEXHIP_GRID_SEGMENT)) max_seg = std::atoi(e) == 1024 ? 1024 : 512;
    const long long segs = (nx + max_seg - 1) / max_seg;
    long long seg_len = (nx + segs - 1) / segs; seg_len += seg_len & 1;
    const int threads = (int)std::min<long long>(max_seg / 2, ((seg_len + 1) / 2 + 63) / 64 * 64);
    // a position row of a class table covers what the lanes of the LAST segment read: lanes beyond the end of the line read
    // (and ignore) two bytes each.  Lines of 2521 .. 2560 points -- three segments of 842 .. 854 rows, 448 lanes -- reach up to
    // 28 bytes beyond the rounded line length: the plan of such a matrix was refused by the product ("bad grid plan")
    const long long pitch = (std::max(grid_pitch(nx), (segs - 1) * seg_len + 2 * (long long)threads) + 15) / 16 * 16;
    const long long tiles = (ny + 1) / 2 * segs;
    cus = std::max(1ll, cus);
    // Planes per workgroup.  Few, long workgroups (plane.hip: every workgroup re-reads the planes around its walk) -- but
    //   * lines that are not 512 points long want at least ~6 waves per CU (their requests straddle cache lines, a wave hides
    //     less of the latency by itself), and
    //   * the workgroups should fill the CUs EVENLY: all of them are resident at once, the ones that share a CU with more
    //     neighbours fall behind, and tiles that are not at the same plane at the same time fetch their halo lines from HBM
    //     instead of from the L2 their neighbour filled.
    // Measured (profiles/r04_grid_sweep.json; pair product 0.312 / 0.781 ms): 384^3 (192 tiles x 3 waves) walks of 48 / 96 /
    // 128 / 192 / 384 planes = 0.202 / 0.184 / 0.217 / 0.227 / 0.349 ms -- 96 planes = 768 workgroups = 3 per CU exactly; 500^3
    // (250 x 4) walks of 62 / 125 / 166 / 250 / 500 = 0.433 / 0.419 / 0.415 / 0.389 / 0.520 ms.  The estimate below orders all of
    // these as measured: (workgroups per CU, rounded up) x (planes per walk + 6 for the start of a walk), times 6 / (waves per
    // CU) where that is below 6.
    const long long wpw = threads / 64;
    long long chunks = 1;
    if (threads <= 256) {
        double best = 0;
        const long long resident = std::max(1ll, 16 / wpw);       // workgroups a CU holds at once (128 registers per lane)
        for (long long c = 1; c <= std::max(1ll, nz / 8); ++c) {
            const long long per_cu = (tiles * c + cus - 1) / cus;
            if (c > 1 && per_cu > resident) break;              // a second round of workgroups walks out of step with the first
            double est = (double)per_cu * (double)((nz + c - 1) / c + 6);
            if (per_cu * wpw < 6) est *= 6.0 / (double)(per_cu * wpw);
            if (c == 1 || est < best) { best = est; chunks = c; }
        }
    } else {
        // Round 5 -- workgroups of 5 .. 8 waves (lines of 513 .. 1024 points): a CU holds two or three of them and a plane has
        // 1.25 .. 2 tiles per CU, so ONE walk per tile leaves half of the CUs with twice the work of the others (640^3: 1.25 ms).
        // Many short walks even that out through the dispatcher: at least six workgroups per CU, then the same estimate (rounds x
        // (planes per walk + 6)), ties to the shorter walk.  Measured (profiles/r05_grid_long_lines.json, ms by walk depth):
        //   640^3  640 / 160 / 80 / 40    1.252 / 0.875 / 0.874 / 0.891        768^3  768 / 192 / 96 / 48     1.728 / 1.408 / 1.425 / 1.429
        //   700^3  700 / 175 / 88 / 44    1.628 / 1.316 / 1.270 / 1.277        800^3  800 / 200 / 80 / 40     2.049 / 1.814 / 1.694 / 1.701
        //   900^3  450 / 150 / 90 / 45    2.704 / 2.572 / 2.546 / 2.537        1024^3 256 / 171 / 128 / 64    3.317 / 3.359 / 3.365 / 3.523
        double best = 0;
        const long long cmax = std::max(1ll, nz / 16);
        for (long long c = 1; c <= cmax; ++c) {
            const long long per_cu = (tiles * c + cus - 1) / cus;
            if (per_cu < 6 && c < cmax) continue;
            const double est = (double)per_cu * (double)((nz + c - 1) / c + 6);
            if (best == 0 || est <= best * 1.01) { if (best == 0 || est < best) best = est; chunks = c; }
            if (per_cu > 24) break;
        }
    }
    long long depth = (nz + chunks - 1) / chunks;
    if (const char *e = env(ENV_VEXHIP_PLANE_DEPTH)) depth = std::max(1, std::atoi(e));
    depth = std::min(depth, nz);
    const long long plane_bytes = ny * nx * 8;
    while ((depth + 4) * plane_bytes >= (1ll << 32) && depth > 8) depth = (depth + 1) / 2;
    if ((depth + 4) * plane_bytes >= (1ll << 32)) return false;

    geo->segs = segs; geo->seg_len = seg_len; geo->pitch = pitch; geo->depth = depth; geo->threads = threads;
    return true;
}

void grid_fill_plan(vexhip_grid *out, long long nx, long long ny, long long nz, const grid_geometry &geo, int hot, int nclasses, long long x_last)
{
    out->nx = (int32_t)nx; out->lines_per_plane = (int32_t)ny; out->planes = (int32_t)nz; out->depth = (int32_t)geo.depth;
    out->segments = (int32_t)geo.segs; out->segment_rows = (int32_t)geo.seg_len; out->threads = geo.threads;
    out->hot_class = hot; out->classes = nclasses; out->pitch = (int32_t)geo.pitch;
    out->store_policy = 1;
    if (const char *e = env(ENV_VEXHIP_PLANE_STORE)) out->store_policy = std::max(0, std::min(3, std::atoi(e)));
    out->x_last = x_last;
    out->flat = 0; out->reserved = 0;
}

} // namespace
} // namespace vexhip
/* the virtual grid line a 2-D row of `row_length` points is cut into (vexhip.h): host arithmetic only */
extern "C" int64_t vexhip_sell8_grid_virtual_line(int64_t W)
{
    if (W < 16 || W > (1ll << 30)) return 0;
    if (W % 512 == 0 && (W / 512) % 2 == 0 && W / 512 >= 4) return 512;
    for (long long d = std::min<long long>(1024, W / 10); d >= 128; --d)
        if (W % d == 0 && d % 2 == 0) return d;
    return 0;
}
namespace vexhip {
namespace {

// the diagonals: {0, +-1, +-nx, +-P} with 1 < nx < P, P a multiple of nx (a subset that names both); false: not a grid matrix
bool grid_diagonals(const std::vector<int> &table, long long rows, long long *nx_out, long long *far_out)
{
    std::vector<long long> mags;
    for (int d : table) { const long long a = std::llabs((long long)d); if (a && std::find(mags.begin(), mags.end(), a) == mags.end()) mags.push_back(a); }
    std::sort(mags.begin(), mags.end());
    if (mags.size() == 2 && mags[0] == 1) {
        // Round 5 -- a 5-point operator on a 2-D grid (or any band matrix {0, +-1, +-W}): no line-above / line-below pair.  Its rows
        // of W points are cut into VIRTUAL lines of nx points (W / nx of them make a "plane"), so that +-W is the far pair of the
        // walk and positions +-nx simply never occur; the +-1 entries that join two virtual lines of one row are ordinary entries of
        // the class tables (all x addressing is linear).  The reference's SpMatCCSR has no notion of dimension either
        // (spmat/ccsr.hpp:55-113).  Taken where 512-point lines fit (an even number of them, four or more): the plane product.
        const long long W = mags[1];
        if (W < 16 || rows % W != 0 || W > (1ll << 30)) return false;
        // Virtual lines: 512 points where an even number (>= 4) of them make a row (the plane product); round 6: any other row length -- the
        // walk of a FLAT plan (vexhip_grid.flat) no longer requests the lines above and below a tile, which is what kept these rows behind
        // the pair product in round 5 (below) -- the longest even divisor of W up to 1024 points, ten or more of them per row (the first and
        // the last line of a row are classes of their own, so are the lines of the first and the last row: the hot class must keep three
        // lines in four).  vexhip_sell8_grid_virtual_line is this rule by itself; VEXHIP_GRID_2D_LINE = points per virtual line (a
        // divisor of W) overrides the second part, 0: as in round 5.
        long long nx = vexhip_sell8_grid_virtual_line(W);
        if (nx != 512) {
            if (const char *e = env(ENV_VEXHIP_GRID_2D_LINE)) {
                const long long want = std::atoll(e);
                nx = (want > 0 && W % want == 0 && want >= 8) ? want : (want == 0 ? 0 : nx);
            }
        }
        // (other row lengths walked along a divisor of W -- 10000^2 as 20 lines of 500 points, 12000 x 9000, 7000 x 20000 -- ran level
        //  with the pair product of the SELL-512 storage or behind it, 0.528 / 0.462 / 0.656 ms against 0.493 / 0.467 / 0.678: few
        //  lines per "plane", and the walk still requests the line above and below, which a 2-D operator never uses.  They keep the
        //  pair product; tools/r05_2d.py, profiles/r05_2d.json)
        if (nx < 8) return false;
        *nx_out = nx; *far_out = W;
        return true;
    }
    if (mags.size() != 3 || mags[0] != 1) return false;
    const long long nx = mags[1], far = mags[2];
    if (nx < 8 || far % nx != 0 || far / nx < 2 || rows % nx != 0 || far > (1ll << 30)) return false;
    *nx_out = nx; *far_out = far;
    return true;
}

template <typename T> struct dev_buf {       // device scratch of the plan, freed on every exit path
    T *p = nullptr;
    ~dev_buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1) * sizeof(T)); }
    void free_now() { if (p) (void)hipFree(p); p = nullptr; }
    T *release() { T *r = p; p = nullptr; return r; }
};


// ---- host side of the direct build ----
// rows x rows-or-more matrix in CSR on the device -> vexhip_grid (usable = 1), the diagonal table (sorted, 256 ints on the device,
// INT_MAX behind the last) and the value table (256 values on the device, 0.0 behind the last), ELL width and largest column;
// usable = 0: not a matrix for this storage, nothing was written that the classic set-up would read.
// fp32 (V = float): the same tables; the products are plane32.hip (512-point lines) and grid32.hip (any line length).
template <typename P, typename V>
int grid_build(int dev, void *stream, int64_t rows, const P *ptr, const int32_t *col, const V *val,
        int32_t *deltas, V *values, int *ndeltas, int *nvalues, int64_t *ell_width, int64_t *x_last_out, vexhip_grid *out, int64_t min_cols)
{
    VEXHIP_REQUIRE(out && ndeltas && nvalues && ell_width && x_last_out, "NULL output");
    std::memset(out, 0, sizeof(*out));
    *ndeltas = -1; *nvalues = -1; *ell_width = 0; *x_last_out = -1;
    const bool force = env(ENV_VEXHIP_PLANE_FORCE) != nullptr;          // tests: small grids
    if (env(ENV_VEXHIP_NO_GRID) || env(ENV_VEXHIP_NO_GRID_BUILD)) return 0;
    // small matrices (x within the L2s / the Infinity Cache) keep the pair product of the SELL-512 storage
    if (!ptr || !col || !val || !deltas || !values || rows < 64 || (rows < (1 << 23) && !force) || rows >= (1ll << 31)) return 0;
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    setup_trace trace(s);

    // ---- probe: the diagonals of (up to) 8192 rows in the middle of the matrix ----
    // [diagonal set 16 ints][value info 2 ints][pad][value set GB_VSLOTS x 8]
    dev_buf<unsigned long long> d_probe;
    VEXHIP_TRY(d_probe.alloc(16 + GB_VSLOTS));
    int *d_set = reinterpret_cast<int *>(d_probe.p), *d_vinfo = d_set + 16;
    unsigned long long *d_vset = d_probe.p + 16;
    VEXHIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_set), (int)0x80000000, 15, s));
    VEXHIP_TRY(hipMemsetAsync(d_set + 15, 0, sizeof(int) * 3, s));
    VEXHIP_TRY(hipMemsetAsync(d_vset, 0xff, sizeof(unsigned long long) * GB_VSLOTS, s));
    const long long probe_rows = std::min<long long>(rows, 8192), probe_first = (rows - probe_rows) / 2;
    grid_probe_kernel<P, V><<<(unsigned)((probe_rows + 255) / 256), 256, 0, s>>>(ptr, col, val, probe_first, probe_rows, d_set, d_vset, d_vinfo);
    VEXHIP_LAUNCH_CHECK();
    int set[18];
    VEXHIP_TRY(hipMemcpyAsync(set, d_set, sizeof(set), hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    if (set[15] || set[17]) return 0;              // a diagonal set that is not a grid's, or more values than codes: the classic set-up
    std::vector<int> table;
    for (int k = 0; k < 15; ++k) if (set[k] != (int)0x80000000) table.push_back(set[k]);
    if (table.size() < 4 || table.size() > 7) return 0;
    long long nx = 0, far = 0;
    if (!grid_diagonals(table, rows, &nx, &far) || nx > 4096) return 0;         // (the line's seven position rows live in LDS)
    const long long ny = far / nx, lines = rows / nx, nz = (lines + ny - 1) / ny;
    if (nz < 4 && !force) return 0;
    grid_geometry geo;
    if (!grid_geometry_for(dev, nx, ny, nz, &geo)) return 0;
    trace.mark("  grid: probe");

    // ---- the pass ----
    // control block: [keys GB_KEYS x 8][vkeys GB_VSLOTS x 8][ids GB_KEYS x 4][vstate GB_VSLOTS x 4][vcodes GB_VSLOTS x 4][ints GBI_INTS x 4]
    const size_t ctl_bytes = GB_KEYS * 8 + GB_VSLOTS * 8 + GB_KEYS * 4 + GB_VSLOTS * 4 + GB_VSLOTS * 4 + GBI_INTS * 4;
    dev_buf<unsigned char> d_ctl, d_table; dev_buf<int> d_cls;
    VEXHIP_TRY(d_ctl.alloc(ctl_bytes)); VEXHIP_TRY(d_cls.alloc((size_t)lines)); VEXHIP_TRY(d_table.alloc((size_t)GB_MAX_CLASSES * 7 * (size_t)geo.pitch));
    gb_dev g;
    g.n = rows; g.lines = lines; g.far = far; g.nx = (int)nx; g.pitch = (int)geo.pitch; g.pieces = (int)((nx + GB_PIECE - 1) / GB_PIECE);
    g.keys = reinterpret_cast<unsigned long long *>(d_ctl.p);
    g.vkeys = g.keys + GB_KEYS;
    g.ids = reinterpret_cast<int *>(g.vkeys + GB_VSLOTS);
    g.vstate = g.ids + GB_KEYS; g.vcodes = g.vstate + GB_VSLOTS; g.ints = g.vcodes + GB_VSLOTS;
    g.table = d_table.p; g.line_class = d_cls.p;
    VEXHIP_TRY(hipMemsetAsync(d_ctl.p, 0, ctl_bytes, s));
    VEXHIP_TRY(hipMemsetAsync(g.ids, 0xff, GB_KEYS * 4, s));                     // -1: no number yet
    VEXHIP_TRY(hipMemsetAsync(g.ints + GBI_MAXCOL, 0xff, sizeof(int), s));        // -1
    const size_t lds = GB_VSLOTS * 8 + GB_VSLOTS + 520 * 8 + GB_CAP * 4 + GB_CAP + ((7 * (size_t)geo.pitch + 15) / 16) * 16 + 16 * 8;
    const long long cus = std::max(1, info(dev).cus);
    long long per_cu = std::max<long long>(1, std::min<long long>(4, (150 * 1024) / (long long)(lds + 2048)));
    if (const char *e = env(ENV_VEXHIP_GRID_BUILD_WGS)) per_cu = std::max(1, std::atoi(e));
    const unsigned wgs = (unsigned)std::min<long long>(lines, cus * per_cu);
    grid_build_kernel<P, V><<<wgs, 256, lds, s>>>(ptr, col, val, g);
    VEXHIP_LAUNCH_CHECK();
    std::vector<unsigned char> ctl(ctl_bytes);
    VEXHIP_TRY(hipMemcpyAsync(ctl.data(), d_ctl.p, ctl_bytes, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    trace.mark("  grid: pass");
    const unsigned long long *h_vkeys = reinterpret_cast<const unsigned long long *>(ctl.data()) + GB_KEYS;
    const int *h_ids = reinterpret_cast<const int *>(h_vkeys + GB_VSLOTS);
    const int *h_vstate = h_ids + GB_KEYS, *h_vcodes = h_vstate + GB_VSLOTS, *h_ints = h_vcodes + GB_VSLOTS;
    if (env(ENV_VEXHIP_DEBUG))
        std::fprintf(stderr, "grid build: nx %lld ny %lld lines %lld classes %d values %d flags %d positions %#x max column %d widest row %d\n",
                     nx, ny, lines, h_ints[GBI_COUNT], h_ints[GBI_VCOUNT], h_ints[GBI_FLAGS], h_ints[GBI_POSMASK], h_ints[GBI_MAXCOL], h_ints[GBI_MAXLEN]);
    const int nclasses = h_ints[GBI_COUNT], nv = h_ints[GBI_VCOUNT];
    if (h_ints[GBI_FLAGS] != 0 || nclasses < 1 || nclasses > GB_MAX_CLASSES || nv < 1 || nv > GB_MAX_VALUES) return 0;   // the classic set-up takes over
    const long long x_last = std::max<long long>(h_ints[GBI_MAXCOL], min_cols - 1);      // min_cols: the caller vouches for that many elements of x (VEXHIP_SPMAT_SQUARE)
    if (x_last < 0 || x_last + 1 < rows) return 0;
    const int *uses = h_ints + GBI_USES;
    const int hot = (int)(std::max_element(uses, uses + nclasses) - uses);
    if ((lines - uses[hot]) * 4 > lines && !force) return 0;
    // the tables the products read: values by code (0.0 behind the last: code 255 reads it), diagonals sorted
    std::vector<V> vals(256, V(0));
    for (int k = 0; k < GB_VSLOTS; ++k)
        if (h_vstate[k] == 2 && h_vcodes[k] >= 0 && h_vcodes[k] < nv) { double v; std::memcpy(&v, &h_vkeys[k], sizeof(double)); vals[(size_t)h_vcodes[k]] = (V)v; }
    const long long by_pos[7] = {-far, -nx, -1, 0, 1, nx, far};
    std::vector<int> dl;
    for (int p = 0; p < 7; ++p) if (h_ints[GBI_POSMASK] & (1 << p)) dl.push_back((int)by_pos[p]);
    const int nd = (int)dl.size();
    if (nd < 1) return 0;
    dl.resize(256, INT_MAX);
    VEXHIP_TRY(hipMemcpyAsync(values, vals.data(), sizeof(V) * 256, hipMemcpyHostToDevice, s));
    VEXHIP_TRY(hipMemcpyAsync(deltas, dl.data(), sizeof(int) * 256, hipMemcpyHostToDevice, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    grid_fill_plan(out, nx, ny, nz, geo, hot, nclasses, x_last);
    out->flat = (h_ints[GBI_POSMASK] & ((1 << 1) | (1 << 5))) == 0;
    out->line_class = d_cls.release(); out->table = d_table.release();
    out->usable = 1;
    *ndeltas = nd; *nvalues = nv; *ell_width = h_ints[GBI_MAXLEN]; *x_last_out = x_last;
    trace.mark("  grid: tables");
    d_probe.free_now(); d_ctl.free_now();
    trace.mark("  grid: scratch freed");
    return 0;
}

} // namespace

// internal entry points of the direct build (spmat.hip)
int grid_build_p32(int dev, void *stream, int64_t rows, const int32_t *ptr, const int32_t *col, const double *val,
        int32_t *deltas, double *values, int *ndeltas, int *nvalues, int64_t *ell_width, int64_t *x_last, vexhip_grid *out, int64_t min_cols)
{ return grid_build<int32_t, double>(dev, stream, rows, ptr, col, val, deltas, values, ndeltas, nvalues, ell_width, x_last, out, min_cols); }
int grid_build_p64(int dev, void *stream, int64_t rows, const long long *ptr, const int32_t *col, const double *val,
        int32_t *deltas, double *values, int *ndeltas, int *nvalues, int64_t *ell_width, int64_t *x_last, vexhip_grid *out, int64_t min_cols)
{ return grid_build<long long, double>(dev, stream, rows, ptr, col, val, deltas, values, ndeltas, nvalues, ell_width, x_last, out, min_cols); }
int grid_build_p32(int dev, void *stream, int64_t rows, const int32_t *ptr, const int32_t *col, const float *val,
        int32_t *deltas, float *values, int *ndeltas, int *nvalues, int64_t *ell_width, int64_t *x_last, vexhip_grid *out, int64_t min_cols)
{ return grid_build<int32_t, float>(dev, stream, rows, ptr, col, val, deltas, values, ndeltas, nvalues, ell_width, x_last, out, min_cols); }
int grid_build_p64(int dev, void *stream, int64_t rows, const long long *ptr, const int32_t *col, const float *val,
        int32_t *deltas, float *values, int *ndeltas, int *nvalues, int64_t *ell_width, int64_t *x_last, vexhip_grid *out, int64_t min_cols)
{ return grid_build<long long, float>(dev, stream, rows, ptr, col, val, deltas, values, ndeltas, nvalues, ell_width, x_last, out, min_cols); }

} // namespace vexhip

using namespace vexhip;

namespace vexhip {
int grid_apply_axpby(int dev, void *stream, int64_t n, double alpha, int zm, const double *zs, double beta, const double *values,
        const double *x, double *y, const vexhip_grid *g);
}

extern "C" {

#define GP_DECLINE(k) do { if (env(ENV_VEXHIP_DEBUG)) std::fprintf(stderr, "grid plan from the SELL-512 storage: declined at check %d (grid.hip:%d)\n", (k), __LINE__); return 0; } while (0)
int vexhip_sell8_grid_plan(int dev, void *stream, const int32_t *deltas, int ndeltas, const void *codes, const int32_t *blocks,
        int64_t ell_width, int64_t rows, int64_t tail_nnz, int value_bytes, int64_t x_last, vexhip_grid *out)
{
    reload_env();
    VEXHIP_REQUIRE(out, "NULL output");
    std::memset(out, 0, sizeof(*out));
    const bool force = env(ENV_VEXHIP_PLANE_FORCE) != nullptr;          // tests: small grids
    if (env(ENV_VEXHIP_NO_GRID)) GP_DECLINE(1);
    if ((value_bytes != 8 && value_bytes != 4) || !deltas || !codes || ndeltas < 4 || ndeltas > 7) GP_DECLINE(2);
    // small matrices (x within the L2s / the Infinity Cache: 127^3 = 0.019 ms here, 0.016 ms through the pair product; 168^3 0.029 /
    // 0.026; 256^3 0.052 / 0.074) keep the pair product
    if (ell_width < 1 || ell_width > 8 || tail_nnz != 0 || rows < 8 || (rows < (1 << 23) && !force)) GP_DECLINE(3);
    if (x_last < 0 || x_last + 1 < rows || x_last >= (1ll << 31)) GP_DECLINE(4);
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    std::vector<int> table((size_t)ndeltas);
    VEXHIP_TRY(hipMemcpyAsync(table.data(), deltas, sizeof(int) * (size_t)ndeltas, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    long long nx = 0, far = 0;
    if (!grid_diagonals(table, rows, &nx, &far)) GP_DECLINE(5);
    const long long ny = far / nx, lines = rows / nx, nz = (lines + ny - 1) / ny;
    if (nz < 4 && !force) GP_DECLINE(6);

    codes_dev cd;
    cd.buf = static_cast<const char *>(codes); cd.blocks = blocks; cd.w = (int)ell_width; cd.ndeltas = ndeltas;
    std::memset(cd.pos, 0, sizeof(cd.pos));
    for (int c = 0; c < ndeltas; ++c) {
        const long long d = table[(size_t)c];
        cd.pos[c] = (signed char)(d == 0 ? 3 : d == -1 ? 2 : d == 1 ? 4 : d == -nx ? 1 : d == nx ? 5 : d == -far ? 0 : 6);
    }
    // pass 1: a hash per line, rows checked (codes in the table, positions ascending)
    dev_buf<unsigned long long> d_hash; dev_buf<int> d_bad;
    VEXHIP_TRY(d_hash.alloc((size_t)lines)); VEXHIP_TRY(d_bad.alloc(1));
    VEXHIP_TRY(hipMemsetAsync(d_bad.p, 0, sizeof(int), s));
    const unsigned wgs = (unsigned)((lines + 3) / 4);
    grid_line_hash_kernel<<<wgs, 256, 0, s>>>(cd, lines, (int)nx, d_hash.p, d_bad.p);
    VEXHIP_LAUNCH_CHECK();
    std::vector<unsigned long long> hash((size_t)lines);
    int bad = 0;
    VEXHIP_TRY(hipMemcpyAsync(hash.data(), d_hash.p, sizeof(unsigned long long) * (size_t)lines, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipMemcpyAsync(&bad, d_bad.p, sizeof(int), hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    if (bad) GP_DECLINE(7);
    // classes: lines with equal hashes (a handful; the verify pass below compares the rows themselves)
    constexpr size_t max_classes = 128;
    std::vector<unsigned long long> seen; std::vector<long long> rep, uses;
    std::vector<int> cls((size_t)lines);
    size_t lastc = 0;
    for (long long l = 0; l < lines; ++l) {
        const unsigned long long h = hash[(size_t)l];
        size_t c = seen.size();
        if (!seen.empty() && seen[lastc] == h) c = lastc;
        else for (size_t k = 0; k < seen.size(); ++k) if (seen[k] == h) { c = k; break; }
        if (c == seen.size()) {
            if (seen.size() == max_classes) GP_DECLINE(8);
            seen.push_back(h); rep.push_back(l); uses.push_back(0);
        }
        cls[(size_t)l] = (int)c; ++uses[c]; lastc = c;
    }
    const int nclasses = (int)seen.size();
    const int hot = (int)(std::max_element(uses.begin(), uses.end()) - uses.begin());
    if ((lines - uses[(size_t)hot]) * 4 > lines && !force) GP_DECLINE(9);         // each change of the other class is a decode, a step with another class reads its values from LDS

    grid_geometry geo;
    if (!grid_geometry_for(dev, nx, ny, nz, &geo)) GP_DECLINE(10);
    const long long pitch = geo.pitch;
    // the table of every class from its representative line, then every line against it
    dev_buf<long long> d_rep; dev_buf<int> d_cls; dev_buf<unsigned char> d_table;
    VEXHIP_TRY(d_rep.alloc((size_t)nclasses)); VEXHIP_TRY(d_cls.alloc((size_t)lines)); VEXHIP_TRY(d_table.alloc((size_t)nclasses * 7 * (size_t)pitch));
    VEXHIP_TRY(hipMemcpyAsync(d_rep.p, rep.data(), sizeof(long long) * (size_t)nclasses, hipMemcpyHostToDevice, s));
    VEXHIP_TRY(hipMemcpyAsync(d_cls.p, cls.data(), sizeof(int) * (size_t)lines, hipMemcpyHostToDevice, s));
    grid_line_table_kernel<<<dim3((unsigned)((pitch + 255) / 256), (unsigned)nclasses), 256, 0, s>>>(cd, d_rep.p, (int)nx, (int)pitch, d_table.p);
    VEXHIP_LAUNCH_CHECK();
    grid_line_verify_kernel<<<wgs, 256, 0, s>>>(cd, lines, (int)nx, (int)pitch, d_cls.p, d_table.p, d_bad.p);
    VEXHIP_LAUNCH_CHECK();
    VEXHIP_TRY(hipMemcpyAsync(&bad, d_bad.p, sizeof(int), hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    if (bad) GP_DECLINE(11);                      // two different lines with one hash: not this product's matrix

    grid_fill_plan(out, nx, ny, nz, geo, hot, nclasses, x_last);
    out->flat = 1;
    for (int c = 0; c < ndeltas; ++c) if (cd.pos[c] == 1 || cd.pos[c] == 5) out->flat = 0;
    out->line_class = d_cls.release(); out->table = d_table.release();
    out->usable = 1;
    return 0;
}

#undef GP_DECLINE
int vexhip_sell8_grid_release(int dev, vexhip_grid *g)
{
    if (!g) return 0;
    if (g->line_class || g->table) {
        VEXHIP_SET_DEVICE(dev);
        if (g->line_class) (void)hipFree(const_cast<int32_t *>(g->line_class));
        if (g->table) (void)hipFree(const_cast<void *>(g->table));
    }
    std::memset(g, 0, sizeof(*g));
    return 0;
}

int vexhip_sell8_grid_geometry(int cus, int64_t nx, int64_t lines_per_plane, int64_t planes, vexhip_grid *out)
{
    VEXHIP_REQUIRE(out, "NULL output");
    std::memset(out, 0, sizeof(*out));
    reload_env();
    VEXHIP_REQUIRE(nx >= 8 && nx < (1ll << 30) && lines_per_plane >= 2 && lines_per_plane < (1ll << 30) && planes >= 1 && planes < (1ll << 30), "bad grid");
    grid_geometry geo;
    if (!grid_geometry_with(cus, nx, lines_per_plane, planes, &geo)) return 0;          // depth = 0: no geometry (a plane too large for 32-bit offsets)
    grid_fill_plan(out, nx, lines_per_plane, planes, geo, 0, 0, -1);
    return 0;
}

int vexhip_sell8_grid_check(const vexhip_grid *g, int64_t n)
{
    VEXHIP_REQUIRE(g, "NULL plan");
    VEXHIP_REQUIRE(g->nx >= 8 && n > 0 && n % g->nx == 0 && g->lines_per_plane >= 2 && g->depth >= 1 && g->planes >= 1 && g->segments >= 1
                   && g->segment_rows >= 2 && g->segment_rows <= 1024 && g->segment_rows % 2 == 0 && (long long)g->segments * g->segment_rows >= g->nx
                   && g->threads >= 64 && g->threads <= 512 && g->threads % 64 == 0 && 2 * g->threads >= g->segment_rows
                   && (long long)(g->segments - 1) * g->segment_rows + 2 * g->threads <= g->pitch && g->pitch % 16 == 0
                   && ((n / g->nx + g->lines_per_plane - 1) / g->lines_per_plane) == g->planes
                   && ((long long)g->depth + 4) * g->lines_per_plane * g->nx * 8 < (1ll << 32), "bad grid plan");
    return 0;
}

int vexhip_spmv_sell8v_grid_f64(int dev, void *stream, int64_t n, double alpha, int append, const double *values,
        const double *x, double *y, const vexhip_grid *g)
{
    return grid_apply_axpby(dev, stream, n, alpha, append ? 1 : 0, y, 1.0, values, x, y, g);
}

} // extern "C"

namespace vexhip {
// y = alpha A x + [zm 1: beta zs | zm 2: beta x] through the grid product (spmat.hip vexhip_spmat_apply_axpby_f64)
int grid_apply_axpby(int dev, void *stream, int64_t n, double alpha, int zm, const double *zs, double beta, const double *values,
        const double *x, double *y, const vexhip_grid *g)
{
    VEXHIP_REQUIRE(g && g->usable && g->line_class && g->table && values && x && y, "bad grid product arguments");
    if (int rc = vexhip_sell8_grid_check(g, n)) return rc;
    VEXHIP_REQUIRE(g->x_last + 1 >= n, "bad grid plan");
    VEXHIP_REQUIRE((reinterpret_cast<uintptr_t>(x) & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0, "grid product: x and y must be 8-byte aligned");
    VEXHIP_REQUIRE(zm == 0 || zm == 2 || (zm == 1 && zs && (reinterpret_cast<uintptr_t>(zs) & 7) == 0), "grid product: the addend must be an 8-byte aligned vector");
    VEXHIP_SET_DEVICE(dev);
    grid_dev gd;
    gd.lines = n / g->nx; gd.x_last = g->x_last; gd.n = n;
    gd.nx = g->nx; gd.ny = g->lines_per_plane; gd.nz = g->planes; gd.depth = g->depth;
    gd.segs = g->segments; gd.seg_len = g->segment_rows;
    gd.tiles = (gd.ny + 1) / 2 * gd.segs; gd.tpx = (gd.tiles + 7) / 8; gd.hot = g->hot_class; gd.pitch = g->pitch; gd.flat = g->flat;
    const long long chunks = (gd.nz + gd.depth - 1) / gd.depth;
    gd.cpx = gd.flat ? (int)((chunks + 7) / 8) : 0;
    const long long grid = gd.cpx ? 8ll * gd.cpx * gd.tiles : 8ll * gd.tpx * chunks;
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    const unsigned char *tb = static_cast<const unsigned char *>(g->table);
    hipStream_t s = as_stream(stream);
    const halo_dev none = halo_dev();
#define GRID_LAUNCH(AP, AUX) { if (g->threads > 256) sell8_grid_kernel<AP, AUX, 512><<<(unsigned)grid, (unsigned)g->threads, 0, s>>>(x, y, alpha, zs, beta, g->line_class, tb, values, gd, none); \
                               else sell8_grid_kernel<AP, AUX, 256><<<(unsigned)grid, (unsigned)g->threads, 0, s>>>(x, y, alpha, zs, beta, g->line_class, tb, values, gd, none); }
#define GRID_AUX(AP) switch (g->store_policy) { case 1: GRID_LAUNCH(AP, 18); break; case 2: GRID_LAUNCH(AP, 17); break; case 3: GRID_LAUNCH(AP, 0); break; default: GRID_LAUNCH(AP, 2); }
    if (zm == 1) { GRID_AUX(1) } else if (zm == 2) { GRID_AUX(2) } else { GRID_AUX(0) }
#undef GRID_AUX
#undef GRID_LAUNCH
    VEXHIP_LAUNCH_CHECK();
    return 0;
}


// One device's product step in one launch on a matrix stored by grid line with lines of ANY length (halo.hpp, the pull form): the
// grid product over the planes [H.z0, H.z1) of the stored grid of n_ext rows; x and y are the device's own segments.
int grid_apply_halo(int dev, hipStream_t s, int64_t n_ext, double alpha, int append, const double *values, const double *x, double *y,
        const vexhip_grid *g, halo_dev H)
{
    VEXHIP_REQUIRE(g && g->usable && g->line_class && g->table && values && x && y, "bad grid product arguments");
    if (int rc = vexhip_sell8_grid_check(g, n_ext)) return rc;
    VEXHIP_REQUIRE(H.pull && H.z0 >= 0 && H.z1 > H.z0 && H.z1 <= g->planes && H.step && H.done && H.err, "bad halo step");
    VEXHIP_REQUIRE((long long)H.halo == (long long)g->lines_per_plane * g->nx, "the ghost planes must be planes of the stored grid");
    VEXHIP_REQUIRE((!H.lo || H.z0 >= 1) && (!H.hi || H.z1 < g->planes), "a ghost plane outside the stored grid");
    VEXHIP_REQUIRE((reinterpret_cast<uintptr_t>(x) & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0, "grid product: x and y must be 8-byte aligned");
    VEXHIP_SET_DEVICE(dev);
    grid_dev gd;
    gd.lines = n_ext / g->nx; gd.x_last = g->x_last; gd.n = n_ext;
    gd.nx = g->nx; gd.ny = g->lines_per_plane; gd.nz = g->planes; gd.depth = g->depth;
    gd.segs = g->segments; gd.seg_len = g->segment_rows;
    gd.tiles = (gd.ny + 1) / 2 * gd.segs; gd.tpx = (gd.tiles + 7) / 8; gd.hot = g->hot_class; gd.pitch = g->pitch; gd.flat = g->flat;
    // the walks are the plan's (chosen for the balance of the CUs, grid_geometry_with: shorter ones cost more than the wait they would
    // save -- a strip of 640^3 / 8 in walks of 20 planes took 170 us against 125 us in the plan's, profiles/r06_dist_step_f64_640_first.json)
    const int nzr = H.z1 - H.z0;
    gd.depth = std::min(gd.depth, nzr);
    const long long chunks = (nzr + gd.depth - 1) / gd.depth;
    gd.cpx = gd.flat ? (int)((chunks + 7) / 8) : 0;
    const long long grid = gd.cpx ? 8ll * gd.cpx * gd.tiles : 8ll * gd.tpx * chunks;
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    const long long plane = (long long)gd.ny * gd.nx;
    const double *xe = x - (long long)H.z0 * plane;            // the kernel addresses x and y in the numbering of the stored grid
    double *ye = y - (long long)H.z0 * plane;
    const unsigned char *tb = static_cast<const unsigned char *>(g->table);
#define GRID_HLAUNCH(AP, AUX) { if (g->threads > 256) sell8_grid_kernel<AP, AUX, 512, true><<<(unsigned)grid, (unsigned)g->threads, 0, s>>>(xe, ye, alpha, ye, 1.0, g->line_class, tb, values, gd, H); \
                                else sell8_grid_kernel<AP, AUX, 256, true><<<(unsigned)grid, (unsigned)g->threads, 0, s>>>(xe, ye, alpha, ye, 1.0, g->line_class, tb, values, gd, H); }
#define GRID_HAUX(AP) switch (g->store_policy) { case 1: GRID_HLAUNCH(AP, 18); break; case 2: GRID_HLAUNCH(AP, 17); break; case 3: GRID_HLAUNCH(AP, 0); break; default: GRID_HLAUNCH(AP, 2); }
    if (append) { GRID_HAUX(1) } else { GRID_HAUX(0) }
#undef GRID_HAUX
#undef GRID_HLAUNCH
    VEXHIP_LAUNCH_CHECK();
    return 0;
}
} // namespace vexhip

VEXHIP_WARM_TU(grid)
