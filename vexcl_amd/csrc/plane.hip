// The PLANE product (round 4): y (=|+=) alpha * A * x for value-coded SELL-512 storage with a slice dictionary whose diagonals
// are those of a 7-point operator on a grid with 512-point lines -- {0, +-1, +-512, +-P}, P = 512 * (lines per plane).
// This is the headline kernel of vex::SpMat on the 512^3 Poisson matrix; semantics are the reference's ELL product
// (/root/reference/vexcl/spmat/hybrid_ell.inl:238-269: the row's entries in storage order, products rounded before they are
// added, the scale applied to the sum), results bit-identical to the CSR loop (spmat/csr.inl:163-170).
//
// What it does differently from the march product (sell8.hip), and why (profiles/r04_pm_proto.json):
//   * the march product walks a run of consecutive slices and keeps the +-512 window in an LDS ring; the +-P diagonals are
//     two more 16-byte requests per lane and slice that other workgroups satisfy from HBM a second and third time whenever
//     the three touches of a line do not meet in one L2 (1.21 x the read traffic, L2 hit 58 %), and every x element crosses
//     the LDS (55 % busy, 39 % of it bank conflicts).
//   * here a workgroup owns TWO adjacent grid lines and walks through the PLANES.  Lane t owns rows 2t, 2t + 1 of both lines in
//     every plane, so the +-512 neighbours (the line above / below) and the +-P neighbours (the same line one plane back /
//     ahead) of its rows are 16-byte pairs THE SAME LANE loaded itself: they stay in registers.  Only the +-1 neighbours
//     belong to other lanes: one DPP wave shift each; the element beyond either end of a wave's 128 rows is a scalar load.
//     No LDS, no barrier.  Per plane step a lane requests 4 pairs (two centre lines, the halo line above and below) and stores
//     2: every x line is requested twice instead of three times, and the halo lines are the centre lines of the neighbouring
//     tile, which the same XCD works on at the same plane (tiles are dealt to XCDs in contiguous ranges).
//   * the matrix enters as it does in the march product: 4 bytes per slice (its dictionary block) + the pool of distinct code
//     blocks.  A lane decodes a block into values and validity of its two rows at the seven diagonal positions; the block
//     that most slices use (the plan's `hot` block) keeps them in registers with scalar lane masks, one other block sits
//     next to it and is re-decoded when a line needs a third one (boundary planes / lines: < 1/16 of the slices, the plan
//     checks).  Rows need not be pair-aligned: x comes by diagonal position, not by ELL column.
// The plan (vexhip_sell8_plane_plan) validates every dictionary block on the host -- diagonals in the set, positions ascending
// within a row (so that position order IS storage order) -- and declines otherwise; the march and pair products remain.
// Compiled with -ffp-contract=off.
#include "common.hpp"
#include "lanes.hpp"
#include "halo.hpp"
#include "plane.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace vexhip {
namespace {

// HALO (round 5, halo.hpp): the launch is one rank's whole product step.  x and y are addressed in the numbering of the STORED
// grid, whose plane z0 - 1 / z1 (if the rank has a neighbour there) is a ghost plane: its lines are read from the rank's window
// (H.lo / H.hi) behind the owner's `arrive` flag; the first workgroups of the launch copy the rank's own boundary planes to the
// neighbours.  (`consumed` is raised and the step number advanced by halo_signal_kernel behind this launch: a kernel boundary
// orders every workgroup's reads of the ghost planes before the owners may overwrite them.)
// (the walk of one workgroup; the kernel below adds what follows it in a HALO launch)
// ZM, the addend of the result (round 6): 0 none (y = alpha A x); 1 an array `zs` scaled by beta (y = alpha A x + beta zs: '+=' is zs = y,
// beta = 1; a residual b - A x is zs = b, alpha = -1); 2 beta times x ITSELF, taken from the registers that hold the centre lines (y = x + 2 A x
// moves not a byte more than y = A x).
// FLAT (round 6): no block has an entry at +-512 -- a 5-point operator on a 2-D grid whose rows are virtual 512-point lines (grid.hip
// grid_diagonals): the lines above / below the tile are never requested (two of a step's six 16-byte requests).  A template flag: the
// instantiations of the 3-D operators -- the headline's -- are what they were.
template <int TY, int ZM, int STORE_AUX, bool HALO, bool FLAT = false>
__device__ __forceinline__
void plane_walk(const double *__restrict__ x, double *__restrict__ y, double alpha, const double *__restrict__ zs, double beta,
        const int *__restrict__ blocks, const char *__restrict__ pool, const int *__restrict__ deltas, const double *__restrict__ values,
        const plane_dev &pd, const halo_dev &H, [[maybe_unused]] const unsigned long long step_in)
{
    // LDS: per diagonal code its position (x 2), the value table, and the decoded values of the OTHER block, lane-private
    // ([position * 2 + row][lane]: consecutive lanes, consecutive 8-byte words -- conflict-free; row 14 takes what padding
    // "writes").  33 KiB: four workgroups per CU.
    __shared__ int s_slot[256];
    __shared__ double s_value[256];
    __shared__ double s_other[15][256];

    const int t = threadIdx.x;
    unsigned b = blockIdx.x;
    [[maybe_unused]] unsigned long long step = 0;
    [[maybe_unused]] __shared__ int s_flag[2];
    [[maybe_unused]] unsigned long long dbg_t0 = 0, dbg_flag = 0, dbg_data = 0;
    if constexpr (HALO) {
        if (H.debug) dbg_t0 = wall_clock64();
        step = step_in;
        halo_announce(H, step);
        const unsigned npush = H.pull ? 0u : (H.dst_lo ? (unsigned)H.push_blocks : 0u) + (H.dst_hi ? (unsigned)H.push_blocks : 0u);
        if (b < npush) {
            // ---- copy one of the rank's boundary planes into the neighbour's window (16-byte pieces), raise `arrive` there ----
            const bool down = H.dst_lo && b < (unsigned)H.push_blocks;           // the FIRST plane goes to the lower neighbour
            const unsigned j = down ? b : b - (H.dst_lo ? (unsigned)H.push_blocks : 0u);
            if (t == 0) s_flag[0] = spin_until(down ? H.sent_lo : H.sent_hi, step - 1ull, H.err, H.ticks, 0) ? 1 : 0;   // the neighbour has read the previous share
            __syncthreads();
            if (s_flag[0]) {                                                     // uniform; a neighbour that does not answer is not written to
                const double *src = x + (down ? (long long)H.z0 * pd.far : (long long)(H.z1 - 1) * pd.far);
                double *dst = down ? H.dst_lo : H.dst_hi;
                const int per = ((H.halo + H.push_blocks - 1) / H.push_blocks + 511) / 512 * 512;
                const int i0 = (int)j * per, i1 = i0 + per < H.halo ? i0 + per : H.halo;
                for (int i = i0 + 2 * t; i < i1; i += 512)
                    *reinterpret_cast<d2 *>(dst + i) = *reinterpret_cast<const d2 *>(src + i);
                // The window is UNCACHED memory (here and through the peer mapping): its stores go past every cache, and a wave's
                // stores have been performed at the destination once its store counter is back at zero -- so the flag may follow
                // behind `s_waitcnt vmcnt(0)` + a barrier + a relaxed count of the workgroups, with NO release fence.  A fence at
                // agent or system scope writes back the XCD's L2, which at this moment is full of the product's freshly stored y:
                // with ACQ_REL counts and a system fence in front of the flag the push of 4 MB took ~50 us of a 120 us step
                // (tools/r05_dist_step.py, VEXHIP_HALO_NO_PUSH against the default; profiles/r05_dist_step_*.json).
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (t == 0) {
                    unsigned *cnt = H.done + (down ? 1 : 2);
                    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old + 1u == (unsigned)H.push_blocks) {
                        __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (H.release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");       // a window in cached memory (diagnostic): the model-correct hand-off
                        __hip_atomic_store(down ? H.peer_arrive_lo : H.peer_arrive_hi, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
            if (H.debug && t == 0) { unsigned long long *d = H.debug + 6ull * blockIdx.x; d[0] = dbg_t0; d[1] = 0; d[2] = 0; d[3] = wall_clock64(); d[4] = ~0ull; d[5] = down ? 0ull : 1ull; }
            return;
        }
        b -= npush;
    }
    const unsigned xcd = b & 7u, q = b >> 3;
    const int zc = (int)(q / (unsigned)pd.tpx), tyl = (int)(q - (unsigned)zc * (unsigned)pd.tpx);
    const int tile = (int)xcd * pd.tpx + tyl;
    if (tile >= pd.tiles) return;                                   // the whole workgroup
    const int y0 = TY * tile;
    int z, zend;
    [[maybe_unused]] bool second = false;
    [[maybe_unused]] int z2 = 0, zend2 = 0;
    if constexpr (HALO) {
        // chunks of the planes [z0, z1): the planes next to a ghost plane form SHORT chunks of their own, dispatched behind the
        // main chunks -- they wait for the neighbour's share and read it from uncached memory (slow: few requests in flight)
        // while the main chunks, which never touch a ghost plane, stream the bulk of the strip
        const int mid0 = H.z0 + H.lo_planes, mid1 = H.z1 - H.hi_planes;
        const int nmain = (mid1 - mid0 + pd.depth - 1) / pd.depth;
        if (zc < nmain) { z = mid0 + zc * pd.depth; zend = z + pd.depth < mid1 ? z + pd.depth : mid1; }
        else if (H.hi_planes && zc == nmain) { z = mid1; zend = H.z1; }
        else if (H.lo_two_pass && mid0 - H.z0 >= 2) {
            // the chunk next to the LOWER ghost plane needs that plane for its FIRST plane: it walks the planes above it first and
            // comes back for plane z0 in a second pass, when the neighbour's share has arrived (the push of a plane takes 45 us and
            // more: tools/r05_halo_timeline.py) -- one plane of work behind the flag instead of the whole chunk
            z = H.z0 + 1; zend = mid0; second = true; z2 = H.z0; zend2 = H.z0 + 1;
        } else { z = H.z0; zend = mid0; }
    } else {
        z = zc * pd.depth;
        zend = z + pd.depth < pd.nz ? z + pd.depth : pd.nz;
    }
    if (z >= zend) return;
    [[maybe_unused]] const int z_first0 = z, mid_end = zend;
    if constexpr (HALO) {
        // push_blocks == 0: NO dedicated push workgroups -- the first 2 x 256 product workgroups each copy 1024 elements of a
        // boundary plane before they start their walk.  Stores into the uncached window complete one after the other per wave
        // (32 of them per lane took the 32 dedicated workgroups 45 us on some boxes and 84 us on others: tools/r05_halo_timeline.py);
        // two per lane from 2048 waves at once are done in a few microseconds, and the neighbours' ghost flags rise that early.
        if (!H.pull && H.push_blocks == 0 && b < 512u) {
            const bool down = b < 256u;
            double *dst = down ? H.dst_lo : H.dst_hi;
            if (dst) {                                                           // uniform
                if (t == 0) s_flag[0] = spin_until(down ? H.sent_lo : H.sent_hi, step - 1ull, H.err, H.ticks, 0) ? 1 : 0;
                __syncthreads();
                if (s_flag[0]) {
                    const double *src = x + (down ? (long long)H.z0 * pd.far : (long long)(H.z1 - 1) * pd.far);
                    const int per = ((H.halo + 255) / 256 + 511) / 512 * 512;
                    const int i0 = (int)(b & 255u) * per, i1 = i0 + per < H.halo ? i0 + per : H.halo;
                    for (int i = i0 + 2 * t; i < i1; i += 512)
                        *reinterpret_cast<d2 *>(dst + i) = *reinterpret_cast<const d2 *>(src + i);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (t == 0) {
                        unsigned *cnt = H.done + (down ? 1 : 2);
                        const unsigned npieces = (unsigned)((H.halo + per - 1) / per);
                        const bool mine = i0 < H.halo;
                        if (mine) {
                            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (old + 1u == npieces) {
                                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (H.release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                                __hip_atomic_store(down ? H.peer_arrive_lo : H.peer_arrive_hi, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                        }
                    }
                }
                __syncthreads();                                                 // s_flag is used again by the ghost waits
            }
        }
    }
    const int ny = pd.ny;
    const int nslices = (int)pd.nslices, xlines = (int)pd.xlines;
    const long long x_last = pd.x_last;
    const unsigned lane_b = 16u * (unsigned)t;
    // the element beyond either end of the wave's 128 rows of a line: lane 63 reads the one behind them, every other lane the
    // one in front (lane 0 uses it; one cache line for the rest) -- byte offset from the start of the line
    const int edge_b = (t >> 6) * 1024 + ((t & 63) == 63 ? 1024 : -8);

    s_slot[t] = 2 * position_of(deltas[t], pd.far); s_value[t] = values[t];
    __syncthreads();

    // ---- a dictionary block -> values (into s_other) and validity (returned) of this lane's rows at the seven positions ----
    const int wp = (pd.w + 1) >> 1;
    auto decode = [&](int blk) -> unsigned {
        if (pd.pitch > 0) {                      // the matrix stored by grid line (grid.hip): a value code per position and row, 255 = no entry
            const char *tb = pool + (long long)blk * 7 * pd.pitch + 2 * t;
            unsigned bits = 0;
#pragma unroll
            for (int p = 0; p < 7; ++p) {
                const unsigned c2 = *reinterpret_cast<const unsigned short *>(tb + (long long)p * pd.pitch);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const unsigned code = (c2 >> (8 * r)) & 255u;
                    s_other[2 * p + r][t] = s_value[code];            // entry 255 of the value table is 0.0
                    bits |= (code != 255u ? 1u : 0u) << (2 * p + r);
                }
            }
            return bits;
        }
        const unsigned *cw = reinterpret_cast<const unsigned *>(pool + (long long)blk * ((long long)wp * 2048)) + t;
        unsigned dcw[4], vcw[4];                // diagonal codes, value codes: one word per pair of ELL columns
#pragma unroll
        for (int u = 0; u < 4; ++u) { dcw[u] = u < wp ? cw[u * 256] : 0xffffffffu; vcw[u] = u < wp ? cw[(wp + u) * 256] : 0u; }
#pragma unroll
        for (int p = 0; p < 14; ++p) s_other[p][t] = 0.0;
        unsigned bits = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned cword = dcw[j >> 1] >> (16 * (j & 1)), vword = vcw[j >> 1] >> (16 * (j & 1));
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned code = j < pd.w ? (cword >> (8 * r)) & 255u : 255u;
                const bool real = code < PL_PAD_FIRST;
                const int slot = real ? s_slot[code] + r : 14;
                s_other[slot][t] = s_value[real ? (vword >> (8 * r)) & 255u : 255u];      // entry 255 of the table is 0.0
                bits |= (real ? 1u : 0u) << slot;
            }
        }
        return bits & 0x3fffu;
    };

    const int hot = pd.hot;
    double aH[7][2];                            // the hot block: values ...
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
    int other = -1;                             // what s_other holds now is the hot block's: never asked for

    // clamped requests (prologue, slow steps): line `l` of the tile's window (0 = the line above the tile, 1 .. TY = the tile,
    // TY + 1 = the line below) in plane zz.  A line outside x is never referenced by an entry; what is loaded in its place is
    // multiplied by +0.0 behind a mask
    // HALO: the lines of x and y that exist are those of the planes [z0, z1); the plane below / above them is the ghost plane
    const int line_lo = HALO ? H.z0 * ny : 0, line_hi = HALO ? H.z1 * ny : xlines;       // x: [line_lo, line_hi)
    const int yline_hi = HALO ? H.z1 * ny : nslices;
    [[maybe_unused]] bool got_lo = false, got_hi = false, ghost_bad = false;
    auto line_of = [&](int zz, int l) -> int {
        int li = zz * ny + (y0 - 1 + l);
        li = li < line_lo ? line_lo : li; li = li >= line_hi ? line_hi - 1 : li;
        return li;
    };
    auto ld = [&](int zz, int l) -> d2 {
        if constexpr (HALO) {
            const int li = zz * ny + (y0 - 1 + l);                                   // uniform
            if (li < line_lo || li >= line_hi) {
                const bool below = li < line_lo;
                const int gl = below ? li - (line_lo - ny) : li - line_hi;           // line of the ghost plane
                const double *g = below ? H.lo : H.hi;
                d2 r = {0.0, 0.0};
                if (!g || gl < 0 || gl >= ny || l == 0 || l == TY + 1) return r;     // no neighbour there / not the adjacent plane / the line above or below the tile IN a ghost plane: never referenced by an entry
                bool &got = below ? got_lo : got_hi;
                if (!got && H.pull != 2) {
                    // the first line of this ghost plane the workgroup needs: has the owner's share of THIS product arrived?
                    // (PULL: is the owner's x final?  pull == 2: the host has ordered the streams, there is nothing to wait for)
                    if (t == 0) s_flag[below ? 0 : 1] = spin_until(below ? H.arrive_lo : H.arrive_hi, step, H.err, H.ticks, H.acquire) ? 1 : 0;
                    __syncthreads();
                    if (!s_flag[below ? 0 : 1]) ghost_bad = true;
                    // (no fence per lane: the ghost planes are UNCACHED memory -- nothing of them is ever held in a cache -- and
                    //  lane 0 has acquired at system scope inside spin_until in front of the barrier; a system-scope acquire by
                    //  every lane of every workgroup doubled the time of the whole product: 55 -> 119 us)
                    got = true;
                    if (H.debug && !dbg_flag) dbg_flag = wall_clock64();
                }
                if (ghost_bad) { r.x = r.y = __builtin_nan(""); return r; }         // never numbers from stale ghosts (comm.hip)
                if (H.debug && !dbg_data) {                                          // diagnostics: how long the first ghost line takes to arrive
                    const d2 v = *reinterpret_cast<const d2 *>(reinterpret_cast<const char *>(g + (long long)gl * PL_ROWS) + lane_b);
                    asm volatile("s_waitcnt vmcnt(0)" :: "v"(v.x), "v"(v.y) : "memory");
                    dbg_data = wall_clock64();
                    return v;
                }
                return *reinterpret_cast<const d2 *>(reinterpret_cast<const char *>(g + (long long)gl * PL_ROWS) + lane_b);
            }
        }
        const char *p = reinterpret_cast<const char *>(x + (long long)line_of(zz, l) * PL_ROWS);
        return *reinterpret_cast<const d2 *>(p + lane_b);
    };
    auto edge = [&](int zz, int l) -> double {
        if constexpr (HALO) {
            const int li = zz * ny + (y0 - 1 + l);
            if (li < line_lo || li >= line_hi) return 0.0;                           // the +-1 neighbours inside a ghost line: no row of this rank has them
            long long i = (long long)li * PL_ROWS + (edge_b >> 3);
            const long long a = (long long)line_lo * PL_ROWS, e = (long long)line_hi * PL_ROWS - 1;
            i = i < a ? a : i; i = i > e ? e : i;
            return x[i];
        }
        long long i = (long long)line_of(zz, l) * PL_ROWS + (edge_b >> 3);
        i = i < 0 ? 0 : i; i = i > x_last ? x_last : i;
        return x[i];
    };
    auto yold = [&](int zz, int l) -> d2 {
        int li = zz * ny + (y0 + l);
        li = li < line_lo ? line_lo : li; li = li >= yline_hi ? yline_hi - 1 : li;
        return *reinterpret_cast<const d2 *>(reinterpret_cast<const char *>(zs + (long long)li * PL_ROWS) + lane_b);
    };

    // ---- state at the top of the step for plane z (canonical naming) ----
    // Cs[0..3]: the tile's two centre lines in planes z-1, z, z+1, z+2;  Hs[0..1]: the halo lines (above, below) in planes z, z+1;
    // Es[0..1]: per centre line the edge element of this lane (lane 0: the one in front of the wave's rows, lane 63: the one behind)
    // in planes z, z+1.  A step consumes plane z-1's centres and plane z's halos and edges and requests into the SAME registers
    // what plays that role three (centres) or two planes later: every request has two steps to arrive.  The fast loop runs
    // GROUPS of four steps with the names rotated: after four steps every name is back in place and no register is copied (a
    // copy behind a request makes the step wait for it: the one-step form of this loop ran 8 % slower).  It contains no memory
    // instruction other than these requests and the stores, so that the waits of a step count what the step before requested;
    // which block a line uses is known for up to 64 planes ahead (one look at blocks[] per entry).
    constexpr int NPASS = HALO ? 2 : 1;
    for (int pass = 0; pass < NPASS; ++pass) {
    if constexpr (HALO) if (pass == 1) { if (!second) break; z = z2; zend = zend2; }
    d2 Cs[4][TY], Hs[2][2], Yo[TY];
    double Es[2][TY];
    const unsigned plane_b32 = (unsigned)ny * (PL_ROWS * 8u);          // bytes from a line to the same line of the next plane (the plan: (depth + 4) of them < 2^32)
    const int z_first = z;
    // buffer resources of the fast loop: x from the line above the tile in the workgroup's first plane, y from the tile's first
    // line in that plane (no range check: the fast loop only runs where every request lies inside the arrays)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(x + ((long long)z_first * ny + (y0 - 1)) * PL_ROWS), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y + ((long long)z_first * ny + y0) * PL_ROWS, 0, -1, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(ZM == 1 ? zs : x) + ((long long)z_first * ny + y0) * PL_ROWS, 0, -1, 0x00020000);
#pragma unroll
    for (int l = 0; l < TY; ++l) { Cs[0][l] = ld(z - 1, l + 1); Cs[1][l] = ld(z, l + 1); Cs[2][l] = ld(z + 1, l + 1); Cs[3][l] = ld(z + 2, l + 1); }
    if constexpr (FLAT) { const d2 dzero = {0.0, 0.0}; Hs[0][0] = Hs[0][1] = Hs[1][0] = Hs[1][1] = dzero; }
    else { Hs[0][0] = ld(z, 0); Hs[0][1] = ld(z, TY + 1); Hs[1][0] = ld(z + 1, 0); Hs[1][1] = ld(z + 1, TY + 1); }
#pragma unroll
    for (int l = 0; l < TY; ++l) {
        Es[0][l] = edge(z, l + 1); Es[1][l] = edge(z + 1, l + 1);
        if (ZM == 1) Yo[l] = yold(z, l);
    }

    // the lane's sums for tile line l: x at the seven positions from the registers named above
#define PLANE_XS(P, C, N, H, E, l)                                                                                                   \
        const d2 c = C[l], up = (l) == 0 ? H[0] : C[(l) > 0 ? (l) - 1 : 0], dn = (l) == TY - 1 ? H[1] : C[(l) < TY - 1 ? (l) + 1 : 0];                                                  \
        const double xs0[7] = {P[l].x, up.x, shift_from_lower_lane(c.y, E[l]), c.x, c.y, dn.x, N[l].x};                                \
        const double xs1[7] = {P[l].y, up.y, c.x, c.y, shift_from_upper_lane(c.x, E[l]), dn.y, N[l].y};
#define PLANE_HOT_SUMS(s0, s1)                                                                                                       \
        _Pragma("unroll") for (int p = 0; p < 7; ++p) { s0 += aH[p][0] * keep_lanes(xs0[p], mH[p][0]); s1 += aH[p][1] * keep_lanes(xs1[p], mH[p][1]); }
#define PLANE_OTHER_SUMS(s0, s1)                                                                                                     \
        _Pragma("unroll") for (int p = 0; p < 7; ++p) {                                                                               \
            s0 += s_other[2 * p][t] * keep_bit(xs0[p], bitsO, 2 * p); s1 += s_other[2 * p + 1][t] * keep_bit(xs1[p], bitsO, 2 * p + 1); }

    // fast steps need nothing clamped: planes up to z + 3 inside x, both lines inside y
    int zh = zend;
    {
        const int a = (line_hi - 1 - TY - y0) / ny - 3, bb = (yline_hi - TY - y0) / ny - (ZM == 1 ? 1 : 0);      // largest z with (z+3) ny + y0 + TY <= xlines - 1 / z ny + y0 + TY - 1 <= nslices - 1 ('+=': the old y is requested one plane ahead -- inside y also when x is longer than y)
        if (line_hi - 1 - TY - y0 < 0 || yline_hi - TY - y0 < 0) zh = 0;
        else { zh = zh < a + 1 ? zh : a + 1; zh = zh < bb + 1 ? zh : bb + 1; }
    }

    while (z < zend) {
        // ---- how many of the next planes (<= 64) can take fast steps: both lines use the hot block or the other block ----
        unsigned long long use_hot[TY];          // bit k: line l of plane z + k uses the hot block (else: the other block)
        int run;
        {
            const int k = t & 63, zz = z + k;
            const bool in = zz < zh;
            bool ok = in;
#pragma unroll
            for (int l = 0; l < TY; ++l) {
                const int bk = in ? blocks[(long long)zz * ny + (y0 + l)] : hot;
                ok = ok && (bk == hot || bk == other);
                use_hot[l] = __builtin_amdgcn_ballot_w64(bk == hot);
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
            run = ~m ? __builtin_ctzll(~m) : 64;
        }
        if (run >= 4) {
            // Buffer addressing: a resource per array (scalar), the lane's byte offset (ONE vector register for the whole
            // kernel) and a scalar offset that moves with the planes -- no 64-bit vector address arithmetic, which cost a dozen
            // registers and, re-using the registers of requests still in flight, forced early waits.  xo: plane z + 2, the
            // line above the tile; yo: plane z, the tile's first line; both relative to the workgroup's first plane.
            unsigned xo = (unsigned)((z + 2 - z_first) * plane_b32), yo = (unsigned)((z - z_first) * plane_b32);
            auto fast_step = [&](d2 (&P)[TY], d2 (&C)[TY], d2 (&N)[TY], d2 (&H)[2], double (&E)[TY]) {
                d2 o[TY];
#pragma unroll
                for (int l = 0; l < TY; ++l) {
                    PLANE_XS(P, C, N, H, E, l)
                    double s0 = 0.0, s1 = 0.0;
                    if (use_hot[l] & 1ull) { PLANE_HOT_SUMS(s0, s1) } else { PLANE_OTHER_SUMS(s0, s1) }      // uniform
                    o[l].x = alpha * s0; o[l].y = alpha * s1;
                    if (ZM == 1) { o[l].x = beta * Yo[l].x + o[l].x; o[l].y = beta * Yo[l].y + o[l].y; }
                    if (ZM == 2) { o[l].x = beta * c.x + o[l].x; o[l].y = beta * c.y + o[l].y; }
                }
#pragma unroll
                for (int l = 0; l < TY; ++l) use_hot[l] >>= 1;
#pragma unroll
                for (int l = 0; l < TY; ++l)       // written once, not read again by this kernel
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o[l]), ry, (int)lane_b, (int)(yo + l * 4096u), STORE_AUX);
                if (ZM == 1) {
#pragma unroll
                    for (int l = 0; l < TY; ++l) Yo[l] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rz, (int)lane_b, (int)(yo + plane_b32 + l * 4096u), 0));
                }
                if constexpr (!FLAT) {
                    H[0] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)lane_b, (int)xo, 0));
                    H[1] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)lane_b, (int)(xo + (TY + 1) * 4096u), 0));
                }
#pragma unroll
                for (int l = 0; l < TY; ++l) {
                    P[l] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)lane_b, (int)(xo + plane_b32 + (l + 1) * 4096u), 0));
                    E[l] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx, edge_b + 8, (int)(xo + (l + 1) * 4096u - 8u), 0));
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
        // ---- a slow step: a line needs another block decoded, the last planes (clamped requests), the ragged last plane, what
        // a run leaves over after its groups of four; names rotated by copies ----
#pragma unroll
        for (int l = 0; l < TY; ++l) {
            const int li = z * ny + (y0 + l);
            if (li < yline_hi) {                                          // uniform
                PLANE_XS(Cs[0], Cs[1], Cs[2], Hs[0], Es[0], l)
                double s0 = 0.0, s1 = 0.0;
                const int blk = __builtin_amdgcn_readfirstlane(blocks[li]);
                if (blk == hot) { PLANE_HOT_SUMS(s0, s1) }
                else {
                    if (blk != other) { bitsO = decode(blk); other = blk; }
                    PLANE_OTHER_SUMS(s0, s1)
                }
                d2 o; o.x = alpha * s0; o.y = alpha * s1;
                if (ZM == 1) { o.x = beta * Yo[l].x + o.x; o.y = beta * Yo[l].y + o.y; }
                if (ZM == 2) { o.x = beta * c.x + o.x; o.y = beta * c.y + o.y; }
                __builtin_nontemporal_store(o, reinterpret_cast<d2 *>(reinterpret_cast<char *>(y + (long long)li * PL_ROWS) + lane_b));
            }
        }
#pragma unroll
        for (int l = 0; l < TY; ++l) {
            Cs[0][l] = Cs[1][l]; Cs[1][l] = Cs[2][l]; Cs[2][l] = Cs[3][l]; Cs[3][l] = ld(z + 3, l + 1);
            Es[0][l] = Es[1][l]; Es[1][l] = edge(z + 2, l + 1);
            if (ZM == 1) Yo[l] = yold(z + 1, l);
        }
#pragma unroll
        for (int l = 0; l < 2; ++l) { Hs[0][l] = Hs[1][l]; if constexpr (!FLAT) Hs[1][l] = ld(z + 2, (TY + 1) * l); }
        ++z;
    }
    }       // pass
    if constexpr (HALO) {
        if (H.debug && t == 0) {
            unsigned long long *d = H.debug + 6ull * blockIdx.x;
            d[0] = dbg_t0; d[1] = dbg_flag; d[2] = dbg_data; d[3] = wall_clock64(); d[4] = (unsigned long long)(second ? z2 : z_first0); d[5] = (unsigned long long)(second ? mid_end : zend);
        }
    }
#undef PLANE_XS
#undef PLANE_HOT_SUMS
#undef PLANE_OTHER_SUMS
}

#ifndef VEXHIP_HALO_WAVES
#define VEXHIP_HALO_WAVES 2
#endif
template <int TY, int ZM, int STORE_AUX, bool HALO = false, bool FLAT = false>
__global__ __launch_bounds__(256, (TY == 2 && !HALO) ? 4 : (HALO ? VEXHIP_HALO_WAVES : 2))       // (HALO: a few registers more than 128)
void sell8_plane_kernel(const double *__restrict__ x, double *__restrict__ y, double alpha, const double *__restrict__ zs, double beta,
        const int *__restrict__ blocks, const char *__restrict__ pool, const int *__restrict__ deltas, const double *__restrict__ values,
        plane_dev pd, halo_dev H)
{
    if constexpr (!HALO) {
        plane_walk<TY, ZM, STORE_AUX, false, FLAT>(x, y, alpha, zs, beta, blocks, pool, deltas, values, pd, H, 0ull);
    } else {
        const unsigned long long step = *H.step;
        plane_walk<TY, ZM, STORE_AUX, true>(x, y, alpha, zs, beta, blocks, pool, deltas, values, pd, H, step);
        halo_finish(H, step);          // the launch's last workgroup raises `consumed` and advances the step number (halo.hpp)
    }
}

// The yardstick of the plane / march products (bench.py roofline.device_copy_hand): x copied to y with one 16-byte pair per lane,
// no loop, non-temporal stores -- the same HBM traffic as the product (x once, y once) and nothing else to do.  6.23 TB/s at
// 512^3 elements on the box where the library's copy (torch) reaches 4.94 (profiles/r04_pm_proto.json).
__global__ __launch_bounds__(256)
void stream_copy_kernel(const double *__restrict__ x, double *__restrict__ y, long long npairs, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < npairs) __builtin_nontemporal_store(reinterpret_cast<const d2 *>(x)[i], reinterpret_cast<d2 *>(y) + i);
    else if (i == npairs && (n & 1)) y[n - 1] = x[n - 1];
}

// behind a HALO launch: the ghost planes of product `step` have been read -- the owners may write the next ones
__global__ void halo_signal_kernel(halo_dev H) {
    const unsigned long long step = *H.step;
    if (H.consumed_lo) __hip_atomic_store(H.consumed_lo, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (H.consumed_hi) __hip_atomic_store(H.consumed_hi, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (H.pull == 1) {
        // PULL: the neighbours read this rank's boundary planes of x IN PLACE -- the stream must not go on (to a kernel that may
        // overwrite x) before they say they have (a pushed share is a copy: there the next push waits instead)
        if (H.sent_lo) (void)spin_until(H.sent_lo, step, H.err, H.ticks, 0);
        if (H.sent_hi) (void)spin_until(H.sent_hi, step, H.err, H.ticks, 0);
    }
    *H.step = step + 1ull;
}

} // namespace
} // namespace vexhip

namespace vexhip {
// y = alpha A x + [zm 1: beta zs | zm 2: beta x] through the plane product (spmat.hip vexhip_spmat_apply_axpby_f64; the exported
// vexhip_spmv_sell8v_plane_f64_i32 is zm = append, zs = y, beta = 1)
int plane_apply_axpby(int dev, void *stream, int64_t n, double alpha, int zm, const double *zs, double beta, int64_t w, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const double *values, const double *x, double *y, const vexhip_plane *plane)
{
    VEXHIP_REQUIRE(plane && plane->usable && pool && blocks && deltas && values && x && y, "bad plane product arguments");
    VEXHIP_REQUIRE(n > 0 && n % PL_ROWS == 0 && w >= 1 && w <= 8, "bad plane product geometry");
    VEXHIP_REQUIRE(plane->table_pitch == 0 || plane->table_pitch >= PL_ROWS + 2, "bad plane plan (table pitch)");
    VEXHIP_REQUIRE((plane->tile == 2 || plane->tile == 4) && plane->lines_per_plane >= 4 && plane->lines_per_plane % plane->tile == 0 && plane->depth >= 1 && plane->planes >= 1
                   && (plane->x_last + 1) % PL_ROWS == 0
                   && ((long long)plane->depth + 4) * plane->lines_per_plane * 4096 < (1ll << 32), "bad plane plan");
    VEXHIP_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0, "plane product: x and y must be 16-byte aligned");
    VEXHIP_REQUIRE(zm == 0 || zm == 2 || (zm == 1 && zs && (reinterpret_cast<uintptr_t>(zs) & 15) == 0), "plane product: the addend must be a 16-byte aligned vector");
    VEXHIP_SET_DEVICE(dev);
    plane_dev pd;
    pd.nslices = n / PL_ROWS; pd.xlines = (plane->x_last + 1) / PL_ROWS; pd.x_last = plane->x_last;
    pd.ny = plane->lines_per_plane; pd.nz = plane->planes; pd.depth = plane->depth;
    pd.tiles = pd.ny / plane->tile; pd.tpx = (pd.tiles + 7) / 8; pd.hot = plane->hot_block; pd.w = (int)w; pd.far = pd.ny * PL_ROWS;
    pd.pitch = plane->table_pitch;
    const long long chunks = (pd.nz + pd.depth - 1) / pd.depth;
    const long long grid = 8ll * pd.tpx * chunks;
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    const int store_kind = plane->store_policy;
    const char *cpool = static_cast<const char *>(pool);
    hipStream_t s = as_stream(stream);
    const halo_dev none = halo_dev();
#define PLANE_LAUNCH(TY, ZM, AUX) sell8_plane_kernel<TY, ZM, AUX><<<(unsigned)grid, 256, 0, s>>>(x, y, alpha, zs, beta, blocks, cpool, deltas, values, pd, none)
#define PLANE_AUX(TY, AP) switch (store_kind) { case 1: PLANE_LAUNCH(TY, AP, 18); break; case 2: PLANE_LAUNCH(TY, AP, 17); break; case 3: PLANE_LAUNCH(TY, AP, 0); break; default: PLANE_LAUNCH(TY, AP, 2); }
#define PLANE_FLAT(ZM, AUX) sell8_plane_kernel<2, ZM, AUX, false, true><<<(unsigned)grid, 256, 0, s>>>(x, y, alpha, zs, beta, blocks, cpool, deltas, values, pd, none)
#define PLANE_FAUX(AP) switch (store_kind) { case 1: PLANE_FLAT(AP, 18); break; case 2: PLANE_FLAT(AP, 17); break; case 3: PLANE_FLAT(AP, 0); break; default: PLANE_FLAT(AP, 2); }
    if (plane->tile == 4) { if (zm == 1) { PLANE_AUX(4, 1) } else if (zm == 2) { PLANE_AUX(4, 2) } else { PLANE_AUX(4, 0) } }
    else if (plane->flat) { if (zm == 1) { PLANE_FAUX(1) } else if (zm == 2) { PLANE_FAUX(2) } else { PLANE_FAUX(0) } }
    else { if (zm == 1) { PLANE_AUX(2, 1) } else if (zm == 2) { PLANE_AUX(2, 2) } else { PLANE_AUX(2, 0) } }
#undef PLANE_FAUX
#undef PLANE_FLAT
#undef PLANE_AUX
#undef PLANE_LAUNCH
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

} // namespace vexhip

using namespace vexhip;

// lines per workgroup (2 or 4), planes per workgroup, store policy of a plane plan; false: the walk does not fit 32-bit offsets
static bool plane_geometry_with(long long cus, long long ny, long long nz, int hot, vexhip_plane *out);
static bool plane_geometry(int dev, long long ny, long long nz, int hot, vexhip_plane *out)
{
    return plane_geometry_with(std::max(1, info(dev).cus), ny, nz, hot, out);
}
// (host arithmetic only: vexhip_sell8_plane_geometry exposes it so that the choice of walks can be checked without a device)
static bool plane_geometry_with(long long cus, long long ny, long long nz, int hot, vexhip_plane *out)
{
    cus = std::max(1ll, cus);
    auto depth_for = [&](long long tl) {
        const long long tiles = ny / tl;
        long long chunks = std::max(1ll, std::min(nz / 8, (cus + tiles / 2) / tiles));
        // Round 5 -- ONE workgroup per CU, all of them resident and in step, is the best a launch can do (512 lines per plane:
        // walks of 512 / 256 / 128 / 64 / 43 planes = 0.377 / 0.385 / 0.387 / 0.387 / 0.385 ms) -- but only where the tiles come out
        // as one per CU.  Planes of 640 lines are 1.25 tiles per CU: with one walk per tile a quarter of the CUs do twice the work
        // of the others (0.850 ms for 512 x 640 x 640; walks of 80 planes: 0.621).  So unless the workgroups above number
        // 0.95 .. 1 per CU: many short walks, at least six per CU for the dispatcher to even out, the fewest rounds of
        // (planes per walk + 6) -- the rule of the grid product (grid.hip).  Measured, plan before -> after
        // (profiles/r05_plane_shapes.json): 512 x 640 x 640 0.850 -> 0.621 ms, 512 x 768 x 512 0.768 -> 0.626, 512 x 384 x 768
        // 0.492 -> 0.431, 512 x 320 x 1024 0.702 -> 0.52, 512 x 1024 x 256 0.434 -> 0.428.
        if (tiles * chunks > cus || tiles * chunks * 20 < cus * 19) {
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
        // SHORT walks (a rank's strip of a partitioned grid: 64 planes at 512^3 / 8) want two workgroups per CU: the start and the
        // end of a walk -- four planes requested before the first row is stored, the last planes clamped -- are a tenth of a
        // 64-plane walk and overlap with the other workgroup's steady state: 56.6 -> 47.0 us for 16.8 M rows (depth 22 / 16 / 11:
        // 50.0 / 50.8 / 52.4; tools/r05_dist_step.py, "local part alone").  Long walks lose 2 % that way (512 planes, round 4).
        if (chunks == 1 && nz >= 32 && nz <= 192 && tiles * 2 <= 2 * cus) chunks = 2;
        return (nz + chunks - 1) / chunks;
    };
    long long tile = 2;
    if (const char *e = env(ENV_VEXHIP_PLANE_TILE)) tile = (std::atoi(e) == 4 && ny % 4 == 0) ? 4 : 2;
    long long depth = depth_for(tile);
    if (const char *e = env(ENV_VEXHIP_PLANE_DEPTH)) depth = std::max(1, std::atoi(e));
    depth = std::min(depth, nz);
    while ((depth + 4) * ny * 4096 >= (1ll << 32) && depth > 8) depth = (depth + 1) / 2;      // 32-bit byte offsets inside a workgroup's walk
    if ((depth + 4) * ny * 4096 >= (1ll << 32)) return false;
    out->lines_per_plane = (int32_t)ny; out->planes = (int32_t)nz; out->depth = (int32_t)depth; out->hot_block = hot; out->tile = (int32_t)tile;
    // Cache policy of the y stores: 0 = non-temporal, 1 = non-temporal + sc1, 2 = sc0 sc1 (write-through, the line leaves the L2:
    // more of it is left for the halo lines of x), 3 = plain.  Same sweeps, tile 4 x 256: 0.395 / 0.393 / 0.384 / 0.387 ms;
    // tile 2 x 512: 0.395 / 0.391 / 0.396 / 0.401.  VEXHIP_PLANE_STORE overrides.
    out->store_policy = tile == 4 ? 2 : 1;
    if (const char *e = env(ENV_VEXHIP_PLANE_STORE)) out->store_policy = std::max(0, std::min(3, std::atoi(e)));
    return true;
}

namespace vexhip {
// The plane plan of a matrix stored by grid line (grid.hip grid_build) whose lines are 512 points long: the plane kernel reads the
// class tables instead of SELL-512 code blocks (table_pitch > 0); a line IS a slice, its class takes the place of its block.
// The build has checked what the plan checks on dictionary blocks (diagonals, positions ascending, few lines off the hot class).
int plane_plan_from_grid(int dev, const vexhip_grid *grid, int64_t rows, vexhip_plane *out)
{
    VEXHIP_REQUIRE(grid && out, "NULL argument");
    std::memset(out, 0, sizeof(*out));
    const bool force = env(ENV_VEXHIP_PLANE_FORCE) != nullptr;
    if (!grid->usable || grid->nx != PL_ROWS || grid->segments != 1 || rows % PL_ROWS != 0) return 0;
    const long long ny = grid->lines_per_plane, nz = grid->planes;
    if (ny < 4 || ny % 2 != 0 || (nz < 4 && !force) || (rows / PL_ROWS < 64 && !force)) return 0;
    if ((grid->x_last + 1) % PL_ROWS != 0 || grid->x_last + 1 < rows) return 0;
    if (!plane_geometry(dev, ny, nz, grid->hot_class, out)) return 0;
    out->table_pitch = grid->pitch;
    out->flat = grid->flat;
    out->x_last = grid->x_last; out->usable = 1;
    return 0;
}
} // namespace vexhip

namespace vexhip {
// One rank's product step in one launch (halo.hpp): the plane product over the planes [H.z0, H.z1) of the stored grid of n_ext
// rows, x and y being the rank's own segments (their element 0 is row H.z0 * far of the stored grid); ghost planes from the
// window, boundary planes pushed by the first workgroups; then the kernel that raises `consumed` and advances the step number.
int plane_apply_halo(int dev, hipStream_t s, int64_t n_ext, double alpha, int append, int64_t w, const void *pool, const int32_t *blocks,
        const int32_t *deltas, const double *values, const double *x, double *y, const vexhip_plane *plane, halo_dev H)
{
    VEXHIP_REQUIRE(plane && plane->usable && pool && blocks && deltas && values && x && y, "bad plane product arguments");
    VEXHIP_REQUIRE(n_ext > 0 && n_ext % PL_ROWS == 0 && w >= 1 && w <= 8, "bad plane product geometry");
    VEXHIP_REQUIRE(plane->lines_per_plane >= 4 && plane->lines_per_plane % 2 == 0 && plane->depth >= 1 && plane->planes >= 1
                   && ((long long)plane->depth + 4) * plane->lines_per_plane * 4096 < (1ll << 32), "bad plane plan");
    VEXHIP_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0, "plane product: x and y must be 16-byte aligned");
    VEXHIP_REQUIRE(H.z0 >= 0 && H.z1 > H.z0 && H.z1 <= plane->planes && H.step && H.done && H.err && H.push_blocks >= 0, "bad halo step");
    VEXHIP_REQUIRE(H.halo == plane->lines_per_plane * PL_ROWS, "the ghost planes must be planes of the stored grid");
    VEXHIP_REQUIRE((!H.lo || H.z0 >= 1) && (!H.hi || H.z1 < plane->planes), "a ghost plane outside the stored grid");
    VEXHIP_SET_DEVICE(dev);
    plane_dev pd;
    pd.nslices = n_ext / PL_ROWS; pd.xlines = (plane->x_last + 1) / PL_ROWS; pd.x_last = plane->x_last;
    pd.ny = plane->lines_per_plane; pd.nz = plane->planes; pd.depth = plane->depth;
    pd.tiles = pd.ny / 2; pd.tpx = (pd.tiles + 7) / 8; pd.hot = plane->hot_block; pd.w = (int)w; pd.far = pd.ny * PL_ROWS;
    pd.pitch = plane->table_pitch;
    const int nzr = H.z1 - H.z0;
    int edge_planes = 8;
    if (const char *e = env(ENV_VEXHIP_HALO_EDGE_PLANES)) edge_planes = std::max(1, std::atoi(e));
    int lo_planes = edge_planes, hi_planes = edge_planes;
    if (const char *e = env(ENV_VEXHIP_HALO_LO_PLANES)) lo_planes = std::max(1, std::atoi(e));
    if (const char *e = env(ENV_VEXHIP_HALO_HI_PLANES)) hi_planes = std::max(1, std::atoi(e));
    H.lo_planes = H.lo ? std::min(lo_planes, nzr) : 0;
    H.hi_planes = H.hi ? std::min(hi_planes, nzr - H.lo_planes) : 0;
    const int mid = nzr - H.lo_planes - H.hi_planes;
    pd.depth = std::max(1, mid);          // ONE main chunk (two of 25 planes beside the short chunks: 88-90 us against 76 for the step)
    if (const char *e = env(ENV_VEXHIP_HALO_DEPTH)) pd.depth = std::max(1, std::atoi(e));
    const long long chunks = (H.lo_planes ? 1 : 0) + (H.hi_planes ? 1 : 0) + (mid + pd.depth - 1) / pd.depth;
    const long long npush = H.pull ? 0 : (H.dst_lo ? H.push_blocks : 0) + (H.dst_hi ? H.push_blocks : 0);
    const long long grid = npush + 8ll * pd.tpx * chunks;
    VEXHIP_REQUIRE(H.pull || H.push_blocks > 0 || 8ll * pd.tpx * chunks >= 512 || !(H.dst_lo || H.dst_hi), "too few workgroups to push the boundary planes");
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    // the kernel addresses x and y in the numbering of the stored grid
    const double *xe = x - (long long)H.z0 * pd.far;
    double *ye = y - (long long)H.z0 * pd.far;
    const char *cpool = static_cast<const char *>(pool);
    if (append) sell8_plane_kernel<2, 1, 18, true><<<(unsigned)grid, 256, 0, s>>>(xe, ye, alpha, ye, 1.0, blocks, cpool, deltas, values, pd, H);
    else        sell8_plane_kernel<2, 0, 18, true><<<(unsigned)grid, 256, 0, s>>>(xe, ye, alpha, nullptr, 0.0, blocks, cpool, deltas, values, pd, H);
    VEXHIP_LAUNCH_CHECK();
    if (H.pull != 2 && !H.one_launch) {   // (events: the host advances nothing on the device -- there are no flags to number; one_launch: the last workgroup has done it)
        halo_signal_kernel<<<1, 1, 0, s>>>(H);
        VEXHIP_LAUNCH_CHECK();
    }
    return 0;
}
} // namespace vexhip

extern "C" {

int vexhip_sell8_plane_geometry(int cus, int64_t lines_per_plane, int64_t planes, vexhip_plane *out)
{
    VEXHIP_REQUIRE(out, "NULL output");
    std::memset(out, 0, sizeof(*out));
    reload_env();
    VEXHIP_REQUIRE(lines_per_plane >= 4 && lines_per_plane % 2 == 0 && lines_per_plane < (1ll << 30) && planes >= 1 && planes < (1ll << 30), "bad grid");
    if (!plane_geometry_with(cus, lines_per_plane, planes, 0, out)) { std::memset(out, 0, sizeof(*out)); return 0; }      // depth = 0: no geometry
    return 0;
}

int vexhip_sell8_plane_plan(int dev, void *stream, const int32_t *deltas, int ndeltas, const int32_t *blocks, int64_t nslices,
        const void *pool, int64_t dictionary_blocks, int64_t ell_width, int64_t rows, int64_t tail_nnz, int value_bytes,
        int64_t x_last, vexhip_plane *out)
{
    VEXHIP_REQUIRE(out, "NULL output");
    std::memset(out, 0, sizeof(*out));
    reload_env();
    const bool force = env(ENV_VEXHIP_PLANE_FORCE) != nullptr;          // tests: small grids, many blocks
    if ((value_bytes != 8 && value_bytes != 4) || !deltas || !blocks || !pool || ndeltas < 2 || ndeltas > 7 || dictionary_blocks < 1 || dictionary_blocks > 128) return 0;
    if (ell_width < 1 || ell_width > 8 || tail_nnz != 0 || rows != nslices * PL_ROWS || (nslices < 64 && !force)) return 0;
    if (x_last < 0 || (x_last + 1) % PL_ROWS != 0 || x_last + 1 < rows) return 0;
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    const int wp = (int)((ell_width + 1) / 2);
    const size_t code_bytes = (size_t)wp * 2048;
    std::vector<int> table((size_t)ndeltas), id((size_t)nslices);
    std::vector<unsigned> codes((size_t)dictionary_blocks * code_bytes / 4);
    VEXHIP_TRY(hipMemcpyAsync(table.data(), deltas, sizeof(int) * (size_t)ndeltas, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipMemcpyAsync(id.data(), blocks, sizeof(int) * (size_t)nslices, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipMemcpyAsync(codes.data(), pool, (size_t)dictionary_blocks * code_bytes, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    // the diagonals: {0, +-1, +-512} and one far pair +-P, P a multiple of 1024 (two lines per workgroup)
    long long far = 0;
    for (int d : table) {
        const long long a = std::llabs((long long)d);
        if (a == 0 || a == 1 || a == PL_ROWS) continue;
        if (far == 0) far = a;
        if (a != far) return 0;
    }
    if (far < 4 * PL_ROWS || far % (2 * PL_ROWS) != 0 || far > (1ll << 30)) return 0;
    const long long ny = far / PL_ROWS, nz = (nslices + ny - 1) / ny;
    if (nz < 4 && !force) return 0;
    auto position = [&](int d) { return d == 0 ? 3 : d == -1 ? 2 : d == 1 ? 4 : d == -PL_ROWS ? 1 : d == PL_ROWS ? 5 : d == -far ? 0 : 6; };
    // every row of every dictionary block: positions strictly ascending along the ELL columns (position order = storage order)
    for (int64_t blk = 0; blk < dictionary_blocks; ++blk) {
        const unsigned *cw = codes.data() + (size_t)blk * code_bytes / 4;
        for (int t = 0; t < 256; ++t)
            for (int r = 0; r < 2; ++r) {
                int last = -1;
                for (int j = 0; j < (int)ell_width; ++j) {
                    const unsigned cword = cw[(size_t)(j >> 1) * 256 + t] >> (16 * (j & 1));
                    const unsigned code = (cword >> (8 * r)) & 255u;
                    if (code >= PL_PAD_FIRST) continue;
                    if ((int)code >= ndeltas) return 0;
                    const int p = position(table[code]);
                    if (p <= last) return 0;
                    last = p;
                }
            }
    }
    // the hot block; lines that use another one must be few (each change of the other block is a decode: two dependent loads)
    std::vector<int64_t> uses((size_t)dictionary_blocks, 0);
    for (int64_t k = 0; k < nslices; ++k) {
        if (id[(size_t)k] < 0 || id[(size_t)k] >= dictionary_blocks) return 0;
        ++uses[(size_t)id[(size_t)k]];
    }
    const int hot = (int)(std::max_element(uses.begin(), uses.end()) - uses.begin());
    if ((nslices - uses[(size_t)hot]) * 16 > nslices && !force) return 0;
    // Lines per workgroup (2 or 4) and planes per workgroup.  Measured at 512^3 (profiles/r04_plane_sweep*.json, r04_plane_offsets.json;
    // march product 0.466 ms beside them): FEW, LONG workgroups win -- with every request two steps ahead a wave hides the memory
    // latency by itself, and every workgroup re-reads the 4 planes around its range: tile 2 x depth 128 / 256 / 512 = 0.393 /
    // 0.389 / 0.381 ms, tile 4 x depth 128 / 256 = 0.409 / 0.388 ms (tile 4 x 512: half the CUs idle, 0.57).  Tile 4 halves the
    // halo requests (HBM read 1.04 x instead of 1.14 x of x at depth 128) but needs 164 registers and is not faster.  So: two
    // lines, about one workgroup per CU.  The time also depends on where x and y lie relative to each other (0.378 - 0.41 ms for
    // the same kernel, r04_plane_offsets.json; the copy kernel and the march product do not show it).
    if (!plane_geometry(dev, ny, nz, hot, out)) return 0;
    out->flat = 1;
    for (int d : table) if (std::llabs((long long)d) == PL_ROWS) out->flat = 0;
    out->x_last = x_last; out->usable = 1;
    return 0;
}

int vexhip_spmv_sell8v_plane_f64_i32(int dev, void *stream, int64_t n, double alpha, int append, int64_t w, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const double *values, const double *x, double *y, const vexhip_plane *plane)
{
    return plane_apply_axpby(dev, stream, n, alpha, append ? 1 : 0, y, 1.0, w, pool, blocks, deltas, values, x, y, plane);
}

int vexhip_stream_copy_f64(int dev, void *stream, const double *x, double *y, int64_t n)
{
    VEXHIP_REQUIRE(n >= 0 && (n == 0 || (x && y)), "bad copy arguments");
    VEXHIP_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0, "stream copy: x and y must be 16-byte aligned");
    if (n == 0) return 0;
    VEXHIP_SET_DEVICE(dev);
    const long long npairs = n / 2, grid = (npairs + 1 + 255) / 256;
    VEXHIP_REQUIRE(grid < (1ll << 31), "vector too large for one launch");
    stream_copy_kernel<<<(unsigned)grid, 256, 0, as_stream(stream)>>>(x, y, npairs, (long long)n);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

} // extern "C"

VEXHIP_WARM_TU(plane)
