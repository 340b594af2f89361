// The PLANE product for fp32 (round 5): y (=|+=) alpha * A * x for value-coded SELL-512 storage with a slice dictionary whose
// diagonals are those of a 7-point operator on a grid with 512-point lines -- the storage, the plan and the walk of plane.hip,
// with FOUR rows per lane: a 512-point line of floats is 2 KB = 128 lanes x 16 bytes, so a workgroup is two waves, owns two
// adjacent grid lines and walks through the planes.  Every request is 16 bytes per lane as in the fp64 kernel (the march
// product, which float matrices took until now, moves x in 8-byte pairs through the LDS: 0.289 ms at 512^3, 0.46 of the HBM
// peak by the bytes that must move -- profiles/r05_fp32.json).  The +-512 / +-P neighbours of a lane's rows are quads the lane
// loaded itself; of the +-1 neighbours only the first and the last of the four rows look at another lane (one DPP shift
// each), the element beyond either end of a wave's 256 rows is a 4-byte load of lane 0 / lane 63.
// Semantics: the reference's ELL product (/root/reference/vexcl/spmat/hybrid_ell.inl:238-269: the row's entries in storage
// order, products rounded before they are added, the scale applied to the sum); results bit-identical to the CSR loop
// (spmat/csr.inl:163-170) in fp32.  Compiled with -ffp-contract=off.
#include "common.hpp"
#include "halo.hpp"
#include "lanes.hpp"
#include "plane.hpp"

#include <algorithm>
#include <cstdlib>

namespace vexhip {
namespace {

constexpr int P32_LANES = 128;             // lanes of a workgroup: 4 rows each
constexpr int P32_LINE_B = PL_ROWS * 4;    // bytes of a line

// HALO (round 6; halo.hpp, the PULL form only): the launch is one device's whole product step, as in grid.hip -- x and y addressed in
// the numbering of the STORED grid, the plane below z0 / above z1 - 1 read from the neighbour's x in place (H.lo / H.hi point at
// floats here), the two walks that touch a ghost plane dispatched last.
// ZM: the addend of the result (plane.hip): 0 none, 1 beta times the array zs ('+=': zs = y, beta = 1), 2 beta times x itself (from the registers)
template <int ZM, int STORE_AUX, bool HALO>
__device__ __forceinline__
void plane32_walk(const float *__restrict__ x, float *__restrict__ y, float alpha, const float *__restrict__ zs, float beta,
        const int *__restrict__ blocks, const char *__restrict__ pool, const int *__restrict__ deltas, const float *__restrict__ values,
        const plane_dev &pd, const halo_dev &H, [[maybe_unused]] const unsigned long long step)
{
    constexpr int TY = 2;
    // LDS: per diagonal code its position (x 4), the value table, and the decoded values of the OTHER block, lane-private
    // ([position * 4 + row][lane]; row 28 takes what padding "writes")
    __shared__ int s_slot[256];
    __shared__ float s_value[256];
    __shared__ float s_other[29][P32_LANES];

    const int t = threadIdx.x;
    const unsigned b = blockIdx.x;
    const unsigned xcd = b & 7u, q = b >> 3;
    const int zc = (int)(q / (unsigned)pd.tpx), tyl = (int)(q - (unsigned)zc * (unsigned)pd.tpx);
    const int tile = (int)xcd * pd.tpx + tyl;
    if (tile >= pd.tiles) return;                                   // the whole workgroup
    const int y0 = TY * tile;
    int z, zend_;
    [[maybe_unused]] bool ghost_bad = false;
    [[maybe_unused]] __shared__ int s_flag;
    if constexpr (HALO) {
        const int nch = (H.z1 - H.z0 + pd.depth - 1) / pd.depth;
        const int c = nch > 1 ? (zc + 1) % nch : 0;               // the first and the last walk touch a ghost plane: dispatched behind the others
        z = H.z0 + c * pd.depth; zend_ = z + pd.depth < H.z1 ? z + pd.depth : H.z1;
        if (zc >= nch || z >= zend_) return;
        ghost_bad = !halo_wait(H, step, H.lo && z == H.z0, H.hi && zend_ == H.z1, &s_flag);
    } else {
        z = zc * pd.depth;
        zend_ = z + pd.depth < pd.nz ? z + pd.depth : pd.nz;
        if (z >= zend_) return;
    }
    const int zend = zend_;
    const int ny = pd.ny;
    // HALO: the lines of x and y that exist are those of the planes [z0, z1)
    const int line_lo = HALO ? H.z0 * ny : 0;
    const int nslices = HALO ? H.z1 * ny : (int)pd.nslices, xlines = HALO ? H.z1 * ny : (int)pd.xlines;
    const long long x_last = HALO ? (long long)H.z1 * ny * PL_ROWS - 1 : pd.x_last;
    const unsigned lane_b = 16u * (unsigned)t;
    // the element beyond either end of the wave's 256 rows of a line: lane 63 reads the one behind them, every other lane the
    // one in front (lane 0 uses it) -- byte offset from the start of the line
    const int edge_b = (t >> 6) * 1024 + ((t & 63) == 63 ? 1024 : -4);

    for (int i = t; i < 256; i += P32_LANES) { s_slot[i] = 4 * position_of(deltas[i], pd.far); s_value[i] = values[i]; }
    __syncthreads();

    // ---- a dictionary block -> values (into s_other) and validity (returned) of this lane's four rows at the seven positions ----
    const int wp = (pd.w + 1) >> 1;
    auto decode = [&](int blk) -> unsigned {
        if (pd.pitch > 0) {                      // the matrix stored by grid line (grid.hip): a value code per position and row, 255 = no entry
            const char *tb = pool + (long long)blk * 7 * pd.pitch + 4 * t;
            unsigned bits = 0;
#pragma unroll
            for (int p = 0; p < 7; ++p) {
                const unsigned c4 = *reinterpret_cast<const unsigned *>(tb + (long long)p * pd.pitch);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned code = (c4 >> (8 * r)) & 255u;
                    s_other[4 * p + r][t] = s_value[code];            // entry 255 of the value table is 0.0
                    bits |= (code != 255u ? 1u : 0u) << (4 * p + r);
                }
            }
            return bits;
        }
        // a code block: per pair of ELL columns 256 words of diagonal codes, then as many of value codes; word i holds rows 2i, 2i + 1
        const unsigned *cw = reinterpret_cast<const unsigned *>(pool + (long long)blk * ((long long)wp * 2048)) + 2 * t;
        unsigned dcw[4][2], vcw[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) { dcw[u][h] = u < wp ? cw[u * 256 + h] : 0xffffffffu; vcw[u][h] = u < wp ? cw[(wp + u) * 256 + h] : 0u; }
#pragma unroll
        for (int p = 0; p < 28; ++p) s_other[p][t] = 0.0f;
        unsigned bits = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned cword = dcw[j >> 1][h] >> (16 * (j & 1)), vword = vcw[j >> 1][h] >> (16 * (j & 1));
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    const int r = 2 * h + r2;
                    const unsigned code = j < pd.w ? (cword >> (8 * r2)) & 255u : 255u;
                    const bool real = code < PL_PAD_FIRST;
                    const int slot = real ? s_slot[code] + r : 28;
                    s_other[slot][t] = s_value[real ? (vword >> (8 * r2)) & 255u : 255u];      // entry 255 of the table is 0.0
                    bits |= (real ? 1u : 0u) << slot;
                }
            }
        return bits & 0x0fffffffu;
    };

    const int hot = pd.hot;
    float aH[7][4];                             // the hot block: values ...
    unsigned long long mH[7][4];                // ... and the lanes with an entry, per position and row.  (56 scalar registers: the
    // compiler keeps some of them in lanes of vector registers, 40 v_readlane per line.  Measured at 512^3: these masks 0.2066 ms,
    // the lane's own 28 validity bits in a vector register 0.2085, NO masks at all 0.2035 -- the product is not bound by them.)
    {
        const unsigned bitsH = decode(hot);
#pragma unroll
        for (int p = 0; p < 7; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                aH[p][r] = s_other[4 * p + r][t];
                mH[p][r] = __builtin_amdgcn_ballot_w64((bitsH >> (4 * p + r)) & 1u);
            }
    }
    unsigned bitsO = 0;
    int other = -1;                             // what s_other holds now is the hot block's: never asked for

    // clamped requests (prologue, slow steps): line `l` of the tile's window (0 = the line above the tile, 1 .. TY = the tile,
    // TY + 1 = the line below) in plane zz.  A line outside x is never referenced by an entry; what is loaded in its place is
    // multiplied by +0.0 behind a mask
    auto line_of = [&](int zz, int l) -> int {
        int li = zz * ny + (y0 - 1 + l);
        li = li < line_lo ? line_lo : li; li = li >= xlines ? xlines - 1 : li;
        return li;
    };
    auto ld = [&](int zz, int l) -> f4 {
        if constexpr (HALO) {
            const int li = zz * ny + (y0 - 1 + l);                                   // uniform
            if (li < line_lo || li >= xlines) {
                const bool below = li < line_lo;
                const int gl = below ? li - (line_lo - ny) : li - xlines;            // line of the ghost plane
                const float *g = reinterpret_cast<const float *>(below ? H.lo : H.hi);
                f4 r = {0.0f, 0.0f, 0.0f, 0.0f};
                if (!g || gl < 0 || gl >= ny || l == 0 || l == TY + 1) return r;     // no neighbour there / not the adjacent plane / the line above or below the tile IN a ghost plane: never referenced
                if (ghost_bad) { r.x = r.y = r.z = r.w = __builtin_nanf(""); return r; }
                return *reinterpret_cast<const f4u *>(reinterpret_cast<const char *>(g + (long long)gl * PL_ROWS) + lane_b);
            }
        }
        const char *p = reinterpret_cast<const char *>(x + (long long)line_of(zz, l) * PL_ROWS);
        return *reinterpret_cast<const f4u *>(p + lane_b);
    };
    auto edge = [&](int zz, int l) -> float {
        if constexpr (HALO) {
            const int li = zz * ny + (y0 - 1 + l);
            if (li < line_lo || li >= xlines) return 0.0f;                           // the +-1 neighbours inside a ghost line: no row of this device has them
        }
        long long i = (long long)line_of(zz, l) * PL_ROWS + (edge_b >> 2);
        const long long a = (long long)line_lo * PL_ROWS;
        i = i < a ? a : i; i = i > x_last ? x_last : i;
        return x[i];
    };
    auto yold = [&](int zz, int l) -> f4 {
        int li = zz * ny + (y0 + l);
        li = li < line_lo ? line_lo : li; li = li >= nslices ? nslices - 1 : li;
        return *reinterpret_cast<const f4u *>(reinterpret_cast<const char *>(zs + (long long)li * PL_ROWS) + lane_b);
    };

    // ---- state at the top of the step for plane z: as in plane.hip ----
    // Cs[0..3]: the tile's two centre lines in planes z-1, z, z+1, z+2;  Hs[0..1]: the halo lines (above, below) in planes z, z+1;
    // Es[0..1]: per centre line the edge element of this lane in planes z, z+1.  A step consumes plane z-1's centres and plane
    // z's halos and edges and requests into the SAME registers what plays that role three (centres) or two planes later; the
    // fast loop runs groups of four steps with the names rotated.
    f4 Cs[4][TY], Hs[2][2], Yo[TY];
    float Es[2][TY];
    const unsigned plane_b32 = (unsigned)ny * (unsigned)P32_LINE_B;    // bytes from a line to the same line of the next plane
    const int z_first = z;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x + ((long long)z_first * ny + (y0 - 1)) * PL_ROWS), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y + ((long long)z_first * ny + y0) * PL_ROWS, 0, -1, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ZM == 1 ? zs : x) + ((long long)z_first * ny + y0) * PL_ROWS, 0, -1, 0x00020000);
#pragma unroll
    for (int l = 0; l < TY; ++l) { Cs[0][l] = ld(z - 1, l + 1); Cs[1][l] = ld(z, l + 1); Cs[2][l] = ld(z + 1, l + 1); Cs[3][l] = ld(z + 2, l + 1); }
    Hs[0][0] = ld(z, 0); Hs[0][1] = ld(z, TY + 1); Hs[1][0] = ld(z + 1, 0); Hs[1][1] = ld(z + 1, TY + 1);
#pragma unroll
    for (int l = 0; l < TY; ++l) {
        Es[0][l] = edge(z, l + 1); Es[1][l] = edge(z + 1, l + 1);
        if (ZM == 1) Yo[l] = yold(z, l);
    }

    // x at the seven positions {-far, -512, -1, 0, +1, +512, +far} of the lane's four rows of tile line l
#define P32_XS(P, C, N, H, E, l)                                                                                                     \
        const f4 c = C[l], up = (l) == 0 ? H[0] : C[0], dn = (l) == TY - 1 ? H[1] : C[TY - 1], pv = P[l], nv = N[l];                     \
        const float xs[4][7] = {{pv.x, up.x, shift_from_lower_lane(c.w, E[l]), c.x, c.y, dn.x, nv.x},                                 \
                                {pv.y, up.y, c.x, c.y, c.z, dn.y, nv.y},                                                              \
                                {pv.z, up.z, c.y, c.z, c.w, dn.z, nv.z},                                                              \
                                {pv.w, up.w, c.z, c.w, shift_from_upper_lane(c.x, E[l]), dn.w, nv.w}};
#define P32_HOT_SUMS(s)                                                                                                              \
        _Pragma("unroll") for (int p = 0; p < 7; ++p) _Pragma("unroll") for (int r = 0; r < 4; ++r) s[r] += aH[p][r] * keep_lanes(xs[r][p], mH[p][r]);
#define P32_OTHER_SUMS(s)                                                                                                            \
        _Pragma("unroll") for (int p = 0; p < 7; ++p) _Pragma("unroll") for (int r = 0; r < 4; ++r) s[r] += s_other[4 * p + r][t] * keep_bit(xs[r][p], bitsO, 4 * p + r);

    // fast steps need nothing clamped: planes up to z + 3 inside x, both lines inside y
    int zh = zend;
    {
        const int a = (xlines - 1 - TY - y0) / ny - 3, bb = (nslices - TY - y0) / ny - (ZM == 1 ? 1 : 0);
        if (xlines - 1 - TY - y0 < 0 || nslices - TY - y0 < 0) zh = 0;
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
            // xo: plane z + 2, the line above the tile; yo: plane z, the tile's first line; both relative to the workgroup's first plane
            unsigned xo = (unsigned)((z + 2 - z_first) * plane_b32), yo = (unsigned)((z - z_first) * plane_b32);
            auto fast_step = [&](f4 (&P)[TY], f4 (&C)[TY], f4 (&N)[TY], f4 (&H)[2], float (&E)[TY]) {
                f4 o[TY];
#pragma unroll
                for (int l = 0; l < TY; ++l) {
                    P32_XS(P, C, N, H, E, l)
                    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (use_hot[l] & 1ull) { P32_HOT_SUMS(s) } else { P32_OTHER_SUMS(s) }      // uniform
                    o[l].x = alpha * s[0]; o[l].y = alpha * s[1]; o[l].z = alpha * s[2]; o[l].w = alpha * s[3];
                    if (ZM == 1) o[l] = beta * Yo[l] + o[l];
                    if (ZM == 2) o[l] = beta * c + o[l];
                }
#pragma unroll
                for (int l = 0; l < TY; ++l) use_hot[l] >>= 1;
#pragma unroll
                for (int l = 0; l < TY; ++l)       // written once, not read again by this kernel
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o[l]), ry, (int)lane_b, (int)(yo + l * (unsigned)P32_LINE_B), STORE_AUX);
                if (ZM == 1) {
#pragma unroll
                    for (int l = 0; l < TY; ++l) Yo[l] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rz, (int)lane_b, (int)(yo + plane_b32 + l * (unsigned)P32_LINE_B), 0));
                }
                H[0] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)lane_b, (int)xo, 0));
                H[1] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)lane_b, (int)(xo + (TY + 1) * (unsigned)P32_LINE_B), 0));
#pragma unroll
                for (int l = 0; l < TY; ++l) {
                    P[l] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)lane_b, (int)(xo + plane_b32 + (l + 1) * (unsigned)P32_LINE_B), 0));
                    E[l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, edge_b + 4, (int)(xo + (l + 1) * (unsigned)P32_LINE_B - 4u), 0));
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
            if (li < nslices) {                                           // uniform
                P32_XS(Cs[0], Cs[1], Cs[2], Hs[0], Es[0], l)
                float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                const int blk = __builtin_amdgcn_readfirstlane(blocks[li]);
                if (blk == hot) { P32_HOT_SUMS(s) }
                else {
                    if (blk != other) { bitsO = decode(blk); other = blk; }
                    P32_OTHER_SUMS(s)
                }
                f4 o; o.x = alpha * s[0]; o.y = alpha * s[1]; o.z = alpha * s[2]; o.w = alpha * s[3];
                if (ZM == 1) o = beta * Yo[l] + o;
                if (ZM == 2) o = beta * c + o;
                *reinterpret_cast<f4u *>(reinterpret_cast<char *>(y + (long long)li * PL_ROWS) + lane_b) = o;
            }
        }
#pragma unroll
        for (int l = 0; l < TY; ++l) {
            Cs[0][l] = Cs[1][l]; Cs[1][l] = Cs[2][l]; Cs[2][l] = Cs[3][l]; Cs[3][l] = ld(z + 3, l + 1);
            Es[0][l] = Es[1][l]; Es[1][l] = edge(z + 2, l + 1);
            if (ZM == 1) Yo[l] = yold(z + 1, l);
        }
#pragma unroll
        for (int l = 0; l < 2; ++l) { Hs[0][l] = Hs[1][l]; Hs[1][l] = ld(z + 2, (TY + 1) * l); }
        ++z;
    }
#undef P32_XS
#undef P32_HOT_SUMS
#undef P32_OTHER_SUMS
}

template <int ZM, int STORE_AUX, bool HALO = false>
__global__ __launch_bounds__(P32_LANES)
void sell8_plane_f32_kernel(const float *__restrict__ x, float *__restrict__ y, float alpha, const float *__restrict__ zs, float beta,
        const int *__restrict__ blocks, const char *__restrict__ pool, const int *__restrict__ deltas, const float *__restrict__ values,
        plane_dev pd, halo_dev H)
{
    if constexpr (!HALO) {
        plane32_walk<ZM, STORE_AUX, false>(x, y, alpha, zs, beta, blocks, pool, deltas, values, pd, H, 0ull);
    } else {
        const unsigned long long step = *H.step;
        halo_announce(H, step);
        plane32_walk<ZM, STORE_AUX, true>(x, y, alpha, zs, beta, blocks, pool, deltas, values, pd, H, step);
        halo_finish(H, step);
    }
}

} // namespace
} // namespace vexhip

using namespace vexhip;

namespace vexhip {
int plane32_apply_axpby(int dev, void *stream, int64_t n, float alpha, int zm, const float *zs, float beta, int64_t w, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const float *values, const float *x, float *y, const vexhip_plane *plane);
}

extern "C" {

// Planes per workgroup of the fp32 plane product (host arithmetic only, exported so that the choice can be checked without a device):
// a workgroup is two waves and six of them fit a CU: about two rounds of workgroups (512^3: depth 512 / 256 / 128 / 64 / 43 / 32 / 16 =
// 0.315 / 0.239 / 0.214 / 0.213 / 0.207 / 0.214 / 0.221 ms, profiles/r05_fp32_plane.txt); no walk shorter than 8 planes.  The kernel
// forms byte offsets inside a walk in 32 bits: the depth is halved until (depth + 4) planes of 2048-byte lines fit -- checked on the
// depth chosen HERE, not on the fp64 plan's (round 6: ny = 8192, nz = 256 passed the fp64 check at depth 64 and walked 256 planes).
// 0: no depth fits.
int64_t vexhip_sell8_plane_f32_depth(int cus, int64_t lines_per_plane, int64_t planes)
{
    if (lines_per_plane < 2 || planes < 1) return 0;
    const long long c = std::max(1, cus), tiles = lines_per_plane / 2;
    const long long chunks = std::max(1ll, std::min<long long>(planes / 8, (12 * c + tiles / 2) / tiles));
    long long depth = (planes + chunks - 1) / chunks;
    if (const char *e = env(ENV_VEXHIP_PLANE32_DEPTH)) if (std::atoi(e) > 0) depth = std::min<long long>(std::atoi(e), planes);
    while ((depth + 4) * lines_per_plane * P32_LINE_B >= (1ll << 32) && depth > 8) depth = (depth + 1) / 2;
    return (depth + 4) * lines_per_plane * P32_LINE_B < (1ll << 32) ? depth : 0;
}

int vexhip_spmv_sell8v_plane_f32_i32(int dev, void *stream, int64_t n, float alpha, int append, int64_t w, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const float *values, const float *x, float *y, const vexhip_plane *plane)
{
    return plane32_apply_axpby(dev, stream, n, alpha, append ? 1 : 0, y, 1.0f, w, pool, blocks, deltas, values, x, y, plane);
}

} // extern "C"

namespace vexhip {
// y = alpha A x + [zm 1: beta zs | zm 2: beta x] through the fp32 plane product (spmat.hip vexhip_spmat_apply_axpby_f32)
int plane32_apply_axpby(int dev, void *stream, int64_t n, float alpha, int zm, const float *zs, float beta, int64_t w, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const float *values, const float *x, float *y, const vexhip_plane *plane)
{
    VEXHIP_REQUIRE(plane && plane->usable && pool && blocks && deltas && values && x && y, "bad plane product arguments");
    VEXHIP_REQUIRE(n > 0 && n % PL_ROWS == 0 && w >= 1 && w <= 8, "bad plane product geometry");
    VEXHIP_REQUIRE(plane->table_pitch == 0 || plane->table_pitch >= PL_ROWS + 2, "bad plane plan (table pitch)");
    VEXHIP_REQUIRE(plane->lines_per_plane >= 4 && plane->lines_per_plane % 2 == 0 && plane->depth >= 1 && plane->planes >= 1
                   && (plane->x_last + 1) % PL_ROWS == 0, "bad plane plan");
    // (x and y may start at any element: 16-byte requests at 4-byte addresses are served, a matrix stored by grid line has no
    //  other product to fall back on)
    VEXHIP_SET_DEVICE(dev);
    plane_dev pd;
    pd.nslices = n / PL_ROWS; pd.xlines = (plane->x_last + 1) / PL_ROWS; pd.x_last = plane->x_last;
    pd.ny = plane->lines_per_plane; pd.nz = plane->planes;
    pd.tiles = pd.ny / 2;
    pd.depth = (int)vexhip_sell8_plane_f32_depth(std::max(1, info(dev).cus), pd.ny, pd.nz);
    VEXHIP_REQUIRE(pd.depth > 0, "fp32 plane product: a walk of this grid does not fit 32-bit byte offsets");
    pd.tpx = (pd.tiles + 7) / 8; pd.hot = plane->hot_block; pd.w = (int)w; pd.far = pd.ny * PL_ROWS;
    pd.pitch = plane->table_pitch;
    const long long chunks = (pd.nz + pd.depth - 1) / pd.depth;
    const long long grid = 8ll * pd.tpx * chunks;
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    const char *cpool = static_cast<const char *>(pool);
    hipStream_t s = as_stream(stream);
    int store_kind = 1;
    if (const char *e = env(ENV_VEXHIP_PLANE_STORE)) store_kind = std::max(0, std::min(3, std::atoi(e)));
    const halo_dev none = halo_dev();
#define P32_LAUNCH(AP, AUX) sell8_plane_f32_kernel<AP, AUX><<<(unsigned)grid, P32_LANES, 0, s>>>(x, y, alpha, zs, beta, blocks, cpool, deltas, values, pd, none)
#define P32_AUX(AP) switch (store_kind) { case 1: P32_LAUNCH(AP, 18); break; case 2: P32_LAUNCH(AP, 17); break; case 3: P32_LAUNCH(AP, 0); break; default: P32_LAUNCH(AP, 2); }
    VEXHIP_REQUIRE(zm == 0 || zm == 2 || (zm == 1 && zs), "plane product: the addend must be a vector");
    if (zm == 1) { P32_AUX(1) } else if (zm == 2) { P32_AUX(2) } else { P32_AUX(0) }
#undef P32_AUX
#undef P32_LAUNCH
    VEXHIP_LAUNCH_CHECK();
    return 0;
}


// One device's product step in one launch for a float matrix on 512-point lines (halo.hpp, the pull form): the fp32 plane product over
// the planes [H.z0, H.z1) of the stored grid of n_ext rows; x and y are the device's own segments.
int plane32_apply_halo(int dev, hipStream_t s, int64_t n_ext, float alpha, int append, int64_t w, const void *pool, const int32_t *blocks,
        const int32_t *deltas, const float *values, const float *x, float *y, const vexhip_plane *plane, halo_dev H)
{
    VEXHIP_REQUIRE(plane && plane->usable && pool && blocks && deltas && values && x && y, "bad plane product arguments");
    VEXHIP_REQUIRE(n_ext > 0 && n_ext % PL_ROWS == 0 && w >= 1 && w <= 8, "bad plane product geometry");
    VEXHIP_REQUIRE(plane->lines_per_plane >= 4 && plane->lines_per_plane % 2 == 0 && plane->planes >= 1, "bad plane plan");
    VEXHIP_REQUIRE(H.pull && H.z0 >= 0 && H.z1 > H.z0 && H.z1 <= plane->planes && H.step && H.done && H.err, "bad halo step");
    VEXHIP_REQUIRE(H.halo == plane->lines_per_plane * PL_ROWS, "the ghost planes must be planes of the stored grid");
    VEXHIP_REQUIRE((!H.lo || H.z0 >= 1) && (!H.hi || H.z1 < plane->planes), "a ghost plane outside the stored grid");
    VEXHIP_SET_DEVICE(dev);
    plane_dev pd;
    pd.nslices = n_ext / PL_ROWS; pd.xlines = (plane->x_last + 1) / PL_ROWS; pd.x_last = plane->x_last;
    pd.ny = plane->lines_per_plane; pd.nz = plane->planes;
    pd.tiles = pd.ny / 2;
    const int nzr = H.z1 - H.z0;
    pd.depth = (int)vexhip_sell8_plane_f32_depth(std::max(1, info(dev).cus), pd.ny, nzr);
    VEXHIP_REQUIRE(pd.depth > 0, "fp32 plane product: a walk of this grid does not fit 32-bit byte offsets");
    pd.tpx = (pd.tiles + 7) / 8; pd.hot = plane->hot_block; pd.w = (int)w; pd.far = pd.ny * PL_ROWS;
    pd.pitch = plane->table_pitch;
    const long long chunks = (nzr + pd.depth - 1) / pd.depth;
    const long long grid = 8ll * pd.tpx * chunks;
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    const float *xe = x - (long long)H.z0 * pd.far;            // the kernel addresses x and y in the numbering of the stored grid
    float *ye = y - (long long)H.z0 * pd.far;
    const char *cpool = static_cast<const char *>(pool);
    if (append) sell8_plane_f32_kernel<1, 18, true><<<(unsigned)grid, P32_LANES, 0, s>>>(xe, ye, alpha, ye, 1.0f, blocks, cpool, deltas, values, pd, H);
    else        sell8_plane_f32_kernel<0, 18, true><<<(unsigned)grid, P32_LANES, 0, s>>>(xe, ye, alpha, nullptr, 0.0f, blocks, cpool, deltas, values, pd, H);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}
} // namespace vexhip

VEXHIP_WARM_TU(plane32)
