// The GRID product for fp32 (round 5): y (=|+=) alpha * A * x for a matrix stored by grid line (grid.hip: a class per line, a
// table of value codes per class and position) whose lines have ANY length -- the walk of the fp64 grid product with FOUR rows
// per lane, as plane32.hip does it for 512-point lines: every request is 16 bytes per lane (at 4-byte addresses: lines start
// at any element), a wave covers 256 rows of a segment, workgroups are 1 .. 4 waves.  Until this kernel float matrices on
// grids other than 512-point lines took the pair product of the SELL-512 storage and ran SLOWER than the same matrix in
// double (384^3: 0.248 ms against 0.183; 640^3: 1.39 against 0.87).  Now 384^3 0.10 ms, 500^3 0.23, 640^3 0.54, 700^3 0.69 - 0.73
// (profiles/r05_fp32_sizes.json): 0.47 - 0.56 of the HBM peak by the bytes that must move -- a line of 384 floats fills 96 of
// the 128 lanes of its two waves, one of 640 fills 160 of 192: the lanes beyond the line still request.
// Semantics: the reference's ELL product (/root/reference/vexcl/spmat/hybrid_ell.inl:238-269: entries in storage order,
// products rounded before they are added, the scale applied to the sum); bit-identical to the fp32 CSR loop
// (spmat/csr.inl:163-170).  Compiled with -ffp-contract=off.
#include "common.hpp"
#include "lanes.hpp"
#include "grid.hpp"

#include <algorithm>
#include <cstdlib>

namespace vexhip {
namespace {

constexpr int G32_MAXT = 256;              // lanes of a workgroup at most: 1024 rows of a segment
constexpr unsigned G32_ABSENT = 255;       // table byte of a position without an entry (grid.hip)

// ZM: the addend of the result (plane.hip): 0 none, 1 beta times the array zs ('+=': zs = y, beta = 1), 2 beta times x itself (from the registers)
template <int ZM, int STORE_AUX>
__global__ __launch_bounds__(G32_MAXT)             // 199 registers, two waves per SIMD (capped at 168 for three, with 15 spilled: the same times)
void sell8_grid_f32_kernel(const float *__restrict__ x, float *__restrict__ y, float alpha, const float *__restrict__ zs, float beta,
        const int *__restrict__ line_class, const unsigned char *__restrict__ table, const float *__restrict__ values, grid_dev gd)
{
    constexpr int TY = 2;
    // LDS: the value table and the decoded values of the OTHER class, lane-private ([position * 4 + row][lane])
    __shared__ float s_value[256];
    __shared__ float s_other[28][G32_MAXT];

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
    int z = zc * gd.depth;
    const int zend = z + gd.depth < gd.nz ? z + gd.depth : gd.nz;
    if (z >= zend) return;
    const int nx = gd.nx, ny = gd.ny;
    const long long lines = gd.lines, x_last = gd.x_last, n = gd.n;
    const unsigned lane_b = 16u * (unsigned)t;
    const int nv = len - 4 * t < 0 ? 0 : (len - 4 * t > 4 ? 4 : len - 4 * t);      // rows of this lane inside the segment: it stores that many
    // the element beyond either end of the wave's 256 rows of a segment: lane 63 reads the one behind them, every other lane the
    // one in front (lane 0 uses it) -- byte offset from the start of the segment
    const int edge_b = (t >> 6) * 1024 + ((t & 63) == 63 ? 1024 : -4);

    for (int i = t; i < 256; i += (int)blockDim.x) s_value[i] = values[i];
    __syncthreads();

    // ---- a line class -> values (into s_other) and validity (returned) of this lane's four rows at the seven positions ----
    // (lanes beyond the end of the line read the last bytes of the position row: padding, 255)
    const int tb_off = row0 + 4 * t < gd.pitch - 4 ? row0 + 4 * t : gd.pitch - 4;
    auto decode = [&](int cls) -> unsigned {
        const unsigned char *tb = table + (long long)cls * 7 * gd.pitch + tb_off;
        unsigned bits = 0;
#pragma unroll
        for (int p = 0; p < 7; ++p) {
            const unsigned short *h = reinterpret_cast<const unsigned short *>(tb + (long long)p * gd.pitch);      // (row0 is even)
            const unsigned c4 = (unsigned)h[0] | ((unsigned)h[1] << 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned code = (c4 >> (8 * r)) & 255u;
                s_other[4 * p + r][t] = s_value[code];            // entry 255 of the value table is 0.0
                bits |= (code != G32_ABSENT ? 1u : 0u) << (4 * p + r);
            }
        }
        return bits;
    };

    const int hot = gd.hot;
    float aH[7][4];                             // the hot class: values ...
    unsigned long long mH[7][4];                // ... and the lanes with an entry, per position and row (plane32.hip)
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
    int other = -1;                             // what s_other holds now is the hot class's: never asked for

    // clamped requests (prologue, slow steps): line `l` of the tile's window (0 = the line above the tile, 1 .. TY = the tile,
    // TY + 1 = the line below) in plane zz, element by element.  What lies outside x is never referenced by an entry; what is
    // loaded in its place is multiplied by +0.0 behind a mask
    auto elem = [&](long long i) -> float { i = i < 0 ? 0 : i; i = i > x_last ? x_last : i; return x[i]; };
    auto ld = [&](int zz, int l) -> f4 {
        const long long i = ((long long)zz * ny + (y0 - 1 + l)) * nx + row0 + 4 * t;
        f4 r; r.x = elem(i); r.y = elem(i + 1); r.z = elem(i + 2); r.w = elem(i + 3);
        return r;
    };
    auto edge = [&](int zz, int l) -> float {
        return elem(((long long)zz * ny + (y0 - 1 + l)) * nx + row0 + (edge_b >> 2));
    };
    auto yelem = [&](long long i) -> float { i = i < 0 ? 0 : i; i = i > n - 1 ? n - 1 : i; return zs[i]; };
    auto yold = [&](int zz, int l) -> f4 {
        const long long i = ((long long)zz * ny + (y0 + l)) * nx + row0 + 4 * t;
        f4 r; r.x = yelem(i); r.y = yelem(i + 1); r.z = yelem(i + 2); r.w = yelem(i + 3);
        return r;
    };

    // ---- state at the top of the step for plane z: as in plane.hip / plane32.hip ----
    f4 Cs[4][TY], Hs[2][2], Yo[TY];
    float Es[2][TY];
    const unsigned line_b = (unsigned)nx * 4u;                        // bytes from a line to the next
    const unsigned plane_b32 = (unsigned)ny * line_b;                 // ... to the same line of the next plane (the plan: (depth + 4) of them < 2^32)
    const int z_first = z;
    // buffer resources of the fast loop: x from the segment of the line above the tile in the workgroup's first plane, y from
    // the tile's first line in that plane.  No range check: the fast loop only runs where every request of every lane lies
    // inside the arrays (zh below)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x + (((long long)z_first * ny + (y0 - 1)) * nx + row0)), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y + (((long long)z_first * ny + y0) * nx + row0), 0, -1, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ZM == 1 ? zs : x) + (((long long)z_first * ny + y0) * nx + row0), 0, -1, 0x00020000);
#pragma unroll
    for (int l = 0; l < TY; ++l) { Cs[0][l] = ld(z - 1, l + 1); Cs[1][l] = ld(z, l + 1); Cs[2][l] = ld(z + 1, l + 1); Cs[3][l] = ld(z + 2, l + 1); }
    const bool flat = gd.flat != 0;                                   // uniform (grid.hip: a 2-D operator on virtual lines never reads the lines above / below a tile)
    const f4 fzero = {0.f, 0.f, 0.f, 0.f};
    Hs[0][0] = Hs[0][1] = Hs[1][0] = Hs[1][1] = fzero;
    if (!flat) { Hs[0][0] = ld(z, 0); Hs[0][1] = ld(z, TY + 1); Hs[1][0] = ld(z + 1, 0); Hs[1][1] = ld(z + 1, TY + 1); }
#pragma unroll
    for (int l = 0; l < TY; ++l) {
        Es[0][l] = edge(z, l + 1); Es[1][l] = edge(z + 1, l + 1);
        if (ZM == 1) Yo[l] = yold(z, l);
    }

    // x at the seven positions {-far, -nx, -1, 0, +1, +nx, +far} of the lane's four rows of tile line l
#define G32_XS(P, C, N, H, E, l)                                                                                                     \
        const f4 c = C[l], up = (l) == 0 ? H[0] : C[0], dn = (l) == TY - 1 ? H[1] : C[TY - 1], pv = P[l], nw = N[l];                     \
        const float xs[4][7] = {{pv.x, up.x, shift_from_lower_lane(c.w, E[l]), c.x, c.y, dn.x, nw.x},                                 \
                                {pv.y, up.y, c.x, c.y, c.z, dn.y, nw.y},                                                              \
                                {pv.z, up.z, c.y, c.z, c.w, dn.z, nw.z},                                                              \
                                {pv.w, up.w, c.z, c.w, shift_from_upper_lane(c.x, E[l]), dn.w, nw.w}};
#define G32_HOT_SUMS(s)                                                                                                              \
        _Pragma("unroll") for (int p = 0; p < 7; ++p) _Pragma("unroll") for (int r = 0; r < 4; ++r) s[r] += aH[p][r] * keep_lanes(xs[r][p], mH[p][r]);
#define G32_OTHER_SUMS(s)                                                                                                            \
        _Pragma("unroll") for (int p = 0; p < 7; ++p) _Pragma("unroll") for (int r = 0; r < 4; ++r) s[r] += s_other[4 * p + r][t] * keep_bit(xs[r][p], bitsO, 4 * p + r);

    // fast steps: every request lies inside the arrays.  A lane may sit beyond the end of its line (it stores nothing, but it
    // requests): `over` = the furthest element, from the start of a line, that a lane of this workgroup asks for
    int zh = zend;
    {
        const long long over = row0 + 4 * (long long)blockDim.x + 1;
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
            auto fast_step = [&](f4 (&P)[TY], f4 (&C)[TY], f4 (&N)[TY], f4 (&H)[2], float (&E)[TY]) {
                f4 o[TY];
#pragma unroll
                for (int l = 0; l < TY; ++l) {
                    G32_XS(P, C, N, H, E, l)
                    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (use_hot[l] & 1ull) { G32_HOT_SUMS(s) } else { G32_OTHER_SUMS(s) }      // uniform
                    o[l].x = alpha * s[0]; o[l].y = alpha * s[1]; o[l].z = alpha * s[2]; o[l].w = alpha * s[3];
                    if (ZM == 1) o[l] = beta * Yo[l] + o[l];
                    if (ZM == 2) o[l] = beta * c + o[l];
                }
#pragma unroll
                for (int l = 0; l < TY; ++l) use_hot[l] >>= 1;
#pragma unroll
                for (int l = 0; l < TY; ++l)
                    if (l < nl) {                  // uniform; written once, not read again by this kernel
                        const int at = (int)(yo + l * line_b);
                        if (nv == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o[l]), ry, (int)lane_b, at, STORE_AUX);
                        else if (nv > 0) {          // the lane at the end of the line: its rows one by one
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[l].x), ry, (int)lane_b, at, STORE_AUX);
                            if (nv > 1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[l].y), ry, (int)lane_b + 4, at, STORE_AUX);
                            if (nv > 2) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[l].z), ry, (int)lane_b + 8, at, STORE_AUX);
                        }
                    }
                if (ZM == 1) {
#pragma unroll
                    for (int l = 0; l < TY; ++l) Yo[l] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rz, (int)lane_b, (int)(yo + plane_b32 + l * line_b), 0));
                }
                if (!flat) {
                    H[0] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)lane_b, (int)xo, 0));
                    H[1] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)lane_b, (int)(xo + (TY + 1) * line_b), 0));
                }
#pragma unroll
                for (int l = 0; l < TY; ++l) {
                    P[l] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)lane_b, (int)(xo + plane_b32 + (l + 1) * line_b), 0));
                    E[l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, edge_b + 4, (int)(xo + (l + 1) * line_b - 4u), 0));
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
                G32_XS(Cs[0], Cs[1], Cs[2], Hs[0], Es[0], l)
                float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                const int cls = __builtin_amdgcn_readfirstlane(line_class[li]);
                if (cls == hot) { G32_HOT_SUMS(s) }
                else {
                    if (cls != other) { bitsO = decode(cls); other = cls; }
                    G32_OTHER_SUMS(s)
                }
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = alpha * s[r];
                if (ZM == 1) { o[0] = beta * Yo[l].x + o[0]; o[1] = beta * Yo[l].y + o[1]; o[2] = beta * Yo[l].z + o[2]; o[3] = beta * Yo[l].w + o[3]; }
                if (ZM == 2) { o[0] = beta * c.x + o[0]; o[1] = beta * c.y + o[1]; o[2] = beta * c.z + o[2]; o[3] = beta * c.w + o[3]; }
                float *yr = y + li * nx + row0 + 4 * t;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (r < nv) __builtin_nontemporal_store(o[r], yr + r);
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
#undef G32_XS
#undef G32_HOT_SUMS
#undef G32_OTHER_SUMS
}

} // namespace
} // namespace vexhip

using namespace vexhip;

namespace vexhip {
int grid32_apply_axpby(int dev, void *stream, int64_t n, float alpha, int zm, const float *zs, float beta, const float *values,
        const float *x, float *y, const vexhip_grid *g);
}

extern "C" {

int vexhip_spmv_sell8v_grid_f32(int dev, void *stream, int64_t n, float alpha, int append, const float *values,
        const float *x, float *y, const vexhip_grid *g)
{
    return grid32_apply_axpby(dev, stream, n, alpha, append ? 1 : 0, y, 1.0f, values, x, y, g);
}

} // extern "C"

namespace vexhip {
// y = alpha A x + [zm 1: beta zs | zm 2: beta x] through the fp32 grid product (spmat.hip vexhip_spmat_apply_axpby_f32)
int grid32_apply_axpby(int dev, void *stream, int64_t n, float alpha, int zm, const float *zs, float beta, const float *values,
        const float *x, float *y, const vexhip_grid *g)
{
    VEXHIP_REQUIRE(g && g->usable && g->line_class && g->table && values && x && y, "bad grid product arguments");
    if (int rc = vexhip_sell8_grid_check(g, n)) return rc;
    VEXHIP_REQUIRE(g->x_last + 1 >= n, "bad grid plan");
    VEXHIP_SET_DEVICE(dev);
    grid_dev gd;
    gd.lines = n / g->nx; gd.x_last = g->x_last; gd.n = n;
    gd.nx = g->nx; gd.ny = g->lines_per_plane; gd.nz = g->planes;
    gd.segs = g->segments; gd.seg_len = g->segment_rows;
    gd.tiles = (gd.ny + 1) / 2 * gd.segs; gd.tpx = (gd.tiles + 7) / 8; gd.hot = g->hot_class; gd.pitch = g->pitch; gd.flat = g->flat;
    // A workgroup is 1 .. 4 waves (four rows per lane) and a CU holds eight waves of this kernel.  Measured (ms by walk depth,
    // profiles/r05_fp32_sizes.json): the best launch is ONE round of workgroups that just fills the CUs -- 384^3 (192 tiles x 2 waves)
    // 96 / 77 / 64 / 48 planes = 0.120 / 0.103 / 0.136 / 0.115 (3 / 3.75 / 4.5 / 6 workgroups per CU), 500^3 125 / 100 / 63 = 0.217 /
    // 0.304 / 0.228 (3.9 / 4.9 / 7.8), 512^3 128 / 86 / 64 = 0.214 / 0.244 / 0.259 (4 / 6 / 8) -- a little more than one round is the
    // worst.  Workgroups of three and four waves (two per CU, 1.25 .. 2 tiles per CU) cannot do that: many short walks, at least
    // eight per CU, as the fp64 grid product does it (640^3: 320 / 160 / 80 planes = 0.697 / 0.595 / 0.524).
    const int threads = std::max(64, std::min(G32_MAXT, ((g->segment_rows + 3) / 4 + 63) / 64 * 64));
    {
        const long long cus = std::max(1, info(dev).cus);
        const long long resident = std::max(1, 8 / (threads / 64));
        const long long cmax = std::max(1ll, (long long)gd.nz / 16);
        long long chunks = std::min(cmax, resident * cus / gd.tiles);
        if (resident < 3 || chunks < 1 || gd.tiles * chunks * 20 < resident * cus * 17) {
            double best = 0;
            chunks = 1;
            for (long long c = 1; c <= cmax; ++c) {
                const long long per_cu = (gd.tiles * c + cus - 1) / cus;
                if (per_cu < 8 && c < cmax) continue;
                const double est = (double)per_cu * (double)((gd.nz + c - 1) / c + 6);
                if (best == 0 || est < best) { best = est; chunks = c; }
                if (per_cu > 24) break;
            }
        }
        gd.depth = (int)((gd.nz + chunks - 1) / chunks);
    }
    if (const char *e = env(ENV_VEXHIP_GRID32_DEPTH)) if (std::atoi(e) > 0) gd.depth = std::min(std::atoi(e), (int)gd.nz);
    VEXHIP_REQUIRE(((long long)gd.depth + 4) * gd.ny * gd.nx * 4 < (1ll << 32), "bad grid plan");
    const long long chunks = (gd.nz + gd.depth - 1) / gd.depth;
    gd.cpx = gd.flat ? (int)((chunks + 7) / 8) : 0;
    const long long grid = gd.cpx ? 8ll * gd.cpx * gd.tiles : 8ll * gd.tpx * chunks;
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    const unsigned char *tb = static_cast<const unsigned char *>(g->table);
    hipStream_t s = as_stream(stream);
#define G32_LAUNCH(AP, AUX) sell8_grid_f32_kernel<AP, AUX><<<(unsigned)grid, (unsigned)threads, 0, s>>>(x, y, alpha, zs, beta, g->line_class, tb, values, gd)
#define G32_AUX(AP) switch (g->store_policy) { case 1: G32_LAUNCH(AP, 18); break; case 2: G32_LAUNCH(AP, 17); break; case 3: G32_LAUNCH(AP, 0); break; default: G32_LAUNCH(AP, 2); }
    VEXHIP_REQUIRE(zm == 0 || zm == 2 || (zm == 1 && zs), "grid product: the addend must be a vector");
    if (zm == 1) { G32_AUX(1) } else if (zm == 2) { G32_AUX(2) } else { G32_AUX(0) }
#undef G32_AUX
#undef G32_LAUNCH
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

} // namespace vexhip


VEXHIP_WARM_TU(grid32)
