// What the plane products share (plane.hip: fp64, two rows per lane; plane32.hip: fp32, four rows per lane): the geometry a launch
// is given and the numbering of the seven stencil positions.
#pragma once
#include <hip/hip_runtime.h>

namespace vexhip {
namespace {

constexpr int PL_ROWS = 512;
constexpr unsigned PL_PAD_FIRST = 254;      // codes 254 / 255 are padding (sell8.hip)

struct plane_dev {
    long long nslices;       // 512-row lines of the matrix
    long long xlines;        // lines of x that may be loaded whole: (x_last + 1) / 512
    long long x_last;
    int ny;                  // lines per plane
    int nz;                  // planes: ceil(nslices / ny)
    int depth;               // planes per workgroup
    int tiles;               // ny / tile height
    int tpx;                 // tiles per XCD: ceil(tiles / 8)
    int hot;                 // dictionary block decoded into registers with scalar masks
    int w;                   // ELL width (<= 8)
    int far;                 // 512 * ny
    int pitch;               // 0: `pool` holds SELL-512 code blocks; > 0: class tables of the grid storage ([class][7 positions][pitch] value codes, grid.hip)
};


// diagonal -> position 0..6 in {-far, -512, -1, 0, 1, 512, far} (the plan has checked that it is one of them)
__device__ __forceinline__ int position_of(int d, int far) {
    return d == 0 ? 3 : d == -1 ? 2 : d == 1 ? 4 : d == -PL_ROWS ? 1 : d == PL_ROWS ? 5 : d == -far ? 0 : 6;
}

} // namespace
} // namespace vexhip
