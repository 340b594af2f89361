// What the grid products share (grid.hip: fp64, two rows per lane; grid32.hip: fp32, four rows per lane): the geometry a launch is given.
#pragma once

namespace vexhip {
namespace {

struct grid_dev {
    long long lines;         // grid lines of the matrix: rows / nx
    long long x_last;        // largest valid index of x
    long long n;             // rows
    int nx, ny, nz;          // line length, lines per plane, planes: ceil(lines / ny)
    int depth;               // planes per workgroup
    int segs, seg_len;       // segments per line, rows per segment (even, <= 512)
    int tiles, tpx;          // ceil(ny / 2) * segs, and per XCD: ceil(tiles / 8)
    int hot;                 // line class decoded into registers with scalar masks
    int pitch;               // bytes per position row of a class table (>= segs * 512, padded with 255)
    int flat;                // no entry at +-nx in any line (a 2-D operator on virtual lines): the lines above / below a tile are not requested
    int cpx;                 // 0: an XCD owns TILES (all walks of a tile share its L2: the lines between neighbouring tiles);  > 0 (flat plans:
                             // a row of a 2-D grid has a handful of tiles -- six for 12 000 points -- and nothing to share between them): an XCD
                             // owns cpx consecutive WALKS of every tile, blockIdx -> (xcd, walk, tile)
};

} // namespace
} // namespace vexhip
