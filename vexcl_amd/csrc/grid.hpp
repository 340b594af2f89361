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
};

} // namespace
} // namespace vexhip
