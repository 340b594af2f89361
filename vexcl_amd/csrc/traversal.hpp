// Traversal: which row-block (512-row slice) a workgroup processes
// (include/vexhip.h vexhip_traversal).  Strip order is pure arithmetic -- no
// dependent load at workgroup start; an explicit map is one scalar load.
// Shared by the SELL / SELL8 product kernels (spmv.hip, sell8.hip, spmm.hip).
#pragma once
#include "common.hpp"

namespace vexhip {

struct trav_dev { const int *order; int chunk, planes, plane_blocks; };

__device__ __forceinline__ long long traversal_block(const trav_dev &t, long long nblocks) {
    const long long b = blockIdx.x;
    if (t.order) return t.order[b];
    if (t.chunk > 0) {
        const long long k = b & 7, q = b >> 3;
        const long long i = q % t.chunk, r = q / t.chunk;
        const long long p = r % t.planes, tile = r / t.planes;
        const long long l = tile * 8 * t.chunk + k * t.chunk + i;
        const long long lb = p * t.plane_blocks + l;
        return (l < t.plane_blocks && lb < nblocks) ? lb : -1;
    }
    return b < nblocks ? b : -1;
}

/// Grid size and device-side description of a host traversal (NULL / grid 0 = plain order).
inline trav_dev make_traversal(const vexhip_traversal *tr, long long nblocks, long long *grid) {
    trav_dev t = {nullptr, 0, 0, 0};
    *grid = nblocks;
    if (tr && tr->grid_blocks > 0) {
        t = trav_dev{tr->order, (int)tr->chunk, (int)tr->planes, (int)tr->plane_blocks};
        *grid = tr->grid_blocks;
    }
    return t;
}

} // namespace vexhip
