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

/// y[i], y[i+1] (=|+=) alpha * sum[0], sum[1] -- the two consecutive rows a lane of the SELL kernels owns.
/// One 16-byte store (and load, for +=) when both rows exist and y is 16-byte aligned: i is even.
template <typename V>
__device__ __forceinline__ void store_pair(long long n, long long i, V alpha, int append, const V (&sum)[2], V *__restrict__ y) {
    typedef V v2 __attribute__((ext_vector_type(2)));
    if (i + 1 < n && (reinterpret_cast<unsigned long long>(y) & (2 * sizeof(V) - 1)) == 0) {
        v2 *yp = reinterpret_cast<v2 *>(y + i);
        v2 o; o.x = alpha * sum[0]; o.y = alpha * sum[1];
        if (append) { const v2 old = *yp; o.x = old.x + o.x; o.y = old.y + o.y; *yp = o; }
        else __builtin_nontemporal_store(o, yp);              // y is written once and not re-read by this kernel
    } else {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (i + q < n) {
                V o = alpha * sum[q];
                if (append) o = y[i + q] + o;
                y[i + q] = o;
            }
    }
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
