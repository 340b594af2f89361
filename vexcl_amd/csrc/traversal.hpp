// Traversal: which row-block (512-row slice) a workgroup processes
// (include/vexhip.h vexhip_traversal).  Strip order is pure arithmetic -- no
// dependent load at workgroup start; an explicit map is one scalar load.
// Shared by the SELL / SELL8 product kernels (spmv.hip, sell8.hip, spmm.hip).
#pragma once
#include "common.hpp"

namespace vexhip {

// (round 6) z / beta: the vector a product adds as it stores, y = alpha A x + beta z (vexhip_spmat_apply_axpby_*; NULL: none) -- it travels with
// the traversal because every SELL-family kernel takes one by value and ends in store_pair
struct trav_dev { const int *order; int chunk, planes, plane_blocks; const void *z = nullptr; double beta = 0.0; };

// The addend of the NEXT product launched on this thread (set by vexhip_spmat_apply_axpby_*, taken -- and thereby marked as taken -- by
// the launchers that pass it on; a launcher that does not know it leaves it, and the caller reports an error instead of a wrong y).
struct pending_addend { const void *z = nullptr; double beta = 0.0; bool taken = false; };
inline pending_addend &next_addend() { static thread_local pending_addend a; return a; }
inline trav_dev with_addend(trav_dev t) {
    pending_addend &a = next_addend();
    if (a.z) { t.z = a.z; t.beta = a.beta; a.taken = true; }
    return t;
}

/// Slice of virtual block `vb` (the kernels that loop over several slices pass vb = blockIdx.x + k * gridDim.x;
/// a grid that is a multiple of 8 keeps every block on the XCD strip it started on).  Launch grids are
/// < 2^31 (checked on the host), so the strip arithmetic is 32-bit: a 64-bit division costs a
/// workgroup ~100 instructions of prologue, and the looping kernels pay it per slice.
__device__ __forceinline__ long long traversal_slice(const trav_dev &t, long long nblocks, unsigned vb) {
    if (t.order) return t.order[vb];
    if (t.chunk > 0) {
        const unsigned k = vb & 7u, q = vb >> 3;
        const unsigned chunk = (unsigned)t.chunk, planes = (unsigned)t.planes;
        const unsigned r = q / chunk, i = q - r * chunk;
        const unsigned tile = r / planes, p = r - tile * planes;
        const long long l = (long long)tile * (8 * chunk) + k * chunk + i;
        const long long lb = (long long)p * t.plane_blocks + l;
        return (l < t.plane_blocks && lb < nblocks) ? lb : -1;
    }
    return vb < nblocks ? (long long)vb : -1;
}

__device__ __forceinline__ long long traversal_block(const trav_dev &t, long long nblocks) {
    return traversal_slice(t, nblocks, blockIdx.x);
}

/// y[i], y[i+1] (=|+=) alpha * sum[0], sum[1] -- the two consecutive rows a lane of the SELL kernels owns.
/// One 16-byte store (and load, for +=) when both rows exist and y is 16-byte aligned: i is even.
template <typename V>
__device__ __forceinline__ void store_pair(long long n, long long i, V alpha, int append, const V (&sum)[2], V *__restrict__ y, const trav_dev &tv = trav_dev{nullptr, 0, 0, 0}) {
    typedef V v2 __attribute__((ext_vector_type(2)));
    const V *z = static_cast<const V *>(tv.z);           // != NULL: y = alpha sum + beta z (z may be y; then `append` is not looked at)
    const V beta = (V)tv.beta;
    if (i + 1 < n && ((reinterpret_cast<unsigned long long>(y) | reinterpret_cast<unsigned long long>(z)) & (2 * sizeof(V) - 1)) == 0) {
        v2 *yp = reinterpret_cast<v2 *>(y + i);
        v2 o; o.x = alpha * sum[0]; o.y = alpha * sum[1];
        if (z) { const v2 old = *reinterpret_cast<const v2 *>(z + i); o.x = beta * old.x + o.x; o.y = beta * old.y + o.y; *yp = o; }
        else if (append) { const v2 old = *yp; o.x = old.x + o.x; o.y = old.y + o.y; *yp = o; }
        else __builtin_nontemporal_store(o, yp);              // y is written once and not re-read by this kernel
    } else {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (i + q < n) {
                V o = alpha * sum[q];
                if (z) o = beta * z[i + q] + o;
                else if (append) o = y[i + q] + o;
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
    return t;                          // (the addend is attached by the launchers that pass it on: with_addend)
}

} // namespace vexhip
