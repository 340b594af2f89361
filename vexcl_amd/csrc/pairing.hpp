// Row-pair alignment for the SELL storages (sell8.hip, spmv.hip fill kernels).
//
// A lane of the SELL product kernels owns rows 2t and 2t+1 and reads x for both with ONE
// 16-byte load per ELL column when the two rows hold the same diagonal there (pair kernels,
// sell8.hip).  The fill kernels therefore place the first w entries of the two rows by a
// two-pointer merge of their diagonal lists (diagonal = column - row): equal diagonals share
// a column, otherwise the smaller one takes the column alone and the partner gets padding.
// A merge keeps the entries of EACH row in their CSR order whatever the rows look like
// (sorted or not), so the summation order of a row -- and with it every bit of the result --
// is unchanged.  If the merged list does not fit the ELL width the pair keeps the plain
// packing (entry k in column k).
#pragma once

namespace vexhip {

// The merge itself, over any source of diagonals: d(q, k) = diagonal of the k-th ELL entry of row q of the pair.
template <typename Diag>
struct pair_merge {
    Diag d;
    int n[2];                  // number of ELL entries (min(row length, w)) of the two rows
    int p[2];                  // entries consumed so far
    bool aligned;

    __device__ void init(const Diag &d_, int n0, int n1, int w) {
        d = d_; n[0] = n0; n[1] = n1;
        int pa = 0, pb = 0, merged = 0;
        while (pa < n0 || pb < n1) {
            if (pa < n0 && pb < n1) {
                const long long da = d(0, pa), db = d(1, pb);
                if (da == db) { ++pa; ++pb; } else if (da < db) ++pa; else ++pb;
            } else if (pa < n0) ++pa; else ++pb;
            ++merged;
        }
        aligned = merged <= w;
        p[0] = p[1] = 0;
    }

    /// Entries (index within the row, -1 = none) that go to the next ELL column.
    __device__ void next(int &k0, int &k1) {
        k0 = k1 = -1;
        const bool h0 = p[0] < n[0], h1 = p[1] < n[1];
        if (!aligned) {
            if (h0) k0 = p[0]++;
            if (h1) k1 = p[1]++;
            return;
        }
        if (h0 && h1) {
            const long long da = d(0, p[0]), db = d(1, p[1]);
            if (da <= db) k0 = p[0]++;
            if (db <= da) k1 = p[1]++;
        } else if (h0) k0 = p[0]++;
        else if (h1) k1 = p[1]++;
    }
};

// diagonals read from the CSR arrays (entry offsets are 64-bit: a device may hold 2^31 entries or more)
struct diag_global {
    const int *col; long long row, b[2];
    __device__ __forceinline__ long long operator()(int q, int k) const { return (long long)col[b[q] + k] - (row + q); }
};
// diagonals of a pair whose rows lie in a piece of the column array that its wave copied into LDS: entry k of row q at
// s[off[q] + k] (round 3: the fill kernels copy the 128 rows of a wave with coalesced loads -- a lane reading ITS rows'
// entries from global memory reads 4 bytes at a stride of 56, and the merge is a chain of dependent look-ups on top)
struct diag_lds {
    const int *s; int off[2]; long long row;
    __device__ __forceinline__ long long operator()(int q, int k) const { return (long long)s[off[q] + k] - (row + q); }
};

struct pair_walk {
    pair_merge<diag_global> m;
    __device__ void init(const int *col_, long long row_, long long b0, int n0, long long b1, int n1, int w) {
        diag_global g; g.col = col_; g.row = row_; g.b[0] = b0; g.b[1] = b1;
        m.init(g, n0, n1, w);
    }
    /// Entries (offsets into the CSR arrays, -1 = none) that go to the next ELL column.
    __device__ void next(long long &e0, long long &e1) {
        int k0, k1;
        m.next(k0, k1);
        e0 = k0 >= 0 ? m.d.b[0] + k0 : -1;
        e1 = k1 >= 0 ? m.d.b[1] + k1 : -1;
    }
};

/// Largest column index among the first w entries of every row (atomicMax into *out, which starts at -1):
/// x holds at least that many + 1 elements, so a 16-byte load that ends at x[max] stays inside x.
template <typename P>
static __global__ __launch_bounds__(256)
void ell_max_col_kernel(long long n, int w, const P *__restrict__ ptr, const int *__restrict__ col, int *out) {
    int m = -1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const P b = ptr[i], e = ptr[i + 1];
        for (int j = 0; j < w && b + j < e; ++j) { const int c = col[b + j]; m = c > m ? c : m; }
    }
    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_down(m, o, 64); m = v > m ? v : m; }
    if ((threadIdx.x & 63) == 0 && m >= 0) atomicMax(out, m);
}

} // namespace vexhip
