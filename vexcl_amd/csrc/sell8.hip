// SELL-512 with 8-bit DIAGONAL CODES instead of 32-bit column indices.
//
// For banded / stencil matrices the difference (column - row) takes only a few
// distinct values.  If the ELL part of a matrix uses at most 255 distinct
// diagonals, each column index is stored as one byte -- the position of its
// diagonal in a sorted table -- and re-built in the kernel as row + delta[code]
// (code 255 = padding).  The matrix stream shrinks from 12 to 9 bytes per entry
// (fp64), the arithmetic and its order are unchanged, results stay bit-identical
// to hybrid ELL / CSR.  Matrices with more diagonals simply keep 32-bit columns
// (vexhip_sell8_analyze_i32 reports -1).  The reference stores 32/64-bit columns
// (spmat/hybrid_ell.inl:138-198); this is the HBM-byte lever of the MI355X design.
//
// Slice layout (one contiguous region per 512-row slice, w = ELL width):
//     ceil(w/2) * 1 KiB of codes:  word [jp][t] = { (2jp, 2t), (2jp, 2t+1), (2jp+1, 2t), (2jp+1, 2t+1) }
//     w * 512 values, j-major (value (r, j) at j*512 + r)
// so lane t (rows 2t, 2t+1) reads one 4-byte word per column pair and 16 bytes of
// values per column.  Compiled with -ffp-contract=off like spmv.hip.
#include "common.hpp"
#include "traversal.hpp"
#include "pairing.hpp"
#include "halo.hpp"
#include "lanes.hpp"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace vexhip {

// A/B switch of the SELL products (vexhip_spmv_sell8_set_variant, shared with spmv.hip):
// 0 = pair kernels (one 16-byte gather per lane and column; default), 1 = one 8-byte gather per entry (round 1).
// 2 = pair kernels even where the march kernel applies.
int g_sell8_variant = 0;
// largest column index of the ELL part seen by the last sell8 / sell8v fill on this thread (x holds at least that + 1 elements)
thread_local long long g_fill_max_col = -1;
// set by the fused analysis (below) for the fill that follows it on this thread: the largest ELL column is known, the fill
// skips its own pass over the column indices
thread_local long long g_max_col_hint = -1;
thread_local const void *g_hint_ptr = nullptr;
thread_local long long g_hint_n = -1;

namespace {

constexpr int S8_ROWS = 512;
constexpr int S8_PAD = 255;              // padding (254 = padding too, see the pair kernels)
constexpr unsigned S8_FIRST_PAD = 254;
constexpr unsigned S8_PAD_UNSAFE = 254;     // padding whose partner must not use the 16-byte load (pair kernels)
constexpr int HASH_SLOTS = 1024;
constexpr int LOCAL_SLOTS = 512;          // per-workgroup set: twice the 255 entries a table may hold
constexpr int EMPTY = INT_MIN;

typedef double double2v __attribute__((ext_vector_type(2)));
typedef float  float2v  __attribute__((ext_vector_type(2)));

__host__ __device__ inline long long slice_bytes(long long w, long long value_bytes) {
    return ((w + 1) / 2) * 1024 + w * S8_ROWS * value_bytes;
}

template <typename V> struct vec2;
template <> struct vec2<double> { typedef double2v type; };
template <> struct vec2<float> { typedef float2v type; };

// ---------------------------------------------------------------------------
template <typename V, int W>
__global__ __launch_bounds__(256)
void sell8_kernel(long long n, long long nslices, V alpha, int append, int ell_w,
        const char *__restrict__ buf, const int *__restrict__ deltas,
        const int *__restrict__ csr_ptr, const int *__restrict__ csr_col, const V *__restrict__ csr_val,
        const V *__restrict__ x, V *__restrict__ y, trav_dev trav, const char *__restrict__ pool, const int *__restrict__ blocks)
{
    __shared__ int s_delta[256];
    s_delta[threadIdx.x] = deltas[threadIdx.x];
    __syncthreads();

    const long long s = traversal_block(trav, nslices);
    if (s < 0) return;
    const int t = threadIdx.x;
    const long long i = s * S8_ROWS + 2 * t;
    const int w = W > 0 ? W : ell_w;
    const int wp = (w + 1) / 2;
    const char *slice = buf + s * slice_bytes(w, sizeof(V));
    // slice dictionary (below): the codes of slice s are block blocks[s] of the pool; the values stay where they are
    const unsigned *cw = reinterpret_cast<const unsigned *>(blocks ? pool + (long long)blocks[s] * ((long long)wp * 1024) : slice) + t;
    const V *vp = reinterpret_cast<const V *>(slice + (long long)wp * 1024) + 2 * t;
    typedef typename vec2<V>::type V2;

    V sum[2] = {V(0), V(0)};
    if constexpr (W > 0) {
        constexpr int WP = (W + 1) / 2;
        unsigned c[WP]; V2 v[W];
#pragma unroll
        for (int jp = 0; jp < WP; ++jp) c[jp] = __builtin_nontemporal_load(cw + jp * 256);
#pragma unroll
        for (int j = 0; j < W; ++j) v[j] = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(vp + j * S8_ROWS));
        V xv[W][2];
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned code = (c[j >> 1] >> (8 * ((j & 1) * 2 + q))) & 255u;
                xv[j][q] = (code < S8_FIRST_PAD) ? x[i + q + s_delta[code]] : V(0);
            }
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned code = (c[j >> 1] >> (8 * ((j & 1) * 2 + q))) & 255u;
                if (code < S8_FIRST_PAD) sum[q] += v[j][q] * xv[j][q];
            }
    } else {
        for (int j = 0; j < w; ++j) {
            const unsigned cword = cw[(j >> 1) * 256];
            const V2 vv = *reinterpret_cast<const V2 *>(vp + (long long)j * S8_ROWS);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned code = (cword >> (8 * ((j & 1) * 2 + q))) & 255u;
                if (code < S8_FIRST_PAD) sum[q] += vv[q] * x[i + q + s_delta[code]];
            }
        }
    }
    if (csr_ptr) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (i + q < n)
                for (int j = csr_ptr[i + q], e = csr_ptr[i + q + 1]; j < e; ++j) sum[q] += csr_val[j] * x[csr_col[j]];
    }
    store_pair<V>(n, i, alpha, append, sum, y, trav);
}

// ---- set-up: which diagonals does the ELL part use? ------------------------------------
__device__ __forceinline__ bool set_insert(int *set, int slots, int d, bool *is_new) {
    unsigned h = ((unsigned)d * 2654435761u) % (unsigned)slots;
    for (int probe = 0; probe < slots; ++probe) {
        int old = atomicCAS(&set[h], EMPTY, d);
        if (old == EMPTY) { if (is_new) *is_new = true; return true; }
        if (old == d) return true;
        h = (h + 1) % (unsigned)slots;
    }
    return false;
}

// gset: HASH_SLOTS ints (EMPTY-filled); info[0] = number of distinct diagonals, info[1] = overflow flag
template <typename P>
__global__ __launch_bounds__(256)
void delta_collect_kernel(long long n, int w, const P *__restrict__ ptr, const int *__restrict__ col,
        int *gset, int *info)
{
    __shared__ int s_set[LOCAL_SLOTS];
    __shared__ int s_over, s_count;
    for (int k = threadIdx.x; k < LOCAL_SLOTS; k += blockDim.x) s_set[k] = EMPTY;
    if (threadIdx.x == 0) { s_over = *(volatile int *)&info[1]; s_count = 0; }
    __syncthreads();
    // a matrix without structure overflows the set at once: stop looking (a full set costs a whole probe
    // sequence per insertion) -- s_over starts from the global flag, so later workgroups do not even begin
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (*(volatile int *)&s_over || *(volatile int *)&info[1]) break;
        const P b = ptr[i], e = ptr[i + 1];
        int last = EMPTY;
        for (int j = 0; j < w && b + j < e; ++j) {
            long long dl = (long long)col[b + j] - i;
            if (dl <= INT_MIN || dl > INT_MAX) { s_over = 1; continue; }
            int d = (int)dl;
            if (d == last) continue;
            last = d;
            bool fresh = false;
            if (!set_insert(s_set, LOCAL_SLOTS, d, &fresh)) s_over = 1;
            else if (fresh && atomicAdd(&s_count, 1) >= 254) { s_over = 1; atomicExch(&info[1], 1); }   // a 255th diagonal: not coded
        }
    }
    __syncthreads();
    if (s_over) { if (threadIdx.x == 0) atomicExch(&info[1], 1); return; }
    for (int k = threadIdx.x; k < LOCAL_SLOTS; k += blockDim.x) {
        if (s_set[k] == EMPTY) continue;
        bool is_new = false;
        if (!set_insert(gset, HASH_SLOTS, s_set[k], &is_new)) atomicExch(&info[1], 1);
        else if (is_new) atomicAdd(&info[0], 1);
    }
    if (threadIdx.x == 0 && s_over) atomicExch(&info[1], 1);
}

// code of a diagonal (binary search in the sorted table), S8_PAD if it is not there
__device__ __forceinline__ unsigned delta_code(const int *s_table, int ndeltas, long long dl) {
    if (dl < INT_MIN || dl > INT_MAX) return S8_PAD;
    const int d = (int)dl;
    int lo = 0, hi = ndeltas;                    // first table entry >= d
    while (lo < hi) { int mid = (lo + hi) >> 1; if (s_table[mid] < d) lo = mid + 1; else hi = mid; }
    return (lo < ndeltas && s_table[lo] == d) ? (unsigned)lo : (unsigned)S8_PAD;
}

// padding code of the empty half of a column whose other half holds column index `c` of row half `q`:
// 255 when the 16-byte load of the pair kernels (x[c - q], x[c - q + 1]) stays inside x, 254 otherwise
__device__ __forceinline__ unsigned pad_code(int partner_col, int partner_q, int max_col) {
    if (partner_col < 0) return S8_PAD;                                 // both halves empty
    const long long first = (long long)partner_col - partner_q;        // element the 16-byte load starts at
    return (first >= 0 && first + 1 <= max_col) ? (unsigned)S8_PAD : S8_PAD_UNSAFE;
}

// The entries of the 128 rows a wave fills (64 row pairs) are one contiguous piece of col / val: the wave copies it into its own
// LDS region with coalesced loads -- no workgroup barrier, the waves keep their own pace -- and merge, code look-ups and
// stores work from there.  (History at 512^3, sell8v fill: every lane loading its pair's entries itself, 4 / 8 bytes at a
// stride of 56 / 112, 8.8 ms; the same with the 512 rows of a slice copied by the whole workgroup behind barriers, 56 KiB
// of LDS: 10.9 ms; this: see DESIGN.md 6.)  A piece beyond WAVE_CAP entries (long rows) is walked in global memory.
constexpr int WAVE_CAP = 1024;
template <typename V>
__device__ __forceinline__ bool wave_stage(const int *__restrict__ col, const V *__restrict__ val, long long b0_lane, long long e1_lane,
        int *s_c, V *s_v, long long &e0)
{
    const int lane = threadIdx.x & 63;
    // lanes past the end of the matrix hold b = e = 0: the piece is [first lane's begin, last REAL lane's end)
    const unsigned long long real = __ballot(e1_lane > 0 || b0_lane > 0);
    long long lo = b0_lane, hi = e1_lane;
    e0 = __shfl(lo, 0, 64);                                         // lane 0 is real whenever any lane is
    const int lastl = real ? 63 - __builtin_clzll(real) : 0;
    const long long end = __shfl(hi, lastl, 64);
    const long long cnt = end - e0;
    if (cnt > WAVE_CAP || cnt < 0) return false;                     // uniform
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();       // the previous trip's reads of this region are done
    for (int k = lane; k < (int)cnt; k += 64) { s_c[k] = col[e0 + k]; s_v[k] = val[e0 + k]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return true;
}

// One LDS counter bump per entry -- but the lanes of a wave that walk the rows of one stencil hold the SAME code in the same
// column, and 64 same-address LDS atomics serialise.  A wave whose active lanes agree adds their number once.
__device__ __forceinline__ void count_code(unsigned *s_cnt, unsigned code) {
    const unsigned long long act = __ballot(1);
    const unsigned c0 = __builtin_amdgcn_readfirstlane(code);
    if (__ballot(code == c0) == act) {
        if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicAdd(&s_cnt[c0], (unsigned)__popcll(act));
    } else {
        atomicAdd(&s_cnt[code], 1u);
    }
}

// table: sorted diagonals (ndeltas valid entries); counts[256]: entries per code; info[1]: set if a diagonal is missing;
// max_col: largest column index of the ELL part (ell_max_col_kernel)
template <typename V, typename P, bool STAGED>
__global__ __launch_bounds__(256)
void sell8_fill_kernel(long long n, long long nslices, int w, int ndeltas,
        const P *__restrict__ ptr, const int *__restrict__ col, const V *__restrict__ val,
        const int *__restrict__ table, const int *__restrict__ max_col_p, char *__restrict__ buf, unsigned long long *counts, int *info)
{
    __shared__ int s_table[256];
    __shared__ unsigned s_cnt[256];
    __shared__ int s_cw[STAGED ? 4 : 1][STAGED ? WAVE_CAP : 1];          // STAGED (w <= 8): the pair's entries in lane-private LDS slots (sell8v_fill_kernel)
    __shared__ V s_vw[STAGED ? 4 : 1][STAGED ? WAVE_CAP : 1];
    s_table[threadIdx.x] = threadIdx.x < ndeltas ? table[threadIdx.x] : INT_MAX;
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int max_col = *max_col_p;
    const int wp = (w + 1) / 2;
    const int lt = threadIdx.x;
    // one lane per row PAIR (the unit the product kernel reads)
    for (long long pr = (long long)blockIdx.x * blockDim.x + threadIdx.x; pr < nslices * (S8_ROWS / 2);
         pr += (long long)gridDim.x * blockDim.x) {
        const long long s = pr / (S8_ROWS / 2);
        const int t = (int)(pr % (S8_ROWS / 2));
        char *slice = buf + s * slice_bytes(w, sizeof(V));
        unsigned *cw = reinterpret_cast<unsigned *>(slice) + t;
        V *vp = reinterpret_cast<V *>(slice + (long long)wp * 1024) + 2 * t;
        long long b[2] = {0, 0}, e[2] = {0, 0};
        const long long i = s * S8_ROWS + 2 * t;
        for (int q = 0; q < 2; ++q) if (i + q < n) { b[q] = ptr[i + q]; e[q] = ptr[i + q + 1]; }
        const int n0 = (int)min(e[0] - b[0], (long long)w), n1 = (int)min(e[1] - b[1], (long long)w);
        pair_walk pw;
        pair_merge<diag_lds> pm;
        bool staged = false, same = false;
        int *s_c = s_cw[STAGED ? lt >> 6 : 0]; V *s_v = s_vw[STAGED ? lt >> 6 : 0];
        if constexpr (STAGED) {
            long long e0;
            // (the previous trip's reads of this wave's region are done: same wave, program order)
            staged = wave_stage<V>(col, val, b[0], e[1] > 0 ? e[1] : e[0], s_c, s_v, e0);
            if (staged) {
                diag_lds g; g.s = s_c; g.off[0] = (int)(b[0] - e0); g.off[1] = (int)(b[1] - e0); g.row = i;
                pm.d = g;
                // the common pair: both rows hold the same diagonals in the same order (column of row 2t + 1 = column of
                // row 2t, plus one) -- entry j of both goes to column j, no merge (a chain of dependent look-ups)
                same = n0 == n1 && n0 <= 8;
                if (same) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (k < n0 && s_c[g.off[0] + k] + 1 != s_c[g.off[1] + k]) same = false;
                }
                if (!same) pm.init(g, n0, n1, w);
            }
        }
        if (!staged) pw.init(col, i, b[0], n0, b[1], n1, w);
        for (int jp = 0; jp < wp; ++jp) {
            unsigned word = 0;
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * jp + jj;
                int ec[2] = {-1, -1}; V ev[2] = {V(0), V(0)}; bool has[2] = {false, false};
                if (j < w) {
                    if (staged) {
                        int k[2];
                        if (same) k[0] = k[1] = j < n0 ? j : -1; else pm.next(k[0], k[1]);
                        for (int q = 0; q < 2; ++q) if (k[q] >= 0) { has[q] = true; ec[q] = s_c[pm.d.off[q] + k[q]]; ev[q] = s_v[pm.d.off[q] + k[q]]; }
                    } else {
                        long long en[2]; pw.next(en[0], en[1]);
                        for (int q = 0; q < 2; ++q) if (en[q] >= 0) { has[q] = true; ec[q] = col[en[q]]; ev[q] = val[en[q]]; }
                    }
                }
                for (int q = 0; q < 2; ++q) {
                    unsigned code;
                    if (has[q]) {
                        if (q == 1 && has[0] && ec[1] == ec[0] + 1) code = (word >> (8 * (jj * 2))) & 255u;     // the partner's diagonal: its code
                        else code = delta_code(s_table, ndeltas, (long long)ec[q] - (i + q));
                        if (code != S8_PAD) count_code(s_cnt, code); else atomicExch(&info[1], 1);
                    } else code = pad_code(has[1 - q] ? ec[1 - q] : -1, 1 - q, max_col);
                    word |= code << (8 * (jj * 2 + q));
                    if (j < w) vp[(long long)j * S8_ROWS + q] = ev[q];
                }
            }
            cw[jp * 256] = word;
        }
    }
    __syncthreads();
    if (s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

// ---------------------------------------------------------------------------
// SELL8V: diagonal codes AND value codes.  Matrices assembled from a constant-
// coefficient stencil hold a handful of distinct VALUES (the 7-point Poisson
// matrix: 1, -h^-2, 6 h^-2).  When the ELL part uses at most 255 distinct values
// (bit patterns), each value is stored as one byte -- its position in a sorted
// table that the kernel keeps in LDS -- next to the one-byte diagonal code: 2
// instead of 9 bytes per fp64 entry (the reference's answer to such matrices is a
// separate class, SpMatCCSR; here SpMat detects them).  The arithmetic and its
// order are unchanged: results stay bit-identical to SELL8 / hybrid ELL / CSR.
// Slice layout: ceil(w/2) KiB of diagonal codes, then ceil(w/2) KiB of value codes,
// both packed as in SELL8.
// ---------------------------------------------------------------------------
template <typename V> struct bits_of;
template <> struct bits_of<double> { typedef unsigned long long type; };
template <> struct bits_of<float> { typedef unsigned type; };

template <typename V, int W>
__global__ __launch_bounds__(256)
void sell8v_kernel(long long n, long long nslices, V alpha, int append, int ell_w,
        const char *__restrict__ buf, const int *__restrict__ deltas, const V *__restrict__ values,
        const int *__restrict__ csr_ptr, const int *__restrict__ csr_col, const V *__restrict__ csr_val,
        const V *__restrict__ x, V *__restrict__ y, trav_dev trav, const int *__restrict__ blocks)
{
    __shared__ int s_delta[256];
    __shared__ V s_value[256];
    s_delta[threadIdx.x] = deltas[threadIdx.x];
    s_value[threadIdx.x] = values[threadIdx.x];
    __syncthreads();

    const long long s = traversal_block(trav, nslices);
    if (s < 0) return;
    const int t = threadIdx.x;
    const long long i = s * S8_ROWS + 2 * t;
    const int w = W > 0 ? W : ell_w;
    const int wp = (w + 1) / 2;
    const long long sb = blocks ? (long long)blocks[s] : s;                 // slice dictionary (below): where slice s is stored
    const unsigned *cw = reinterpret_cast<const unsigned *>(buf + sb * ((long long)wp * 2048)) + t;
    const unsigned *vw = cw + wp * 256;

    V sum[2] = {V(0), V(0)};
    if constexpr (W > 0) {
        constexpr int WP = (W + 1) / 2;
        unsigned c[WP], vc[WP];
#pragma unroll
        for (int jp = 0; jp < WP; ++jp) { c[jp] = __builtin_nontemporal_load(cw + jp * 256); vc[jp] = __builtin_nontemporal_load(vw + jp * 256); }
        V xv[W][2];
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned code = (c[j >> 1] >> (8 * ((j & 1) * 2 + q))) & 255u;
                xv[j][q] = (code < S8_FIRST_PAD) ? x[i + q + s_delta[code]] : V(0);
            }
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int sh = 8 * ((j & 1) * 2 + q);
                const unsigned code = (c[j >> 1] >> sh) & 255u;
                if (code < S8_FIRST_PAD) sum[q] += s_value[(vc[j >> 1] >> sh) & 255u] * xv[j][q];
            }
    } else {
        // any width (round 6: eight columns per trip -- their four code words, then their sixteen elements of x, are requested before the
        // first product; with one column per trip a 27-point row waited for memory 27 times: 0.60 -> see profiles/r06_widen_probe_unrolled.json).
        // Sums in column order, as before.
        int j = 0;
        for (; j + 8 <= w; j += 8) {
            unsigned c[4], vc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { c[u] = cw[((j >> 1) + u) * 256]; vc[u] = vw[((j >> 1) + u) * 256]; }
            V xv[8][2];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned code = (c[u >> 1] >> (8 * ((u & 1) * 2 + q))) & 255u;
                    xv[u][q] = (code < S8_FIRST_PAD) ? x[i + q + s_delta[code]] : V(0);
                }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int sh = 8 * ((u & 1) * 2 + q);
                    const unsigned code = (c[u >> 1] >> sh) & 255u;
                    if (code < S8_FIRST_PAD) sum[q] += s_value[(vc[u >> 1] >> sh) & 255u] * xv[u][q];
                }
        }
        for (; j < w; ++j) {
            const unsigned cword = cw[(j >> 1) * 256], vword = vw[(j >> 1) * 256];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int sh = 8 * ((j & 1) * 2 + q);
                const unsigned code = (cword >> sh) & 255u;
                if (code < S8_FIRST_PAD) sum[q] += s_value[(vword >> sh) & 255u] * x[i + q + s_delta[code]];
            }
        }
    }
    if (csr_ptr) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (i + q < n)
                for (int j = csr_ptr[i + q], e = csr_ptr[i + q + 1]; j < e; ++j) sum[q] += csr_val[j] * x[csr_col[j]];
    }
    store_pair<V>(n, i, alpha, append, sum, y, trav);
}

// ---------------------------------------------------------------------------
// RUNS of three diagonals, decoded at set-up (round 6): value-coded slices wider than nine columns whose distinct slices live in the
// dictionary -- 19- and 27-point stencils, 9-point operators in 2-D, dense bands.
// Four forms that only cut the REQUESTS of such a product (a 27-point row holds nine triples d-1, d, d+1: one 16-byte request and two DPP
// shifts serve a lane's six elements of x) left its time where it was (profiles/r06_runs_of_three.md): the product is bound by what it
// does per ENTRY of its two-byte codes -- extract, test for padding, read the value table, select, multiply: ~10 instructions, 3 456
// entries per wave and slice.  So the entries are decoded ONCE per distinct slice, at set-up (sell8v_runs_plan_kernel): for every wave of
// a dictionary block and every ELL column, which lanes hold an entry in their first / second row (two 64-bit masks) and -- where all of
// them carry the same value, as the rows of a constant-coefficient stencil do -- that value; for every aligned group of three columns,
// whether each sits on ONE diagonal and the three are consecutive.  The product reads these with scalar loads: a FAST group costs one
// request for x (+ the wave's two edge elements), and per entry a select of x by the mask and the multiply-add with a scalar value --
// what the grid storage's walk does with its hot class.  Groups that are not fast (the first columns of a wave that holds boundary
// rows, values that differ within a column, the first / last waves of the matrix) take the codes' path of the any-width kernel.
// Sums in column order; an absent entry adds (value) * (+0.0) = +-0.0, which leaves a sum that started at +0.0 bit for bit (the
// pair kernels below argue the same way): bit-identical to the CSR loop.
constexpr int RUNS_GROUPS = 11;                          // groups of three columns: ELL widths up to 33
constexpr int RUNS_HEAD = 12;                            // ints: mask of the fast groups, then every group's centre diagonal
constexpr int RUNS_STRIDE = RUNS_HEAD + 6 * 3 * RUNS_GROUPS;   // + per column {mask of first rows, mask of second rows, value}: 8 bytes each

template <typename V>
__global__ __launch_bounds__(256)
void sell8v_runs_plan_kernel(const char *__restrict__ pool, int w, const int *__restrict__ deltas, const V *__restrict__ values, int *__restrict__ desc, int *__restrict__ total)
{
    __shared__ int s_delta[256];
    __shared__ V s_value[256];
    s_delta[threadIdx.x] = deltas[threadIdx.x];
    s_value[threadIdx.x] = values[threadIdx.x];
    __syncthreads();
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wp = (w + 1) / 2;
    const unsigned *cw = reinterpret_cast<const unsigned *>(pool + (long long)blockIdx.x * ((long long)wp * 2048)) + t;
    const unsigned *vw = cw + wp * 256;
    int *out = desc + ((long long)blockIdx.x * 4 + wave) * RUNS_STRIDE;
    unsigned mask = 0;
    for (int k = 0; k < RUNS_GROUPS; ++k) {
        int dg[3]; bool ok = true;
        for (int u = 0; u < 3; ++u) {
            const int j = 3 * k + u;
            unsigned long long m0 = 0, m1 = 0, vbits = 0;
            dg[u] = INT_MIN;
            if (j < w) {
                const unsigned word = cw[(j >> 1) * 256] >> (16 * (j & 1)), vword = vw[(j >> 1) * 256] >> (16 * (j & 1));
                const unsigned c0 = word & 255u, c1 = (word >> 8) & 255u, v0 = vword & 255u, v1 = (vword >> 8) & 255u;
                const bool e0 = c0 < S8_FIRST_PAD, e1 = c1 < S8_FIRST_PAD;
                m0 = __builtin_amdgcn_ballot_w64(e0); m1 = __builtin_amdgcn_ballot_w64(e1);
                const unsigned long long any = m0 | m1;
                if (any) {
                    const int first = __builtin_ctzll(any);
                    const unsigned cu = (unsigned)__shfl((int)(e0 ? c0 : c1), first, 64), vu = (unsigned)__shfl((int)(e0 ? v0 : v1), first, 64);
                    const bool one_diagonal = !__builtin_amdgcn_ballot_w64((e0 && c0 != cu) || (e1 && c1 != cu));
                    const bool one_value = !__builtin_amdgcn_ballot_w64((e0 && v0 != vu) || (e1 && v1 != vu));
                    const V val = s_value[vu];
                    // (a value that is not finite times the +0.0 of an absent entry would be NaN: such a column keeps the codes' path)
                    if (one_diagonal && one_value && val - val == V(0)) {
                        dg[u] = s_delta[cu];
                        __builtin_memcpy(&vbits, &val, sizeof(V));
                    }
                }
            }
            ok = ok && dg[u] != INT_MIN;
            if (lane == 0 && (m0 | m1)) atomicAdd(total + 1, 1);                       // columns that hold entries at all
            if (lane == 0) {
                int *col = out + RUNS_HEAD + 6 * j;
                col[0] = (int)(unsigned)m0; col[1] = (int)(unsigned)(m0 >> 32); col[2] = (int)(unsigned)m1; col[3] = (int)(unsigned)(m1 >> 32);
                col[4] = (int)(unsigned)vbits; col[5] = (int)(unsigned)(vbits >> 32);
            }
        }
        ok = ok && (long long)dg[1] == (long long)dg[0] + 1 && (long long)dg[2] == (long long)dg[0] + 2;
        if (ok) mask |= 1u << k;
        if (lane == 0) out[1 + k] = ok ? dg[1] : 0;
    }
    if (lane == 0) { out[0] = (int)mask; if (mask) atomicAdd(total, 3 * __popc(mask)); }      // columns in fast groups
}

// +0.0 outside `lanes` (an absent entry must not see what x holds there -- Inf, NaN -- and contributes +-0.0)
__device__ __forceinline__ double zero_outside(double v, unsigned long long lanes) {
    unsigned lo, hi;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(lo) : "v"((unsigned)__double2loint(v)), "s"(lanes));
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(hi) : "v"((unsigned)__double2hiint(v)), "s"(lanes));
    return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ float zero_outside(float v, unsigned long long lanes) { return keep_lanes(v, lanes); }

template <typename V>
__global__ __launch_bounds__(256)
void sell8v_runs_kernel(long long n, long long nslices, V alpha, int append, int w,
        const char *__restrict__ pool, const int *__restrict__ deltas, const V *__restrict__ values,
        const int *__restrict__ csr_ptr, const int *__restrict__ csr_col, const V *__restrict__ csr_val,
        const V *__restrict__ x, V *__restrict__ y, trav_dev trav, const int *__restrict__ blocks, const int *__restrict__ desc, long long x_last)
{
    constexpr int G = RUNS_GROUPS;
    typedef typename vec2<V>::type V2;
    __shared__ int s_delta[256];
    __shared__ V s_value[256];
    s_delta[threadIdx.x] = deltas[threadIdx.x];
    s_value[threadIdx.x] = values[threadIdx.x];
    __syncthreads();

    // slices dealt to the XCDs as eight contiguous ranges (workgroup b runs on XCD b % 8) unless the matrix brought a traversal of its own: the
    // planes above and below a slice are then the same XCD's recent slices -- dealt round-robin, every L2 fetched all of x (HBM read 0.61 GB
    // for 0.26 GB of x at 320^3, profiles/r06_sq_spmv.txt)
    long long s;
    if (!trav.order && trav.chunk == 0) {
        const long long per = (nslices + 7) / 8, q = blockIdx.x >> 3;
        s = (long long)(blockIdx.x & 7u) * per + q;
        if (q >= per || s >= nslices) return;
    } else {
        s = traversal_block(trav, nslices);
        if (s < 0) return;
    }
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const long long i = s * S8_ROWS + 2 * t;
    const long long r_lo = s * S8_ROWS + 128 * wave, r_hi = r_lo + 127;         // the wave's rows
    const int wp = (w + 1) / 2;
    const long long sb = (long long)__builtin_amdgcn_readfirstlane(blocks[s]);
    const unsigned *cw = reinterpret_cast<const unsigned *>(pool + sb * ((long long)wp * 2048)) + t;
    const unsigned *vw = cw + wp * 256;
    const int *dw = desc + (sb * 4 + wave) * RUNS_STRIDE;                         // uniform: scalar loads
    unsigned tmask = (unsigned)__builtin_amdgcn_readfirstlane(dw[0]);

    // every fast group's elements of x, requested before anything is used
    V2 Pk[G];
    V Ek[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
        if ((tmask >> k) & 1u) {                                                       // uniform
            const long long dc = (long long)__builtin_amdgcn_readfirstlane(dw[1 + k]);
            if (r_lo + dc - 1 >= 0 && r_hi + dc + 1 <= x_last) {
                __builtin_memcpy(&Pk[k], x + (i + dc), sizeof(V2));                   // (4-byte alignment is enough for the wide load)
                Ek[k] = x[lane == 63 ? i + dc + 2 : r_lo + dc - 1];
            } else tmask &= ~(1u << k);                                                // the first / last waves of the matrix: the codes' path
        }
    }

    V sum[2] = {V(0), V(0)};
#pragma unroll
    for (int k = 0; k < G; ++k) {
        if (3 * k >= w) continue;                                                      // uniform
        if ((tmask >> k) & 1u) {                                                       // uniform
            const V2 P = Pk[k]; const V E = Ek[k];
            const V xs[3][2] = {{shift_from_lower_lane(P.y, E), P.x},                  // diagonal dc - 1: x[i + dc - 1], x[i + dc]
                                {P.x, P.y},                                            // diagonal dc
                                {P.y, shift_from_upper_lane(P.x, E)}};                 // diagonal dc + 1: x[i + dc + 1], x[i + dc + 2]
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int *col = dw + RUNS_HEAD + 6 * (3 * k + u);
                const unsigned long long m0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(col[1]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(col[0]);
                const unsigned long long m1 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(col[3]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(col[2]);
                V a;
                if constexpr (sizeof(V) == 8) a = __hiloint2double(__builtin_amdgcn_readfirstlane(col[5]), __builtin_amdgcn_readfirstlane(col[4]));
                else a = __int_as_float(__builtin_amdgcn_readfirstlane(col[4]));
                // (a variant of the group without the selects, for waves whose lanes hold all six entries, was slower: 0.255 -> 0.313 ms at 320^3)
                sum[0] += a * zero_outside(xs[u][0], m0);
                sum[1] += a * zero_outside(xs[u][1], m1);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int j = 3 * k + u;
                if (j >= w) continue;                                                  // uniform
                const int sh = 16 * (j & 1);
                const unsigned cword = cw[(j >> 1) * 256], vword = vw[(j >> 1) * 256];
                const unsigned c0 = (cword >> sh) & 255u, c1 = (cword >> (sh + 8)) & 255u;
                if (c0 < S8_FIRST_PAD) sum[0] += s_value[(vword >> sh) & 255u] * x[i + s_delta[c0]];
                if (c1 < S8_FIRST_PAD) sum[1] += s_value[(vword >> (sh + 8)) & 255u] * x[i + 1 + s_delta[c1]];
            }
        }
    }
    if (csr_ptr) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (i + q < n)
                for (int j = csr_ptr[i + q], e = csr_ptr[i + q + 1]; j < e; ++j) sum[q] += csr_val[j] * x[csr_col[j]];
    }
    store_pair<V>(n, i, alpha, append, sum, y, trav);
}

// ---------------------------------------------------------------------------
// PAIR kernels (round 2; the default for W <= 8): one 16-byte gather per lane and
// ELL column instead of two 8-byte gathers.
//
// What bounds these products is the number of vector-memory instructions a wave
// issues, not HBM and not latency (profiles/r02_sell8v_ablation.json: with all 14
// gathers pointed at ONE cache line the value-coded product takes the same
// 0.90 ms; with seven 16-byte gathers it takes 0.77 ms; a resident, software-
// pipelined grid is slower).  A lane owns rows 2t and 2t+1; when both rows hold
// the same diagonal d in ELL column j, x[2t+d] and x[2t+1+d] are adjacent: ONE
// 16-byte load (4-byte alignment is enough for global_load_dwordx4).  The fill
// kernels make that the normal case:
//   * the entries of the two rows of a lane are ALIGNED by diagonal (a two-pointer
//     merge of the two ascending diagonal lists; each row keeps its entries in
//     their order, padding is inserted where the partner has a diagonal the row
//     lacks) whenever the merged list fits the ELL width -- e.g. the identity row
//     at a grid boundary sits in the column where its interior neighbour has its
//     diagonal entry;
//   * a padding entry next to a real one is code 255 ("the 16-byte load of the
//     partner may cover me") unless that load would leave x -- the partner is in
//     column 0 resp. the last column -- then it is code 254.
// Tried on top of this and dropped (round 2, with the slice dictionary in place): three columns on consecutive diagonals
// d, d+1, d+2 -- the -1 / 0 / +1 entries of a stencil row -- need x[i+d .. i+d+3], which the loads of the OUTER two
// columns already hold, so the middle load was skipped per wave (one gather in seven).  Bit-identical, but 0.68 -> 0.84 ms:
// the bookkeeping (per-column ballots, the raw pairs kept apart from the masked values) costs far more than the load.
// Also tried: the -1 / +1 columns taken from the centre column's load by lane shifts, the two wave-edge values by scalar
// loads (two gathers in seven; the ablation harness promised 0.642 -> 0.593 ms): 4-5 % in this kernel before any of the
// guards a correct version needs -- not pursued.
// Per column a lane therefore does one unconditional 16-byte load when its codes are
// {d, d}, {d, 255}, {255, d} or both padding (then from a harmless address), and
// falls back to one 8-byte load per real entry otherwise (a rarely taken branch).
// Values gathered for padding entries are replaced by 0 before use, their matrix
// value is 0: sum + (+-0) == sum bit for bit (sum starts at +0 and cannot become
// -0 in round-to-nearest), so every row still adds its own products in their
// stored order: bit-identical to CSR.
// ---------------------------------------------------------------------------

template <typename V, int W, bool VCODED, bool DICT>
__global__ __launch_bounds__(256)
void sell8_pair_kernel(long long n, long long nslices, V alpha, int append,
        const char *__restrict__ buf, const int *__restrict__ deltas, const V *__restrict__ values,
        const int *__restrict__ csr_ptr, const int *__restrict__ csr_col, const V *__restrict__ csr_val,
        const V *__restrict__ x, V *__restrict__ y, trav_dev trav, const char *__restrict__ pool, const int *__restrict__ blocks)
{
    constexpr int WP = (W + 1) / 2;
    constexpr long long SLICE = VCODED ? (long long)WP * 2048 : ((long long)WP * 1024 + (long long)W * S8_ROWS * (long long)sizeof(V));
    typedef typename vec2<V>::type V2;
    __shared__ int s_delta[256];
    __shared__ V s_value[VCODED ? 256 : 1];
    // Order of the loads.  The code words are written first in this source, but they are only used after the early
    // return below, so the compiler SINKS them there: the machine code stages the tables, passes the barrier and only then
    // loads the slice's codes.  That order is the fast one (tools/r02_spmv_ab.py, profiles/r02_spmv_ab_staging_order.json,
    // same box, value codes / stored values): 0.789 / 1.868 ms as compiled; 0.829 / 1.893 ms with every load issued before
    // the barrier (no early return: codes and tables in flight together); 0.825 / 1.902 ms with a private 64-entry table
    // per wave and no barrier at all.  Shortening the chain tables -> barrier -> codes -> look-ups -> gathers -> store does
    // NOT help: the streaming loads issued early sit in the same queues the x gathers need.  (The CSR kernel, which is
    // bound by its stream, gains from the opposite: spmv.hip, second form.)  Also tried and dropped: tables of <= 64 entries
    // kept in registers and looked up with lane permutes (0.819 ms -- ds_bpermute costs more than the LDS reads it replaces).
    const long long s = traversal_block(trav, nslices);
    const long long sl = s < 0 ? 0 : s;                       // holes of the strip order: load slice 0, store nothing
    const int t = threadIdx.x;
    const long long i = sl * S8_ROWS + 2 * t;
    // DICT (slice dictionary, below): the CODES of slice s are block blocks[s] of a small pool of distinct code blocks; the
    // pool lives in L1 / L2, so its loads are plain (cached) ones -- streamed codes keep their non-temporal loads.  Stored
    // values (VCODED = false) stay in the slice.
    constexpr long long CODE_BYTES = VCODED ? (long long)WP * 2048 : (long long)WP * 1024;
    const unsigned *cw = reinterpret_cast<const unsigned *>(DICT ? pool + (long long)blocks[sl] * CODE_BYTES : buf + sl * SLICE) + t;
    unsigned c[WP], vc[VCODED ? WP : 1];
#pragma unroll
    for (int jp = 0; jp < WP; ++jp) {
        if constexpr (DICT) {
            c[jp] = cw[jp * 256];
            if constexpr (VCODED) vc[jp] = cw[(WP + jp) * 256];
        } else {
            c[jp] = __builtin_nontemporal_load(cw + jp * 256);
            if constexpr (VCODED) vc[jp] = __builtin_nontemporal_load(cw + (WP + jp) * 256);
        }
    }
    V2 v[VCODED ? 1 : W];
    if constexpr (!VCODED) {
        const V *vp = reinterpret_cast<const V *>(buf + sl * SLICE + (long long)WP * 1024) + 2 * t;
#pragma unroll
        for (int j = 0; j < W; ++j) v[j] = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(vp + j * S8_ROWS));
    }
    s_delta[threadIdx.x] = deltas[threadIdx.x];
    if constexpr (VCODED) s_value[threadIdx.x] = values[threadIdx.x];
    __syncthreads();
    if (s < 0) return;
    // table look-ups for every column first, then every gather
    int d[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const unsigned c0 = (c[j >> 1] >> (16 * (j & 1))) & 255u, c1 = (c[j >> 1] >> (16 * (j & 1) + 8)) & 255u;
        d[j] = s_delta[c0 < S8_PAD_UNSAFE ? c0 : c1];
    }
    V xv[W][2];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const unsigned c0 = (c[j >> 1] >> (16 * (j & 1))) & 255u, c1 = (c[j >> 1] >> (16 * (j & 1) + 8)) & 255u;
        const bool m0 = c0 < S8_PAD_UNSAFE, m1 = c1 < S8_PAD_UNSAFE;
        const bool pair = (c0 == c1) || (c0 == S8_PAD && m1) || (c1 == S8_PAD && m0);
        // the 16-byte load is skipped only when NO lane of the wave has a pair in this column (matrices without
        // band structure); lanes without one read the diagonal table instead
        const bool use16 = pair && (m0 || m1);
        V2 p = {V(0), V(0)};
        if (__builtin_amdgcn_ballot_w64(use16) != 0) {
            const V *px = use16 ? x + (i + d[j]) : reinterpret_cast<const V *>(deltas);
            __builtin_memcpy(&p, px, sizeof(V2));
        }
        xv[j][0] = p.x; xv[j][1] = p.y;
        if (!pair) {                       // different diagonals in one lane, or a 16-byte load that would leave x
            if (m0) xv[j][0] = x[i + s_delta[c0]];
            if (m1) xv[j][1] = x[i + 1 + s_delta[c1]];
        }
        xv[j][0] = m0 ? xv[j][0] : V(0);
        xv[j][1] = m1 ? xv[j][1] : V(0);
    }
    V sum[2] = {V(0), V(0)};
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int sh = 16 * (j & 1) + 8 * q;
            V a;
            if constexpr (VCODED) a = s_value[((c[j >> 1] >> sh) & 255u) < S8_PAD_UNSAFE ? (vc[j >> 1] >> sh) & 255u : 255u];   // entry 255 is 0.0
            else a = v[j][q];                                                                                              // stored as 0 for padding
            sum[q] += a * xv[j][q];
        }
    if (csr_ptr) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (i + q < n)
                for (int j = csr_ptr[i + q], e = csr_ptr[i + q + 1]; j < e; ++j) sum[q] += csr_val[j] * x[csr_col[j]];
    }
    store_pair<V>(n, i, alpha, append, sum, y, trav);
}

// ---------------------------------------------------------------------------
// HALO role of the pair product (round 6; halo.hpp, the PULL form): one device's whole product step in ONE launch for ANY matrix stored
// with diagonal codes (a general banded operator: a coefficient per face), not only the grid storages.  The stored strip has an empty
// ghost range of H.halo rows in front of / behind the device's rows where it has a neighbour; rows and columns are numbered in that
// stored form.  Slices whose diagonals stay inside the device's own rows run the pair product as it is (x and y moved by the ghost
// range); the slices within `reach` of an end -- dispatched LAST -- wait for the neighbour's "x is final" flag and take every element
// of x through a translation ghost-below / own / ghost-above (two 8-byte requests per column instead of one 16-byte pair: a pair may
// straddle the seam).  Same codes, same order of the sums: the BITS of the one-device product.
template <typename V, int W, bool VCODED, bool DICT>
__global__ __launch_bounds__(256)
void sell8_pair_halo_kernel(long long own_rows, V alpha, int append,
        const char *__restrict__ buf, const int *__restrict__ deltas, const V *__restrict__ values,
        const V *__restrict__ x, V *__restrict__ y, const char *__restrict__ pool, const int *__restrict__ blocks, halo_dev H)
{
    constexpr int WP = (W + 1) / 2;
    constexpr long long SLICE = VCODED ? (long long)WP * 2048 : ((long long)WP * 1024 + (long long)W * S8_ROWS * (long long)sizeof(V));
    constexpr long long CODE_BYTES = VCODED ? (long long)WP * 2048 : (long long)WP * 1024;
    typedef typename vec2<V>::type V2;
    __shared__ int s_delta[256];
    __shared__ V s_value[VCODED ? 256 : 1];
    __shared__ int s_flag;
    const int t = threadIdx.x;
    // own slices [0, ns); the first / last `edge` of them reach into a ghost range.  Dispatch order: the inner ones, then the lower
    // edge, then the upper edge (a workgroup that waits for a flag holds its CU slot: it must not start before the ones that do not)
    const long long hlo = (long long)H.z0 * H.halo;                       // rows of the lower ghost range (0: no neighbour below)
    const long long ns = (own_rows + S8_ROWS - 1) / S8_ROWS;
    const long long edge = ((long long)H.lo_planes + S8_ROWS - 1) / S8_ROWS;      // H.lo_planes: the reach of the diagonals in rows (comm.hip)
    const long long elo = H.lo ? (edge < ns ? edge : ns) : 0, ehi = H.hi ? (edge < ns - elo ? edge : ns - elo) : 0;
    const long long inner = ns - elo - ehi;
    long long so = blockIdx.x;                                            // -> own slice
    bool need_lo = false, need_hi = false;
    if (so < inner) so += elo;
    else if (so < inner + elo) { so -= inner; need_lo = true; }
    else { so = ns - ehi + (so - inner - elo); need_hi = true; }
    if (H.lo && H.hi && ns < 2 * edge) { need_lo = need_lo || so < edge; need_hi = need_hi || so >= ns - edge; }      // (strips shorter than two reaches)
    // the step number (uncached: the neighbours' flags are compared with it) is read by the workgroups that need it only -- the ones
    // that read a ghost range and workgroup 0, which raises "x is final": loads return in order, and the other 32 000 workgroups of a
    // 16 Mi-row strip would wait for that word before their first code arrives.  The same workgroups are the ones halo_finish counts.
    const bool edge_wg = need_lo || need_hi;                              // uniform
    const bool counted = edge_wg || blockIdx.x == 0;
    const unsigned n_counted = (unsigned)(elo + ehi) + (inner > 0 ? 1u : 0u);
    const unsigned long long step = counted ? *H.step : 0ull;
    halo_announce(H, step);
    const long long sl = so + hlo / S8_ROWS;                              // slice of the stored strip
    const long long i = sl * S8_ROWS + 2 * t;                             // row, stored numbering
    const unsigned *cw = reinterpret_cast<const unsigned *>(DICT ? pool + (long long)blocks[sl] * CODE_BYTES : buf + sl * SLICE) + t;
    unsigned c[WP], vc[VCODED ? WP : 1];
#pragma unroll
    for (int jp = 0; jp < WP; ++jp) {
        if constexpr (DICT) { c[jp] = cw[jp * 256]; if constexpr (VCODED) vc[jp] = cw[(WP + jp) * 256]; }
        else { c[jp] = __builtin_nontemporal_load(cw + jp * 256); if constexpr (VCODED) vc[jp] = __builtin_nontemporal_load(cw + (WP + jp) * 256); }
    }
    V2 v[VCODED ? 1 : W];
    if constexpr (!VCODED) {
        const V *vp = reinterpret_cast<const V *>(buf + sl * SLICE + (long long)WP * 1024) + 2 * t;
#pragma unroll
        for (int j = 0; j < W; ++j) v[j] = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(vp + j * S8_ROWS));
    }
    s_delta[t] = deltas[t];
    if constexpr (VCODED) s_value[t] = values[t];
    __syncthreads();
    bool ghosts_ok = true;
    if (edge_wg) ghosts_ok = halo_wait(H, step, need_lo && H.lo, need_hi && H.hi, &s_flag);
    const V *xe = x - hlo;                                                // x in stored numbering (own range only)
    V xv[W][2];
    if (!edge_wg) {
        int d[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const unsigned c0 = (c[j >> 1] >> (16 * (j & 1))) & 255u, c1 = (c[j >> 1] >> (16 * (j & 1) + 8)) & 255u;
            d[j] = s_delta[c0 < S8_PAD_UNSAFE ? c0 : c1];
        }
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const unsigned c0 = (c[j >> 1] >> (16 * (j & 1))) & 255u, c1 = (c[j >> 1] >> (16 * (j & 1) + 8)) & 255u;
            const bool m0 = c0 < S8_PAD_UNSAFE, m1 = c1 < S8_PAD_UNSAFE;
            const bool pair = (c0 == c1) || (c0 == S8_PAD && m1) || (c1 == S8_PAD && m0);
            const bool use16 = pair && (m0 || m1);
            V2 p = {V(0), V(0)};
            if (__builtin_amdgcn_ballot_w64(use16) != 0) {
                const V *px = use16 ? xe + (i + d[j]) : reinterpret_cast<const V *>(deltas);
                __builtin_memcpy(&p, px, sizeof(V2));
            }
            xv[j][0] = p.x; xv[j][1] = p.y;
            if (!pair) {
                if (m0) xv[j][0] = xe[i + s_delta[c0]];
                if (m1) xv[j][1] = xe[i + 1 + s_delta[c1]];
            }
            xv[j][0] = m0 ? xv[j][0] : V(0);
            xv[j][1] = m1 ? xv[j][1] : V(0);
        }
    } else {
        const long long own_end = hlo + own_rows;
        const V *glo = reinterpret_cast<const V *>(H.lo), *ghi = reinterpret_cast<const V *>(H.hi);
        auto at = [&](long long k) -> V {                                 // element k of x in stored numbering
            if (k < hlo) return (glo && ghosts_ok && k >= 0) ? glo[k] : V(__builtin_nan(""));
            if (k >= own_end) return (ghi && ghosts_ok && k - own_end < (long long)H.halo) ? ghi[k - own_end] : V(__builtin_nan(""));
            return xe[k];
        };
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const unsigned c0 = (c[j >> 1] >> (16 * (j & 1))) & 255u, c1 = (c[j >> 1] >> (16 * (j & 1) + 8)) & 255u;
            xv[j][0] = c0 < S8_PAD_UNSAFE ? at(i + s_delta[c0]) : V(0);
            xv[j][1] = c1 < S8_PAD_UNSAFE ? at(i + 1 + s_delta[c1]) : V(0);
        }
    }
    V sum[2] = {V(0), V(0)};
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int sh = 16 * (j & 1) + 8 * q;
            V a;
            if constexpr (VCODED) a = s_value[((c[j >> 1] >> sh) & 255u) < S8_PAD_UNSAFE ? (vc[j >> 1] >> sh) & 255u : 255u];
            else a = v[j][q];
            sum[q] += a * xv[j][q];
        }
    store_pair<V>(hlo + own_rows, i, alpha, append, sum, y - hlo);
    halo_finish(H, step, counted, n_counted);
}

// ---------------------------------------------------------------------------
// The MARCH product (round 3): the pair product with the x window of a slice staged in LDS and carried from slice to slice.
//
// With its codes on chip (slice dictionary) the pair kernel is bound by what it pulls through L1: seven 16-byte gathers
// per lane and slice, 195 L1 accesses per wave, TA busy 70 % (profiles/r02_sq_summary.txt) -- 0.70 ms against 0.42 ms for a
// plain copy of x to y.  But the NEAR diagonals of a banded matrix (|d| <= ~1000: the -n, -1, 0, +1, +n taps of a grid
// operator) read one contiguous window of x per slice, x[i0 + lo .. i0 + 511 + hi], and the window of the NEXT slice is the
// same window moved by 512 elements.  So a workgroup owns a RUN of consecutive slices and keeps that window in an LDS ring:
//   per slice ONE coalesced 16-byte load per lane brings the 512 new elements (requested one slice ahead, written into the
//   ring slot the previous slice has left), the FAR diagonals (+-n^2: up to two) are requested one slice ahead the same way
//   and parked in lane-private LDS slots, and every column of the slice is one LDS read at "base + position".
//   The codes of a slice are decoded again only when its block number differs from the previous slice's: per lane the two
//   matrix values, per wave the lane masks of the valid entries and where each column reads.
// Everything is still driven by the codes; same products, same order => bit-identical to the pair kernel and to the host
// loop over the CSR arrays (tests/test_gpu_spmv.py).  The host picks this kernel when the matrix has a slice dictionary whose block numbers
// rarely change from slice to slice and the near diagonals fit the LDS budget (vexhip_sell8_march_plan); otherwise the pair
// kernel runs as before.
//
// History at 512^3 (profiles/r03_march_ab_v*.json, r03_march2_*.json; pair kernel 0.69-0.72 ms, torch's copy of x to y 0.423 ms
// on the same boxes):
//   v1  decode per slice as the pair kernel does, far diagonals gathered                                    0.674 ms
//   v2  + far diagonals one slice ahead into registers (114 registers)                                      0.84
//   v3  + the frontier lines touched 1..16 slices ahead instead                                             0.78-0.86
//   v4  decode hoisted out of the slice loop                                                                0.589
//   v6  straight-line slice: one ds_read2 per column, far diagonals parked in LDS                           0.550
//   v7  lane masks in scalar registers                                                                      0.547   (150 vector + 91 scalar instructions per wave and slice)
//   v8  this kernel: mirrored ring (one add per column address), high-word masks, a hot loop without bounds arithmetic
//       or scalar loads, arguments of the cold paths read from the kernarg segment where they are used              0.484   (83 + 43)
//       + the prologue of a run as ONE batch of loads                                                       0.466
// At 0.466 ms the memory system is saturated: requests two slices ahead change nothing with four workgroups per CU (0.465) and
// help with three (0.493 -> 0.480) or two (0.605 -> 0.555); the product moves 2.37 GB (PMC; 2.15 priced) at 5.1 TB/s, the rate
// at which the same box copies x to y.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void march_run(const trav_dev &t, long long nblocks, int R, unsigned mb, long long &first, int &count) {
    long long c;
    if (t.chunk > 0) {       // strip order (traversal.hpp) in runs of R slices: R divides the strip length
        const unsigned k = mb & 7u, q = mb >> 3;
        const unsigned chunk = (unsigned)t.chunk, planes = (unsigned)t.planes, rpc = chunk / (unsigned)R;
        const unsigned r = q / rpc, ri = q - r * rpc;
        const unsigned tile = r / planes, p = r - tile * planes;
        const long long l = (long long)tile * (8 * chunk) + k * chunk + ri * (unsigned)R;
        first = (long long)p * t.plane_blocks + l;
        c = t.plane_blocks - l; if (c > R) c = R;
        if (c > nblocks - first) c = nblocks - first;
    } else {
        first = (long long)mb * R;
        c = nblocks - first; if (c > R) c = R;
    }
    count = c > 0 ? (int)c : 0;
}

template <typename V>
__device__ __forceinline__ typename vec2<V>::type load_pair_clamped(const V *__restrict__ x, long long g, long long x_last) {
    typename vec2<V>::type v = {V(0), V(0)};
    if (g >= 0 && g + 1 <= x_last) __builtin_memcpy(&v, x + g, sizeof(v));
    else {
        if (g >= 0 && g <= x_last) v.x = x[g];
        if (g + 1 >= 0 && g + 1 <= x_last) v.y = x[g + 1];
    }
    return v;
}

// What keeps the instruction count of a slice down (the first seven versions were bound by instruction issue):
//   * the ring is no power of two but span + 2 slices (span = bytes the near diagonals cover, rounded to whole slices) and
//     carries a MIRROR of its first `span` bytes behind its end: a column's pair sits at  window base + offset  without a
//     wrap, the window base is one scalar per slice, a column address is ONE vector add;
//   * masking only the HIGH word of a double an invalid lane "covers": (+0.0) * (a number whose exponent field is 0) is +0
//     whatever the low word holds -- one v_cndmask per value, its lane mask taken straight from a scalar register pair;
//   * slices whose loads lie inside x and whose rows lie inside y (all but the first / last plane's) take loads and the
//     store without any per-lane bounds arithmetic: scalar base + lane offset (the hot loop);
//   * no scalar load inside the hot loop (one in flight turns every partial wait for the LDS reads into a full one): how
//     many slices keep the code block is found once, on entry.
// LDS per workgroup at 512^3 fp64: ring 16 + mirror 8 + far slots 8 + tables 3 = 35 KiB, four workgroups per CU.
struct march_dev { int lo, hi, lo_e, span_b, cap_b, run, nfar, far0, far1; long long x_last; };

// v for the lanes of `lanes`; elsewhere a number a (+0.0) matrix value multiplies to +0: v_cndmask with the lane mask taken
// straight from a scalar register pair (the compiler, given a bool per lane, keeps it in a vector register and rebuilds the
// mask with two instructions per use)
template <typename V> __device__ __forceinline__ V masked_hi(V v, unsigned long long lanes);
template <> __device__ __forceinline__ double masked_hi<double>(double v, unsigned long long lanes) {
    unsigned hi = (unsigned)(__double_as_longlong(v) >> 32), rhi;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(rhi) : "v"(hi), "s"(lanes));
    return __longlong_as_double((long long)(((unsigned long long)rhi << 32) | (unsigned)__double_as_longlong(v)));
}
template <> __device__ __forceinline__ float masked_hi<float>(float v, unsigned long long lanes) {
    unsigned r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(__float_as_uint(v)), "s"(lanes));
    return __uint_as_float(r);
}

// Arguments the hot loop does not touch.  The kernel reads them from its kernarg segment WHERE they are used, through a
// pointer the optimiser cannot see through (cold()): hoisted to the top they would sit in ~25 scalar registers across the
// hot loop, which needs 28 for the lane masks alone -- the first build of this kernel spilled 50 scalars into vector lanes.
template <typename V>
struct march_cold {
    long long n, nslices;
    const int *deltas; const V *values;
    const int *csr_ptr, *csr_col; const V *csr_val;
    const char *pool;
    trav_dev trav;
};
template <typename V, int W>
__global__ __launch_bounds__(256)          // 120 registers, four workgroups per CU; (256, 5): 95 registers and 0.54 instead of 0.48 ms
void sell8_march_kernel(march_cold<V> cold_args /* first: offset 0 of the kernarg segment; read through cold() only */, V alpha, int append,
        const V *__restrict__ x, V *__restrict__ y, const int *__restrict__ blocks, march_dev mp)
{
    constexpr int WP = (W + 1) / 2;
    constexpr long long CODE_BYTES = (long long)WP * 2048;
    constexpr int VB = (int)sizeof(V);
    constexpr int SLB = S8_ROWS * VB;                          // bytes of x per slice
    typedef typename vec2<V>::type V2;
    typedef const __attribute__((address_space(4))) march_cold<V> *kernarg_ptr;
    extern __shared__ __align__(16) unsigned char s_march[];  // [ring: cap_b][mirror of its first span_b bytes][far 0: SLB][far 1: SLB][diagonal table: 1 KiB][value table]

    auto cold = [&]() -> const __attribute__((address_space(4))) march_cold<V> * {
        unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        return (kernarg_ptr)ka;
    };

    const int t = threadIdx.x;
    long long first; int count;
    {
        const auto *c = cold();
        const trav_dev trav = {nullptr, c->trav.chunk, c->trav.planes, c->trav.plane_blocks};   // the plan declines explicit orders
        march_run(trav, c->nslices, mp.run, blockIdx.x, first, count);
    }
    if (count <= 0) return;                                   // the whole workgroup: holes of the strip order

    const int capb = mp.cap_b, spanb = mp.span_b, farb = capb + spanb;
    const int hi2 = mp.lo_e + spanb / VB;                     // the window of a slice: x[i0 + lo_e .. i0 + 512 + hi2)
    const long long i00 = first * S8_ROWS;
    const unsigned lane_b = 2u * (unsigned)t * VB;
    // Iteration k requests the 512 elements slice k + 2 adds to the window and the far diagonals of slice k + 1, and stores
    // the rows of slice k.  [kA, kend): the iterations for which all of that lies inside x and y -- the hot loop.
    int kA = 0, kend = count;
    {
        const auto *c = cold();
        long long m = 2 * S8_ROWS + hi2, M = m;
        if (mp.nfar > 0) { const long long f = S8_ROWS + (long long)mp.far0; m = f < m ? f : m; M = f > M ? f : M; }
        if (mp.nfar > 1) { const long long f = S8_ROWS + (long long)mp.far1; m = f < m ? f : m; M = f > M ? f : M; }
        const long long a = -(i00 + m);
        if (a > 0) { const long long q = (a + S8_ROWS - 1) / S8_ROWS; kA = q > count ? count : (int)q; }
        const long long bb = mp.x_last - (S8_ROWS - 1) - M - i00;
        const long long kB = bb >= 0 ? bb / S8_ROWS + 1 : 0;
        const long long kS = c->n / S8_ROWS - first;
        if (kB < kend) kend = (int)kB;
        if (kS < kend) kend = kS > 0 ? (int)kS : 0;
        if (c->csr_ptr || (reinterpret_cast<unsigned long long>(y) & (2 * VB - 1)) != 0) kend = 0;
    }
    auto load512 = [&](long long gb) -> V2 {
        if (gb >= 0 && gb + S8_ROWS - 1 <= mp.x_last) { V2 v; __builtin_memcpy(&v, x + gb + 2 * t, sizeof(V2)); return v; }
        return load_pair_clamped<V>(x, gb + 2 * t, mp.x_last);
    };
    // The prologue of a run costs round trips, not bytes (the first build: three for the window, one for the registers, three
    // for block number -> codes -> tables; 11 % of the product at 32 slices per run).  Now: the block number, then ONE batch
    // of independent loads -- codes of the first block, the diagonal / value tables (into LDS, behind the far slots), the
    // window of the first slice (ring position 0; <= 4 pairs per lane, the plan sees to that), the 512 elements the second
    // slice adds and the far diagonals of the first slice -- then the decode from LDS.
    int *s_delta = reinterpret_cast<int *>(s_march + farb + 2 * SLB);
    V *s_value = reinterpret_cast<V *>(s_march + farb + 2 * SLB + 1024);
    int cur = blocks[first];
    unsigned cwv[2 * WP];
    V2 chunk = {V(0), V(0)}, f0 = {V(0), V(0)}, f1 = {V(0), V(0)};
    {
        const auto *c = cold();
        const int *deltas = c->deltas; const V *values = c->values;
        const unsigned *cw = reinterpret_cast<const unsigned *>(c->pool + (long long)cur * CODE_BYTES) + t;
#pragma unroll
        for (int u = 0; u < 2 * WP; ++u) cwv[u] = cw[u * 256];
        const int dl = deltas[t]; const V vl = values[t];
        const long long g0 = i00 + mp.lo_e;
        const int wpairs = (SLB + spanb) / (2 * VB);
        V2 w[4];
        const long long glo = g0 < i00 + mp.far0 ? g0 : i00 + mp.far0;
        long long ghi = i00 + 2 * S8_ROWS + hi2;                     // one past the last element any of these loads touches
        if (mp.nfar > 0 && i00 + S8_ROWS + mp.far0 > ghi) ghi = i00 + S8_ROWS + mp.far0;
        if (mp.nfar > 1 && i00 + S8_ROWS + mp.far1 > ghi) ghi = i00 + S8_ROWS + mp.far1;
        if (glo >= 0 && (mp.nfar < 2 || i00 + mp.far1 >= 0) && ghi - 1 <= mp.x_last) {      // uniform: nothing to clamp
#pragma unroll
            for (int u = 0; u < 4; ++u) { w[u] = V2{V(0), V(0)}; if (t + u * 256 < wpairs) __builtin_memcpy(&w[u], x + g0 + 2 * (t + u * 256), sizeof(V2)); }
            __builtin_memcpy(&chunk, x + i00 + S8_ROWS + hi2 + 2 * t, sizeof(V2));
            if (mp.nfar > 0) __builtin_memcpy(&f0, x + i00 + mp.far0 + 2 * t, sizeof(V2));
            if (mp.nfar > 1) __builtin_memcpy(&f1, x + i00 + mp.far1 + 2 * t, sizeof(V2));
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) { w[u] = V2{V(0), V(0)}; if (t + u * 256 < wpairs) w[u] = load_pair_clamped<V>(x, g0 + 2 * (t + u * 256), mp.x_last); }
            if (count > 1) chunk = load512(i00 + S8_ROWS + hi2);
            if (mp.nfar > 0) f0 = load512(i00 + mp.far0);
            if (mp.nfar > 1) f1 = load512(i00 + mp.far1);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (t + u * 256 < wpairs) *reinterpret_cast<V2 *>(s_march + 2 * (t + u * 256) * VB) = w[u];
        s_delta[t] = dl; s_value[t] = vl;
    }
    __syncthreads();

    bool slow = false;                                        // this wave, this block: per-entry loop
    int um[W];                                                // uniform per column: -1 = the ring (its address moves with the window), 0 = a far slot
    int cj[W];                                                // the lane's LDS byte address in column j relative to the window base (ring) / absolute (far slot)
    V a0[W], a1[W];                                           // the matrix values of the lane's two rows (0 for padding)
    unsigned long long v0[W], v1[W];                          // uniform: the lanes whose row 2t / 2t + 1 has an entry in column j
    // code words of a block (cwv) -> the per-column state above
    auto decode = [&]() {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const unsigned cword = cwv[j >> 1] >> (16 * (j & 1)), vword = cwv[WP + (j >> 1)] >> (16 * (j & 1));
            const unsigned c0 = cword & 255u, c1 = (cword >> 8) & 255u;
            const bool m0 = c0 < S8_PAD_UNSAFE, m1 = c1 < S8_PAD_UNSAFE, any = m0 || m1;
            const bool pair = (c0 == c1) || (c0 == S8_PAD && m1) || (c1 == S8_PAD && m0);
            const int d = any ? s_delta[m0 ? c0 : c1] : 0;
            // the wave's diagonal in this column: that of its first lane with an entry
            const unsigned long long have = __builtin_amdgcn_ballot_w64(any);
            const int du = have ? __builtin_amdgcn_readlane(d, __ffsll((long long)have) - 1) : 0;
            bad |= any && (!pair || d != du);
            const bool nearcol = du >= mp.lo && du <= mp.hi;
            const int slot = (mp.nfar > 0 && du == mp.far0) ? 0 : (mp.nfar > 1 && du == mp.far1) ? 1 : -1;
            bad |= have && !nearcol && slot < 0;
            um[j] = __builtin_amdgcn_readfirstlane(nearcol ? -1 : 0);   // an integer in one scalar register, not a condition the optimiser re-derives per slice
            cj[j] = nearcol ? (int)lane_b + (du - mp.lo_e) * VB : farb + (slot > 0 ? SLB : 0) + (int)lane_b;
            v0[j] = __builtin_amdgcn_ballot_w64(m0); v1[j] = __builtin_amdgcn_ballot_w64(m1);
            a0[j] = s_value[m0 ? (vword & 255u) : 255u];           // entry 255 is 0.0
            a1[j] = s_value[m1 ? ((vword >> 8) & 255u) : 255u];
        }
        slow = __builtin_amdgcn_ballot_w64(bad) != 0;
    };
    decode();
    int b = 0;                                                // ring position of the window of slice k

    // what arrived during slice k - 1 goes into the slot slice k - 1 has left (and into the mirror if that slot is mirrored);
    // this slice's far elements go into the lane's own slots
    auto park = [&](bool far, const V2 &c, const V2 &g0, const V2 &g1) {
        int wb = b + SLB + spanb; if (wb >= capb) wb -= capb;
        *reinterpret_cast<V2 *>(s_march + wb + lane_b) = c;
        if (wb < spanb) *reinterpret_cast<V2 *>(s_march + capb + wb + lane_b) = c;
        if (far) {
            *reinterpret_cast<V2 *>(s_march + farb + lane_b) = g0;
            *reinterpret_cast<V2 *>(s_march + farb + SLB + lane_b) = g1;
        }
    };
    auto body = [&](V (&sum)[2]) {
        V2 p[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const V *q = reinterpret_cast<const V *>(s_march + (cj[j] + (b & um[j])));   // b & um: scalar -- the window base for ring columns, 0 for far slots
            p[j].x = q[0]; p[j].y = q[1];                        // one ds_read2: the pair is adjacent, also across the end of the ring (mirror)
        }
#pragma unroll
        for (int j = 0; j < W; ++j) {
            // what a padding entry "covers" is replaced by a number with a zero exponent field (its matrix value is +0):
            // sum + (+0) == sum, whatever x holds there
            sum[0] += a0[j] * masked_hi<V>(p[j].x, v0[j]);
            sum[1] += a1[j] * masked_hi<V>(p[j].y, v1[j]);
        }
    };

    // ---- the hot loop: code block unchanged, everything inside x and y -- straight-line slices.  With far diagonals it always
    // carries two (a matrix with one: the second slot repeats the first) ----
    int k = 0;
    auto hot = [&](auto far_tag) {
        constexpr bool FAR = decltype(far_tag)::value;
        const char *xc = reinterpret_cast<const char *>(x + (i00 + (long long)(k + 2) * S8_ROWS + hi2));   // uniform running pointers
        const char *xf0 = reinterpret_cast<const char *>(x + (i00 + (long long)(k + 1) * S8_ROWS + mp.far0));
        const char *xf1 = reinterpret_cast<const char *>(x + (i00 + (long long)(k + 1) * S8_ROWS + (mp.nfar > 1 ? mp.far1 : mp.far0)));
        char *ys = reinterpret_cast<char *>(y + (i00 + (long long)k * S8_ROWS));
        // how many of the next slices keep this code block: one look at blocks[] per entry, no scalar load inside the loop
        // (a scalar load in flight turns every partial wait for the LDS reads into a full one)
        int kstop;
        {
            const int l = t & 63, left = kend - k;
            const bool same = l < left && blocks[first + k + l] == cur;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(same);
            const int len = ~m ? __builtin_ctzll(~m) : 64;           // >= 1: the caller has checked slice k
            kstop = k + len;
        }
        // one slice: park what was requested for it, request the same for a later slice into the same registers, multiply, store
        auto slice = [&](V2 &c, V2 &g0, V2 &g1) {
            unsigned lb = lane_b;
            asm volatile("" : "+v"(lb));                      // extended to 64 bits HERE: scalar base + lane offset addressing
            park(FAR, c, g0, g1);
            __builtin_memcpy(&c, xc + lb, sizeof(V2));
            if (FAR) { __builtin_memcpy(&g0, xf0 + lb, sizeof(V2)); __builtin_memcpy(&g1, xf1 + lb, sizeof(V2)); }
            V sum[2] = {V(0), V(0)};
            body(sum);
            V2 *yp = reinterpret_cast<V2 *>(ys + lb);
            V2 o; o.x = alpha * sum[0]; o.y = alpha * sum[1];
            if (append == 2) {                                // y = alpha A x + beta z (round 6): z and beta from the cold arguments, where y's element lies in z
                const auto *ca = cold();
                const V bz = (V)ca->trav.beta;
                const V2 old = *reinterpret_cast<const V2 *>(static_cast<const char *>(ca->trav.z) + (reinterpret_cast<const char *>(yp) - reinterpret_cast<const char *>(y)));
                o.x = bz * old.x + o.x; o.y = bz * old.y + o.y;
            } else if (append) { const V2 old = *yp; o.x = old.x + o.x; o.y = old.y + o.y; }
            __builtin_nontemporal_store(o, yp);               // y is written once and not re-read by this kernel
            b += SLB; if (b >= capb) b -= capb;
            xc += SLB; xf0 += SLB; xf1 += SLB; ys += SLB; ++k;
            __syncthreads();              // slice k is done with the ring: its oldest 512 elements may be overwritten
        };
        // (requests two slices ahead instead of one: 0.466 -> 0.465 ms with four workgroups per CU, 0.493 -> 0.480 with three --
        // with four the memory system is saturated; far diagonals two ahead and the window one: 0.472; far diagonals loaded
        // non-temporally (their last use by this workgroup, but not by its neighbours): 0.491.  Not built in.)
        do slice(chunk, f0, f1); while (k < kstop);
    };
    while (k < count) {
        if (k >= kA && k < kend && !slow && blocks[first + k] == cur) {
            if (mp.nfar > 0) hot(std::true_type{}); else hot(std::false_type{});
            continue;
        }
        // ---- a general slice: clamped loads, rows checked against n, the code block decoded if it is a new one ----
        const auto *c = cold();
        const long long i0 = i00 + (long long)k * S8_ROWS;
        park(mp.nfar > 0, chunk, f0, f1);
        if (k + 2 < count) chunk = load512(i0 + 2 * S8_ROWS + hi2);
        if (k + 1 < count) {
            if (mp.nfar > 0) f0 = load512(i0 + S8_ROWS + mp.far0);
            if (mp.nfar > 1) f1 = load512(i0 + S8_ROWS + mp.far1);
        }
        const int blk = blocks[first + k];
        const char *pool = c->pool;
        if (blk != cur) {                                        // uniform: a new code block -- load and decode it
            cur = blk;
            const unsigned *cw = reinterpret_cast<const unsigned *>(pool + (long long)blk * CODE_BYTES) + t;
#pragma unroll
            for (int u = 0; u < 2 * WP; ++u) cwv[u] = cw[u * 256];
            decode();
        }
        V sum[2] = {V(0), V(0)};
        const long long i = i0 + 2 * t;
        if (!slow) body(sum);
        else {
            const unsigned *cw = reinterpret_cast<const unsigned *>(pool + (long long)blk * CODE_BYTES) + t;
#pragma unroll 1
            for (int j = 0; j < W; ++j) {
                const unsigned cword = cw[(j >> 1) * 256] >> (16 * (j & 1)), vword = cw[(WP + (j >> 1)) * 256] >> (16 * (j & 1));
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned code = (cword >> (8 * q)) & 255u;
                    if (code < S8_PAD_UNSAFE) sum[q] += s_value[(vword >> (8 * q)) & 255u] * x[i + q + s_delta[code]];
                }
            }
        }
        const long long n = c->n;
        if (const int *csr_ptr = c->csr_ptr) {
            const int *csr_col = c->csr_col; const V *csr_val = c->csr_val;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (i + q < n)
                    for (int j = csr_ptr[i + q], e = csr_ptr[i + q + 1]; j < e; ++j) sum[q] += csr_val[j] * x[csr_col[j]];
        }
        {
            const trav_dev tz = {nullptr, 0, 0, 0, append == 2 ? c->trav.z : nullptr, append == 2 ? c->trav.beta : 0.0};
            store_pair<V>(n, i, alpha, append == 1, sum, y, tz);
        }
        b += SLB; if (b >= capb) b -= capb;
        ++k;
        __syncthreads();
    }
}

// distinct value bit patterns of the ELL part: gset = HASH_SLOTS words (all-ones = empty), info as delta_collect_kernel
template <typename B>
__device__ __forceinline__ bool vset_insert(B *set, int slots, B v, bool *is_new) {
    const B empty = ~B(0);
    unsigned h = (unsigned)((unsigned long long)v * 0x9E3779B97F4A7C15ull >> 40) % (unsigned)slots;
    for (int probe = 0; probe < slots; ++probe) {
        B old = atomicCAS(&set[h], empty, v);
        if (old == empty) { if (is_new) *is_new = true; return true; }
        if (old == v) return true;
        h = (h + 1) % (unsigned)slots;
    }
    return false;
}

template <typename V, typename P>
__global__ __launch_bounds__(256)
void value_collect_kernel(long long n, int w, const P *__restrict__ ptr, const V *__restrict__ val,
        typename bits_of<V>::type *gset, int *info)
{
    typedef typename bits_of<V>::type B;
    __shared__ B s_set[LOCAL_SLOTS];
    __shared__ int s_over, s_count;
    for (int k = threadIdx.x; k < LOCAL_SLOTS; k += blockDim.x) s_set[k] = ~B(0);
    if (threadIdx.x == 0) { s_over = *(volatile int *)&info[1]; s_count = 0; }
    __syncthreads();
    // A matrix whose values are all different fills the set at once.  Stop at the 256th distinct value (a half-full
    // table: short probe sequences), tell the other workgroups through the global flag, and do not merge: with the
    // first version of this loop (probe until the 512-slot table is FULL, flag raised only at the end) the analysis of
    // the 512^3 variable-coefficient matrix took 143 ms (profiles/r02_*cpp_kernel_stats.csv).
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (*(volatile int *)&s_over || *(volatile int *)&info[1]) break;
        const P b = ptr[i], e = ptr[i + 1];
        B last = ~B(0);
        for (int j = 0; j < w && b + j < e; ++j) {
            B bits; V v = val[b + j];
            __builtin_memcpy(&bits, &v, sizeof(B));
            if (bits == ~B(0)) { s_over = 1; continue; }          // the one pattern that cannot be stored (a NaN payload)
            if (bits == last) continue;
            last = bits;
            bool fresh = false;
            if (!vset_insert<B>(s_set, LOCAL_SLOTS, bits, &fresh)) s_over = 1;
            else if (fresh && atomicAdd(&s_count, 1) >= 255) { s_over = 1; atomicExch(&info[1], 1); }
        }
    }
    __syncthreads();
    if (s_over) { if (threadIdx.x == 0) atomicExch(&info[1], 1); return; }
    for (int k = threadIdx.x; k < LOCAL_SLOTS; k += blockDim.x) {
        if (s_set[k] == ~B(0)) continue;
        bool is_new = false;
        if (!vset_insert<B>(gset, HASH_SLOTS, s_set[k], &is_new)) atomicExch(&info[1], 1);
        else if (is_new) atomicAdd(&info[0], 1);
    }
    if (threadIdx.x == 0 && s_over) atomicExch(&info[1], 1);
}

// vtable: sorted value bit patterns (nvalues valid entries)
// Diagonals, values and the largest column of the ELL part in ONE pass over the CSR arrays (round 3: delta_collect +
// value_collect + ell_max_col read ptr / col / val three times, 12 ms of the 512^3 set-up).  Same sets, same overflow rules
// as the two collect kernels.  A workgroup takes 256 consecutive rows at a time; their entries are one contiguous piece of
// col / val, which the workgroup copies into LDS with coalesced loads (a lane reading ITS row's entries reads 4 / 8 bytes at
// a stride of 28 / 56: 1.25 TB/s, 9.3 ms at 512^3) and every lane then walks its row there.  A piece that does not fit
// (rows with long tails) is read from global memory as before.  (The copy alone: 10.3 ms; with the once-per-lane inserts
// below: 4.9 ms; a wave-private copy without the workgroup barriers: 5.0 ms -- the same.)
constexpr int ANALYZE_CAP = 2560;
template <typename V, typename P>
__global__ __launch_bounds__(256)
void analyze_fused_kernel(long long n, int w, const P *__restrict__ ptr, const int *__restrict__ col, const V *__restrict__ val,
        int *gset_d, int *info_d, typename bits_of<V>::type *gset_v, int *info_v, int *max_col)
{
    typedef typename bits_of<V>::type B;
    __shared__ int s_set[LOCAL_SLOTS];
    __shared__ B s_vset[LOCAL_SLOTS];
    __shared__ int s_over, s_count, s_vover, s_vcount, s_vskip;
    __shared__ long long s_ptr[257];
    __shared__ int s_c[ANALYZE_CAP];
    __shared__ V s_v[ANALYZE_CAP];
    const int t = threadIdx.x;
    for (int k = t; k < LOCAL_SLOTS; k += blockDim.x) { s_set[k] = EMPTY; s_vset[k] = ~B(0); }
    if (t == 0) { s_over = *(volatile int *)&info_d[1]; s_count = 0; s_vover = *(volatile int *)&info_v[1]; s_vcount = 0; }
    __syncthreads();
    int m = -1;
    int seen_d[8]; B seen_v[8];                                  // what this lane inserted last at entry position j (both sets only grow)
#pragma unroll
    for (int j = 0; j < 8; ++j) { seen_d[j] = EMPTY; seen_v[j] = ~B(0); }
    for (long long r0 = (long long)blockIdx.x * 256; r0 < n; r0 += (long long)gridDim.x * 256) {
        const int rows = (int)(n - r0 < 256 ? n - r0 : 256);
        for (int k = t; k <= rows; k += 256) s_ptr[k] = (long long)ptr[r0 + k];
        // more values than codes (a matrix with a coefficient per face says so within its first rows): the rest of the pass is
        // about the diagonals and the largest column only -- the values, two thirds of the bytes, are not read any more
        // (variable-coefficient 512^3: set-up 18.0 -> 17.1 ms -- the pass is not bound by its bytes either)
        if (t == 0) s_vskip = (*(volatile int *)&s_vover || *(volatile int *)&info_v[1]) ? 1 : 0;
        __syncthreads();
        const long long e0 = s_ptr[0], cnt = s_ptr[rows] - e0;
        const bool staged = cnt <= ANALYZE_CAP;                       // uniform
        const bool vskip = s_vskip != 0;                              // uniform
        if (staged) {
            if (vskip) for (int k = t; k < (int)cnt; k += 256) s_c[k] = col[e0 + k];
            else for (int k = t; k < (int)cnt; k += 256) { s_c[k] = col[e0 + k]; s_v[k] = val[e0 + k]; }
        }
        __syncthreads();
        if (t < rows) {
            const long long i = r0 + t, b = s_ptr[t], e = s_ptr[t + 1];
            const bool dstop = *(volatile int *)&s_over || *(volatile int *)&info_d[1];
            const bool vstop = vskip || *(volatile int *)&s_vover || *(volatile int *)&info_v[1];
            int last = EMPTY; B vlast = ~B(0);
            const int nj = (int)(e - b < (long long)w ? e - b : (long long)w), off = (int)(b - e0);
            for (int j = 0; j < nj; ++j) {
                const int c = staged ? s_c[off + j] : col[b + j];
                const V v = vstop ? V(0) : (staged ? s_v[off + j] : val[b + j]);
                m = c > m ? c : m;
                if (!dstop) {
                    const long long dl = (long long)c - i;
                    if (dl <= INT_MIN || dl > INT_MAX) s_over = 1;
                    else if ((int)dl != last && !(j < 8 && (int)dl == seen_d[j])) {
                        // (a lane of a structured matrix meets the same diagonal at the same place of every row: what it has
                        // inserted once it does not insert again -- 64 lanes inserting the same 7 diagonals per row were
                        // 64-way conflicts on the LDS set: the kernel was bound by them, not by its loads)
                        last = (int)dl;
                        if (j < 8) seen_d[j] = last;
                        bool fresh = false;
                        if (!set_insert(s_set, LOCAL_SLOTS, last, &fresh)) s_over = 1;
                        else if (fresh && atomicAdd(&s_count, 1) >= 254) { s_over = 1; atomicExch(&info_d[1], 1); }
                    }
                }
                if (!vstop) {
                    B bits; __builtin_memcpy(&bits, &v, sizeof(B));
                    if (bits == ~B(0)) s_vover = 1;
                    else if (bits != vlast && !(j < 8 && bits == seen_v[j])) {
                        vlast = bits;
                        if (j < 8) seen_v[j] = bits;
                        bool fresh = false;
                        if (!vset_insert<B>(s_vset, LOCAL_SLOTS, bits, &fresh)) s_vover = 1;
                        else if (fresh && atomicAdd(&s_vcount, 1) >= 255) { s_vover = 1; atomicExch(&info_v[1], 1); }
                    }
                }
            }
        }
        __syncthreads();                                               // the next 256 rows overwrite s_ptr / s_c / s_v
    }
    for (int o = 32; o > 0; o >>= 1) { const int u = __shfl_down(m, o, 64); m = u > m ? u : m; }
    if ((t & 63) == 0 && m >= 0) atomicMax(max_col, m);
    __syncthreads();
    if (s_over) { if (t == 0) atomicExch(&info_d[1], 1); }
    else
        for (int k = t; k < LOCAL_SLOTS; k += blockDim.x) {
            if (s_set[k] == EMPTY) continue;
            bool is_new = false;
            if (!set_insert(gset_d, HASH_SLOTS, s_set[k], &is_new)) atomicExch(&info_d[1], 1);
            else if (is_new) atomicAdd(&info_d[0], 1);
        }
    if (s_vover) { if (t == 0) atomicExch(&info_v[1], 1); }
    else
        for (int k = t; k < LOCAL_SLOTS; k += blockDim.x) {
            if (s_vset[k] == ~B(0)) continue;
            bool is_new = false;
            if (!vset_insert<B>(gset_v, HASH_SLOTS, s_vset[k], &is_new)) atomicExch(&info_v[1], 1);
            else if (is_new) atomicAdd(&info_v[0], 1);
        }
}

// The same for the value-coded storage, which does not need the values themselves: the lane that copies an entry looks its
// value up in the sorted table there and then and leaves a 2-byte code (0xffff: not in the table) -- 6 instead of 12 bytes
// of LDS per entry, 27 instead of 53 KiB per workgroup, twice the waves per CU for a kernel whose waves are parked 73 % of
// their time (profiles/r03_sq_setup.txt).
template <typename V>
__device__ __forceinline__ bool wave_stage_codes(const int *__restrict__ col, const V *__restrict__ val, long long b0_lane, long long e1_lane,
        int *s_c, unsigned short *s_vc, const typename bits_of<V>::type *s_vtable, int nvalues, long long &e0)
{
    typedef typename bits_of<V>::type B;
    const int lane = threadIdx.x & 63;
    const unsigned long long real = __ballot(e1_lane > 0 || b0_lane > 0);
    long long lo_ = b0_lane, hi_ = e1_lane;
    e0 = __shfl(lo_, 0, 64);
    const int lastl = real ? 63 - __builtin_clzll(real) : 0;
    const long long end = __shfl(hi_, lastl, 64);
    const long long cnt = end - e0;
    if (cnt > WAVE_CAP || cnt < 0) return false;                     // uniform
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();       // the previous trip's reads of this region are done
    for (int k = lane; k < (int)cnt; k += 64) {
        s_c[k] = col[e0 + k];
        const V v = val[e0 + k];
        B bits; __builtin_memcpy(&bits, &v, sizeof(B));
        int lo = 0, hi = nvalues;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (s_vtable[mid] < bits) lo = mid + 1; else hi = mid; }
        s_vc[k] = (lo < nvalues && s_vtable[lo] == bits) ? (unsigned short)lo : (unsigned short)0xffffu;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return true;
}

// STAGED (w <= 8): the entries of a wave's 128 rows are copied into its LDS region with coalesced loads (wave_stage_codes);
// the merge, the code look-ups and the stores work from there.
template <typename V, typename P, bool STAGED>
__global__ __launch_bounds__(256)
void sell8v_fill_kernel(long long n, long long nslices, int w, int ndeltas, int nvalues,
        const P *__restrict__ ptr, const int *__restrict__ col, const V *__restrict__ val,
        const int *__restrict__ table, const V *__restrict__ vtable, const int *__restrict__ max_col_p,
        char *__restrict__ buf, unsigned long long *counts, int *info)
{
    typedef typename bits_of<V>::type B;
    __shared__ int s_table[256];
    __shared__ B s_vtable[256];
    __shared__ unsigned s_cnt[256];
    __shared__ int s_cw[STAGED ? 4 : 1][STAGED ? WAVE_CAP : 1];
    __shared__ unsigned short s_vcw[STAGED ? 4 : 1][STAGED ? WAVE_CAP : 1];     // value CODES of the staged entries (wave_stage_codes)
    s_table[threadIdx.x] = threadIdx.x < ndeltas ? table[threadIdx.x] : INT_MAX;
    { V v = threadIdx.x < nvalues ? vtable[threadIdx.x] : V(0); B b; __builtin_memcpy(&b, &v, sizeof(B)); s_vtable[threadIdx.x] = threadIdx.x < nvalues ? b : ~B(0); }
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int max_col = *max_col_p;
    const int wp = (w + 1) / 2;
    const int lt = threadIdx.x;
    for (long long pr = (long long)blockIdx.x * blockDim.x + threadIdx.x; pr < nslices * (S8_ROWS / 2);
         pr += (long long)gridDim.x * blockDim.x) {
        const long long s = pr / (S8_ROWS / 2);
        const int t = (int)(pr % (S8_ROWS / 2));
        unsigned *cw = reinterpret_cast<unsigned *>(buf + s * ((long long)wp * 2048)) + t;
        unsigned *vw = cw + wp * 256;
        long long b[2] = {0, 0}, e[2] = {0, 0};
        const long long i = s * S8_ROWS + 2 * t;
        for (int q = 0; q < 2; ++q) if (i + q < n) { b[q] = ptr[i + q]; e[q] = ptr[i + q + 1]; }
        const int n0 = (int)min(e[0] - b[0], (long long)w), n1 = (int)min(e[1] - b[1], (long long)w);
        pair_walk pw;
        pair_merge<diag_lds> pm;
        bool staged = false, same = false;
        int *s_c = s_cw[STAGED ? lt >> 6 : 0]; unsigned short *s_vc = s_vcw[STAGED ? lt >> 6 : 0];
        if constexpr (STAGED) {
            long long e0;
            // (the previous trip's reads of this wave's region are done: same wave, program order)
            staged = wave_stage_codes<V>(col, val, b[0], e[1] > 0 ? e[1] : e[0], s_c, s_vc, s_vtable, nvalues, e0);
            if (staged) {
                diag_lds g; g.s = s_c; g.off[0] = (int)(b[0] - e0); g.off[1] = (int)(b[1] - e0); g.row = i;
                pm.d = g;
                // the common pair: both rows hold the same diagonals in the same order (column of row 2t + 1 = column of
                // row 2t, plus one) -- entry j of both goes to column j, no merge (a chain of dependent look-ups)
                same = n0 == n1 && n0 <= 8;
                if (same) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (k < n0 && s_c[g.off[0] + k] + 1 != s_c[g.off[1] + k]) same = false;
                }
                if (!same) pm.init(g, n0, n1, w);
            }
        }
        if (!staged) pw.init(col, i, b[0], n0, b[1], n1, w);
        for (int jp = 0; jp < wp; ++jp) {
            unsigned word = 0, vword = 0;
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * jp + jj;
                int ec[2] = {-1, -1}; V ev[2] = {V(0), V(0)}; bool has[2] = {false, false};      // the column's two entries: column index, value
                unsigned sc[2] = {0x10000u, 0x10000u};                                          // ... or (staged) the value's code: 0x10000 = look ev up
                if (j < w) {
                    if (staged) {
                        int k[2];
                        if (same) k[0] = k[1] = j < n0 ? j : -1; else pm.next(k[0], k[1]);
                        for (int q = 0; q < 2; ++q) if (k[q] >= 0) { has[q] = true; ec[q] = s_c[pm.d.off[q] + k[q]]; sc[q] = s_vc[pm.d.off[q] + k[q]]; }
                    } else {
                        long long en[2]; pw.next(en[0], en[1]);
                        for (int q = 0; q < 2; ++q) if (en[q] >= 0) { has[q] = true; ec[q] = col[en[q]]; ev[q] = val[en[q]]; }
                    }
                }
                for (int q = 0; q < 2; ++q) {
                    unsigned code, vcode = 255;                     // value code of padding: table entry 255 = 0.0
                    if (has[q]) {
                        if (q == 1 && has[0] && ec[1] == ec[0] + 1) code = (word >> (8 * (jj * 2))) & 255u;     // the partner's diagonal: its code
                        else code = delta_code(s_table, ndeltas, (long long)ec[q] - (i + q));
                        if (code != S8_PAD) count_code(s_cnt, code); else atomicExch(&info[1], 1);
                        if (sc[q] != 0x10000u) {                            // looked up when the entry was staged
                            if (sc[q] != 0xffffu) vcode = sc[q]; else atomicExch(&info[1], 1);
                        } else {
                            B bits; V v = ev[q];
                            __builtin_memcpy(&bits, &v, sizeof(B));
                            int lo = 0, hi = nvalues;
                            while (lo < hi) { int mid = (lo + hi) >> 1; if (s_vtable[mid] < bits) lo = mid + 1; else hi = mid; }
                            if (lo < nvalues && s_vtable[lo] == bits) vcode = (unsigned)lo;
                            else atomicExch(&info[1], 1);
                        }
                    } else code = pad_code(has[1 - q] ? ec[1 - q] : -1, 1 - q, max_col);
                    word |= code << (8 * (jj * 2 + q));
                    vword |= vcode << (8 * (jj * 2 + q));
                }
            }
            cw[jp * 256] = word;
            vw[jp * 256] = vword;
        }
    }
    __syncthreads();
    if (s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

// how many entries of the CSR matrix lie on each diagonal of the table (for the traversal heuristic)
template <typename P>
__global__ __launch_bounds__(256)
void csr_delta_count_kernel(long long n, int w, int ndeltas, const P *__restrict__ ptr, const int *__restrict__ col,
        const int *__restrict__ table, unsigned long long *counts)
{
    __shared__ int s_table[256];
    __shared__ unsigned s_cnt[256];
    s_table[threadIdx.x] = threadIdx.x < ndeltas ? table[threadIdx.x] : INT_MAX;
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const P b = ptr[i], e = ptr[i + 1];
        for (int j = 0; j < w && b + j < e; ++j) {
            const int d = (int)((long long)col[b + j] - i);
            int lo = 0, hi = ndeltas;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (s_table[mid] < d) lo = mid + 1; else hi = mid; }
            if (lo < ndeltas && s_table[lo] == d) atomicAdd(&s_cnt[lo], 1u);
        }
    }
    __syncthreads();
    if (s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

/// LDS of the march product: ring (span + 2 slices) + mirror (span) + one slice per far diagonal.
inline long long march_span_bytes(int lo, int hi, int value_bytes) {
    const long long slb = (long long)S8_ROWS * value_bytes;
    const long long lo_e = (long long)lo & ~1ll, hi_e = ((long long)hi + 1) & ~1ll;
    return ((hi_e - lo_e) * value_bytes + slb - 1) / slb * slb;
}
inline long long march_lds_bytes(int lo, int hi, int value_bytes) {
    const long long slb = (long long)S8_ROWS * value_bytes;
    return 2 * march_span_bytes(lo, hi, value_bytes) + 4 * slb + 1024 + 256 * value_bytes;      // ring + mirror, two far slots, diagonal and value tables
}

template <typename V>
int march_launch(int dev, hipStream_t s, int64_t n, long long ns, V alpha, int append, int w, const int *deltas, const V *values,
        const int *cp, const int *cc, const V *cv, const V *x, V *y, const vexhip_traversal *tr, const char *pool, const int *blocks,
        const vexhip_march *m)
{
    VEXHIP_REQUIRE(m->lo <= 0 && m->hi >= 0 && m->run >= 1 && m->nfar >= 0 && m->nfar <= 2, "bad march plan");
    const bool strips = tr && tr->grid_blocks > 0 && tr->chunk > 0;
    VEXHIP_REQUIRE(!strips || tr->chunk % m->run == 0, "march run does not divide the strip length");
    const long long grid = strips ? tr->grid_blocks / m->run : (ns + m->run - 1) / m->run;
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    trav_dev t8 = {nullptr, 0, 0, 0};
    if (strips) t8 = trav_dev{nullptr, (int)tr->chunk, (int)tr->planes, (int)tr->plane_blocks};
    t8 = with_addend(t8);                                                  // y = alpha A x + beta z: the kernel is told by append == 2
    if (t8.z) append = 2;
    const int lo_e = m->lo & ~1;                                           // window bounds on even elements (16-byte ring accesses)
    const long long span_b = march_span_bytes(m->lo, m->hi, (int)sizeof(V));
    long long lds = march_lds_bytes(m->lo, m->hi, (int)sizeof(V));
    VEXHIP_REQUIRE(lds <= 64 * 1024, "bad march plan: window too large");
    static const long long lds_floor = [] { const char *e = env(ENV_VEXHIP_MARCH_LDS); return e ? std::atoll(e) : 0ll; }();   // experiments: fewer workgroups per CU
    if (lds_floor > lds && lds_floor <= 64 * 1024) lds = lds_floor;
    const march_dev mp = {m->lo, m->hi, lo_e, (int)span_b, (int)(span_b + 2 * S8_ROWS * (long long)sizeof(V)), m->run, m->nfar, m->far[0], m->far[1], (long long)m->x_last};
    const march_cold<V> cold = {(long long)n, ns, deltas, values, cp, cc, cv, pool, t8};
#define MARCH(W) case W: sell8_march_kernel<V, W><<<(unsigned)grid, 256, (size_t)lds, s>>>(cold, alpha, append, x, y, blocks, mp); break;
    switch (w) {
        MARCH(1) MARCH(2) MARCH(3) MARCH(4) MARCH(5) MARCH(6) MARCH(7) MARCH(8)
        default: return fail(__FILE__, __LINE__, "march kernels cover ELL widths 1..8");
    }
#undef MARCH
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

// One device's product step in one launch on a strip stored with diagonal codes (SELL8: values in the slices; SELL8V: value codes),
// with or without a slice dictionary; x and y are the device's own segments (halo.hpp, the pull form; spmat.hip spmat_apply_halo).
template <typename V>
int sell8_apply_halo_impl(int dev, hipStream_t s, long long own_rows, V alpha, int append, int w, bool vcoded, const void *buf, const void *pool_,
        const int *blocks, const int *deltas, const V *values, const V *x, V *y, halo_dev H)
{
    VEXHIP_REQUIRE(H.pull && own_rows > 0 && own_rows % S8_ROWS == 0 && H.halo % S8_ROWS == 0 && w >= 1 && w <= 8, "the one-launch step of a matrix with diagonal codes: whole slices, ELL width <= 8");
    VEXHIP_REQUIRE(deltas && x && y && (buf || pool_) && (!vcoded || values), "NULL argument");
    VEXHIP_SET_DEVICE(dev);
    const long long grid = own_rows / S8_ROWS;
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    const char *b = static_cast<const char *>(buf), *pool = static_cast<const char *>(pool_);
#define HP(W, VC, DICT) sell8_pair_halo_kernel<V, W, VC, DICT><<<(unsigned)grid, 256, 0, s>>>(own_rows, alpha, append, b, deltas, values, x, y, pool, blocks, H)
#define HCASE(W) case W: if (vcoded) { if (blocks) HP(W, true, true); else HP(W, true, false); } else { if (blocks) HP(W, false, true); else HP(W, false, false); } break;
    switch (w) { HCASE(1) HCASE(2) HCASE(3) HCASE(4) HCASE(5) HCASE(6) HCASE(7) HCASE(8) }
#undef HCASE
#undef HP
    VEXHIP_LAUNCH_CHECK();
    return 0;
}
inline int grid_for(int dev, int64_t n) {
    return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, (int64_t)info(dev).cus * 16));
}

// Strip traversal from the diagonals every second row (or more) uses: see vexhip.h vexhip_traversal.
void strip_traversal(int64_t n, const std::vector<int> &table, const std::vector<unsigned long long> &counts,
        vexhip_traversal *out, int64_t rpb = S8_ROWS)
{
    std::memset(out, 0, sizeof(*out));
    const int64_t tile_rows_max = 65536;
    if (n < 8 * tile_rows_max) return;
    int64_t s_big = 0;
    for (size_t k = 0; k < table.size(); ++k)
        if (counts[k] * 2 >= (unsigned long long)n) s_big = std::max<int64_t>(s_big, table[k] < 0 ? -(int64_t)table[k] : table[k]);
    if (s_big < 2 * tile_rows_max || (s_big % rpb) != 0) return;
    const int64_t plane_blocks = s_big / rpb, nb = (n + rpb - 1) / rpb;
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(64 * S8_ROWS / rpb, plane_blocks / 8));   // 32 Ki rows per XCD strip
    const int64_t planes = (nb + plane_blocks - 1) / plane_blocks;
    const int64_t tiles = (plane_blocks + 8 * chunk - 1) / (8 * chunk);
    out->grid_blocks = tiles * planes * 8 * chunk;
    out->chunk = chunk; out->planes = planes; out->plane_blocks = plane_blocks; out->order = nullptr;
    if (out->grid_blocks >= (1ll << 31)) std::memset(out, 0, sizeof(*out));
}

template <typename V, typename P>
int sell8_fill(int dev, void *stream, int64_t n, const P *ptr, const int *col, const V *val, int64_t w,
        const int *deltas, int ndeltas, void *buf, vexhip_traversal *trav)
{
    VEXHIP_REQUIRE(n >= 0 && w >= 1 && ndeltas >= 1 && ndeltas <= 254, "bad SELL8 geometry");
    if (trav) std::memset(trav, 0, sizeof(*trav));
    if (n == 0) return 0;
    VEXHIP_REQUIRE(ptr && col && val && deltas && buf, "NULL argument");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    const long long ns = (n + S8_ROWS - 1) / S8_ROWS;
    unsigned long long *dcounts = nullptr;
    VEXHIP_TRY(hipMalloc(&dcounts, sizeof(unsigned long long) * 256 + 4 * sizeof(int)));
    int *dinfo = reinterpret_cast<int *>(dcounts + 256);                    // [0] unused, [1] error flag, [2] largest ELL column
    VEXHIP_TRY(hipMemsetAsync(dcounts, 0, sizeof(unsigned long long) * 256 + 2 * sizeof(int), s));
    if (g_max_col_hint >= 0 && g_hint_ptr == static_cast<const void *>(ptr) && g_hint_n == n) {   // the fused analysis of THIS matrix ran just before on this thread
        const int known = (int)g_max_col_hint;
        g_max_col_hint = -1;
        VEXHIP_TRY(hipMemcpyAsync(dinfo + 2, &known, sizeof(int), hipMemcpyHostToDevice, s));
        VEXHIP_TRY(hipStreamSynchronize(s));
    } else {
        VEXHIP_TRY(hipMemsetAsync(dinfo + 2, 0xff, sizeof(int), s));
        ell_max_col_kernel<P><<<grid_for(dev, n), 256, 0, s>>>(n, (int)std::min<int64_t>(w, INT_MAX), ptr, col, dinfo + 2);
    }
    if (w <= 8)
        sell8_fill_kernel<V, P, true><<<grid_for(dev, ns * (S8_ROWS / 2)), 256, 0, s>>>(n, ns, (int)w, ndeltas, ptr, col, val, deltas,
                dinfo + 2, static_cast<char *>(buf), dcounts, dinfo);
    else
        sell8_fill_kernel<V, P, false><<<grid_for(dev, ns * (S8_ROWS / 2)), 256, 0, s>>>(n, ns, (int)w, ndeltas, ptr, col, val, deltas,
                dinfo + 2, static_cast<char *>(buf), dcounts, dinfo);
    std::vector<unsigned long long> counts(256);
    std::vector<int> table(ndeltas);
    int hinfo[3] = {0, 0, -1};
    VEXHIP_TRY(hipMemcpyAsync(counts.data(), dcounts, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipMemcpyAsync(hinfo, dinfo, sizeof(hinfo), hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipMemcpyAsync(table.data(), deltas, sizeof(int) * ndeltas, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    VEXHIP_TRY(hipFree(dcounts));
    g_fill_max_col = hinfo[2];
    VEXHIP_REQUIRE(hinfo[1] == 0, "SELL8 fill: the matrix uses a diagonal that is not in the table");
    if (trav) strip_traversal(n, table, counts, trav);
    return 0;
}

template <typename V>
int spmv_sell8(int dev, void *stream, int64_t n, V alpha, int append, int64_t w, const void *buf, const int *deltas,
        const int *cp, const int *cc, const V *cv, const V *x, V *y, const vexhip_traversal *tr,
        const void *pool_ = nullptr, const int *blocks = nullptr)
{
    VEXHIP_REQUIRE(n >= 0 && w >= 1 && w < (1 << 20), "bad SELL8 geometry");
    if (n == 0) return 0;
    VEXHIP_REQUIRE(buf && deltas && x && y && (reinterpret_cast<uintptr_t>(buf) & 15) == 0, "SELL8 buffer must be 16-byte aligned");
    VEXHIP_REQUIRE((pool_ == nullptr) == (blocks == nullptr), "code pool and slice numbers go together");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    const long long ns = (n + S8_ROWS - 1) / S8_ROWS;
    const bool ordered = tr && tr->grid_blocks > 0;
    const long long grid = ordered ? tr->grid_blocks : ns;
    trav_dev t8 = {nullptr, 0, 0, 0};
    if (ordered) t8 = trav_dev{tr->order, (int)tr->chunk, (int)tr->planes, (int)tr->plane_blocks};
    t8 = with_addend(t8);                                          // y = alpha A x + beta z (vexhip_spmat_apply_axpby_*): the kernels' store_pair adds it
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    const char *b = static_cast<const char *>(buf), *pool = static_cast<const char *>(pool_);
#define PAIR(W, DICT) sell8_pair_kernel<V, W, false, DICT><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, b, deltas, (const V *)nullptr, cp, cc, cv, x, y, t8, pool, blocks)
#define CASE(W) case W: if (g_sell8_variant != 1) { if (blocks) PAIR(W, true); else PAIR(W, false); } \
        else sell8_kernel<V, W><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, (int)w, b, deltas, cp, cc, cv, x, y, t8, pool, blocks); break;
    switch (w) {
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
        default: sell8_kernel<V, 0><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, (int)w, b, deltas, cp, cc, cv, x, y, t8, pool, blocks);
    }
#undef CASE
#undef PAIR
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

template <typename V, typename P>
int sell8v_analyze(int dev, void *stream, int64_t n, const P *ptr, const V *val, int64_t w, V *values, int *nvalues)
{
    typedef typename bits_of<V>::type B;
    VEXHIP_REQUIRE(nvalues && values, "NULL output");
    *nvalues = -1;
    if (n <= 0 || w < 1) return 0;
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    B *d = nullptr;
    VEXHIP_TRY(hipMalloc(&d, sizeof(B) * HASH_SLOTS + 2 * sizeof(int)));
    int *dinfo = reinterpret_cast<int *>(d + HASH_SLOTS);
    VEXHIP_TRY(hipMemsetAsync(d, 0xff, sizeof(B) * HASH_SLOTS, s));
    VEXHIP_TRY(hipMemsetAsync(dinfo, 0, 2 * sizeof(int), s));
    value_collect_kernel<V, P><<<grid_for(dev, n), 256, 0, s>>>(n, (int)std::min<int64_t>(w, INT_MAX), ptr, val, d, dinfo);
    std::vector<B> host(HASH_SLOTS);
    int hinfo[2] = {0, 0};
    VEXHIP_TRY(hipMemcpyAsync(host.data(), d, sizeof(B) * HASH_SLOTS, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipMemcpyAsync(hinfo, dinfo, sizeof(hinfo), hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    VEXHIP_TRY(hipFree(d));
    if (hinfo[1] != 0 || hinfo[0] > 255 || hinfo[0] < 1) return 0;          // too many distinct values: keep them as they are
    std::vector<B> table;
    for (B b : host) if (b != ~B(0)) table.push_back(b);
    std::sort(table.begin(), table.end());                                    // sorted by bit pattern (what the fill kernel searches)
    if ((int)table.size() != hinfo[0]) return fail(__FILE__, __LINE__, "value set is inconsistent");
    std::vector<V> vals(256, V(0));
    for (size_t k = 0; k < table.size(); ++k) __builtin_memcpy(&vals[k], &table[k], sizeof(B));
    VEXHIP_TRY(hipMemcpyAsync(values, vals.data(), sizeof(V) * 256, hipMemcpyHostToDevice, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    *nvalues = hinfo[0];
    return 0;
}

// deltas[256] / values[256]: device tables as the two separate analyses write them; *ndeltas / *nvalues = -1 when not applicable
template <typename V, typename P>
int analyze_fused(int dev, void *stream, int64_t n, const P *ptr, const int *col, const V *val, int64_t w,
        int32_t *deltas, int *ndeltas, V *values, int *nvalues)
{
    typedef typename bits_of<V>::type B;
    *ndeltas = -1; *nvalues = -1; g_max_col_hint = -1;
    if (n <= 0 || w < 1) return 0;
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    // [int set: HASH_SLOTS + 2 info][value set: HASH_SLOTS B + 2 int info][max col]
    const size_t dbytes = sizeof(int) * (HASH_SLOTS + 2), vbytes = sizeof(B) * HASH_SLOTS + 2 * sizeof(int);
    char *d = nullptr;
    setup_trace trace(s);
    VEXHIP_TRY(hipMalloc(&d, dbytes + vbytes + sizeof(int) + 64));
    trace.mark("  fused: hipMalloc");
    int *dset = reinterpret_cast<int *>(d);
    B *vset = reinterpret_cast<B *>(d + ((dbytes + 15) / 16) * 16);
    int *vinfo = reinterpret_cast<int *>(vset + HASH_SLOTS);
    int *dmax = vinfo + 2;
    std::vector<int> hd(HASH_SLOTS + 2, EMPTY);
    hd[HASH_SLOTS] = 0; hd[HASH_SLOTS + 1] = 0;
    hipError_t e = hipMemcpyAsync(dset, hd.data(), dbytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(vset, 0xff, sizeof(B) * HASH_SLOTS, s);
    if (e == hipSuccess) e = hipMemsetAsync(vinfo, 0, 2 * sizeof(int), s);
    if (e == hipSuccess) e = hipMemsetAsync(dmax, 0xff, sizeof(int), s);
    std::vector<B> hv(HASH_SLOTS);
    int vi[2] = {0, 0}, hmax = -1;
    trace.mark("  fused: H2D + memsets");
    if (e == hipSuccess) {
        analyze_fused_kernel<V, P><<<grid_for(dev, n), 256, 0, s>>>(n, (int)std::min<int64_t>(w, INT_MAX), ptr, col, val, dset, dset + HASH_SLOTS, vset, vinfo, dmax);
        e = hipGetLastError();
    }
    trace.mark("  fused: kernel");
    if (e == hipSuccess) e = hipMemcpyAsync(hd.data(), dset, dbytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(hv.data(), vset, sizeof(B) * HASH_SLOTS, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(vi, vinfo, sizeof(vi), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&hmax, dmax, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    trace.mark("  fused: D2H");
    (void)hipFree(d);
    trace.mark("  fused: hipFree");
    VEXHIP_TRY(e);
    g_max_col_hint = hmax; g_hint_ptr = ptr; g_hint_n = n;
    if (!(hd[HASH_SLOTS + 1] != 0 || hd[HASH_SLOTS] > 254 || hd[HASH_SLOTS] < 1)) {
        std::vector<int> table;
        for (int k = 0; k < HASH_SLOTS; ++k) if (hd[k] != EMPTY) table.push_back(hd[k]);
        std::sort(table.begin(), table.end());
        if ((int)table.size() != hd[HASH_SLOTS]) return fail(__FILE__, __LINE__, "diagonal set is inconsistent");
        table.resize(256, INT_MAX);
        VEXHIP_TRY(hipMemcpyAsync(deltas, table.data(), sizeof(int) * 256, hipMemcpyHostToDevice, s));
        VEXHIP_TRY(hipStreamSynchronize(s));
        *ndeltas = hd[HASH_SLOTS];
    }
    if (*ndeltas > 0 && !(vi[1] != 0 || vi[0] > 255 || vi[0] < 1)) {
        std::vector<B> table;
        for (B b : hv) if (b != ~B(0)) table.push_back(b);
        std::sort(table.begin(), table.end());
        if ((int)table.size() != vi[0]) return fail(__FILE__, __LINE__, "value set is inconsistent");
        std::vector<V> vals(256, V(0));
        for (size_t k = 0; k < table.size(); ++k) __builtin_memcpy(&vals[k], &table[k], sizeof(B));
        VEXHIP_TRY(hipMemcpyAsync(values, vals.data(), sizeof(V) * 256, hipMemcpyHostToDevice, s));
        VEXHIP_TRY(hipStreamSynchronize(s));
        *nvalues = vi[0];
    }
    return 0;
}

template <typename V, typename P>
int sell8v_fill(int dev, void *stream, int64_t n, const P *ptr, const int *col, const V *val, int64_t w,
        const int *deltas, int ndeltas, const V *values, int nvalues, void *buf, vexhip_traversal *trav)
{
    VEXHIP_REQUIRE(n >= 0 && w >= 1 && ndeltas >= 1 && ndeltas <= 254 && nvalues >= 1 && nvalues <= 255, "bad SELL8V geometry");
    if (trav) std::memset(trav, 0, sizeof(*trav));
    if (n == 0) return 0;
    VEXHIP_REQUIRE(ptr && col && val && deltas && values && buf, "NULL argument");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    const long long ns = (n + S8_ROWS - 1) / S8_ROWS;
    unsigned long long *dcounts = nullptr;
    VEXHIP_TRY(hipMalloc(&dcounts, sizeof(unsigned long long) * 256 + 4 * sizeof(int)));
    int *dinfo = reinterpret_cast<int *>(dcounts + 256);                    // [0] unused, [1] error flag, [2] largest ELL column
    VEXHIP_TRY(hipMemsetAsync(dcounts, 0, sizeof(unsigned long long) * 256 + 2 * sizeof(int), s));
    if (g_max_col_hint >= 0 && g_hint_ptr == static_cast<const void *>(ptr) && g_hint_n == n) {   // the fused analysis of THIS matrix ran just before on this thread
        const int known = (int)g_max_col_hint;
        g_max_col_hint = -1;
        VEXHIP_TRY(hipMemcpyAsync(dinfo + 2, &known, sizeof(int), hipMemcpyHostToDevice, s));
        VEXHIP_TRY(hipStreamSynchronize(s));
    } else {
        VEXHIP_TRY(hipMemsetAsync(dinfo + 2, 0xff, sizeof(int), s));
        ell_max_col_kernel<P><<<grid_for(dev, n), 256, 0, s>>>(n, (int)std::min<int64_t>(w, INT_MAX), ptr, col, dinfo + 2);
    }
    if (w <= 8)
        sell8v_fill_kernel<V, P, true><<<grid_for(dev, ns * (S8_ROWS / 2)), 256, 0, s>>>(n, ns, (int)w, ndeltas, nvalues, ptr, col, val, deltas, values,
                dinfo + 2, static_cast<char *>(buf), dcounts, dinfo);
    else
        sell8v_fill_kernel<V, P, false><<<grid_for(dev, ns * (S8_ROWS / 2)), 256, 0, s>>>(n, ns, (int)w, ndeltas, nvalues, ptr, col, val, deltas, values,
                dinfo + 2, static_cast<char *>(buf), dcounts, dinfo);
    std::vector<unsigned long long> counts(256);
    std::vector<int> table(ndeltas);
    int hinfo[3] = {0, 0, -1};
    VEXHIP_TRY(hipMemcpyAsync(counts.data(), dcounts, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipMemcpyAsync(hinfo, dinfo, sizeof(hinfo), hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipMemcpyAsync(table.data(), deltas, sizeof(int) * ndeltas, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    VEXHIP_TRY(hipFree(dcounts));
    g_fill_max_col = hinfo[2];
    VEXHIP_REQUIRE(hinfo[1] == 0, "SELL8V fill: a diagonal or a value of the matrix is not in its table");
    if (trav) strip_traversal(n, table, counts, trav);
    return 0;
}

template <typename V>
int spmv_sell8v(int dev, void *stream, int64_t n, V alpha, int append, int64_t w, const void *buf, const int *deltas, const V *values,
        const int *cp, const int *cc, const V *cv, const V *x, V *y, const vexhip_traversal *tr, const int *blocks = nullptr,
        const vexhip_march *march = nullptr)
{
    VEXHIP_REQUIRE(n >= 0 && w >= 1 && w < (1 << 20), "bad SELL8V geometry");
    if (n == 0) return 0;
    VEXHIP_REQUIRE(buf && deltas && values && x && y && (reinterpret_cast<uintptr_t>(buf) & 15) == 0, "SELL8V buffer must be 16-byte aligned");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    const long long ns = (n + S8_ROWS - 1) / S8_ROWS;
    long long grid = 0;
    trav_dev t8 = make_traversal(tr, ns, &grid);
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    const char *b = static_cast<const char *>(buf);
    if (march && blocks && w <= 8 && g_sell8_variant == 0 && !(tr && tr->order))
        return march_launch<V>(dev, s, n, ns, alpha, append, (int)w, deltas, values, cp, cc, cv, x, y, tr, b, blocks, march);
    t8 = with_addend(t8);
#define PAIRV(W, DICT) sell8_pair_kernel<V, (W <= 8 ? W : 8), true, DICT><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, b, deltas, values, cp, cc, cv, x, y, t8, b, blocks)
#define CASE(W) case W: if (g_sell8_variant != 1 && W <= 8) { if (blocks) PAIRV(W, true); else PAIRV(W, false); } \
        else sell8v_kernel<V, W><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, (int)w, b, deltas, values, cp, cc, cv, x, y, t8, blocks); break;
    switch (w) {
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
        default: sell8v_kernel<V, 0><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, (int)w, b, deltas, values, cp, cc, cv, x, y, t8, blocks);
    }
#undef CASE
#undef PAIRV
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Slice dictionary.  A matrix assembled from a constant-coefficient stencil on a structured grid repeats itself slice
// after slice: with value codes a 512-row slice is nothing but codes, and the 262 144 slices of the 512^3 Poisson
// matrix hold TWO distinct blocks (a grid line inside the domain, a grid line on its boundary).  dictionary():
//   hash every slice (64 bits) -> the host numbers the distinct hashes in order of first appearance -> a second kernel
//   compares every slice with the representative of its number word by word (a collision, or more than max_blocks
//   distinct slices, gives *nblocks = -1 and the storage stays as it is) -> the representatives are copied into `pool`.
// The product then reads slice s at pool + blocks[s] * slice_bytes: the code stream (half of the value-coded product's
// HBM traffic) is replaced by one 4-byte index per slice, and the pool stays in L1 / L2.  Same codes, same arithmetic:
// bit-identical.  Measured before it was built (profiles/r02_sell8v_ablation.json: every slice reading one of 8 fixed
// blocks): 0.764 -> 0.641 ms with cached loads, 0.808 ms with the non-temporal loads of the streamed layout.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void slice_hash_kernel(long long nslices, long long stride_words, long long slice_words, const unsigned *__restrict__ buf, unsigned long long *__restrict__ hash)
{
    __shared__ unsigned long long s_part[4];
    for (long long s = blockIdx.x; s < nslices; s += gridDim.x) {
        const unsigned *w = buf + s * stride_words;
        unsigned long long h = 0;
        for (long long k = threadIdx.x; k < slice_words; k += 256) {
            unsigned long long z = ((unsigned long long)w[k] + 0x9E3779B97F4A7C15ull) * (2 * (unsigned long long)k + 0xD1B54A32D192ED03ull);
            z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
            h += z;                                                   // position enters through the multiplier: order-sensitive
        }
        for (int o = 32; o > 0; o >>= 1) h += __shfl_down(h, o, 64);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = h;
        __syncthreads();
        if (threadIdx.x == 0) hash[s] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256)
void slice_verify_kernel(long long nslices, long long stride_words, long long slice_words, const unsigned *__restrict__ buf,
        const int *__restrict__ blocks, const int *__restrict__ reps, int *__restrict__ mismatch)
{
    for (long long s = blockIdx.x; s < nslices; s += gridDim.x) {
        const long long r = reps[blocks[s]];
        if (r == s) continue;
        const unsigned *a = buf + s * stride_words, *b = buf + r * stride_words;
        bool bad = false;
        for (long long k = threadIdx.x; k < slice_words; k += 256) bad |= a[k] != b[k];
        if (bad) atomicExch(mismatch, 1);
    }
}

__global__ __launch_bounds__(256)
void slice_pool_kernel(long long stride_words, long long slice_words, const unsigned *__restrict__ buf, const int *__restrict__ reps, unsigned *__restrict__ pool)
{
    const unsigned *a = buf + (long long)reps[blockIdx.x] * stride_words;
    unsigned *o = pool + (long long)blockIdx.x * slice_words;
    for (long long k = threadIdx.x; k < slice_words; k += 256) o[k] = a[k];
}

int slice_dictionary(int dev, void *stream, int64_t nslices, int64_t stride_bytes, int64_t slice_bytes, const void *buf, int64_t max_blocks,
        int32_t *blocks, void *pool, int64_t *nblocks)
{
    VEXHIP_REQUIRE(nblocks, "NULL output");
    *nblocks = -1;
    VEXHIP_REQUIRE(nslices >= 0 && slice_bytes > 0 && slice_bytes % 4 == 0 && stride_bytes >= slice_bytes && stride_bytes % 4 == 0 &&
            max_blocks >= 1 && max_blocks < (1ll << 31), "bad dictionary geometry");
    if (nslices == 0) { *nblocks = 0; return 0; }
    VEXHIP_REQUIRE(buf && blocks && pool, "NULL argument");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    const long long words = slice_bytes / 4, stride = stride_bytes / 4;
    unsigned long long *dh = nullptr;
    VEXHIP_TRY(hipMalloc(&dh, sizeof(unsigned long long) * (size_t)nslices));
    const int grid = (int)std::min<int64_t>(nslices, (int64_t)info(dev).cus * 32);
    slice_hash_kernel<<<grid, 256, 0, s>>>(nslices, stride, words, static_cast<const unsigned *>(buf), dh);
    std::vector<unsigned long long> h((size_t)nslices);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), dh, sizeof(unsigned long long) * (size_t)nslices, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dh);
    VEXHIP_TRY(e);
    // number the distinct hashes in order of first appearance
    std::vector<int32_t> id((size_t)nslices), reps;
    {
        std::vector<std::pair<unsigned long long, int32_t>> table;          // open addressing, power of two >= 4 * max_blocks
        size_t cap = 64; while (cap < 4 * (size_t)max_blocks) cap <<= 1;
        table.assign(cap, std::make_pair(0ull, (int32_t)-1));
        for (int64_t k = 0; k < nslices; ++k) {
            size_t pos = (size_t)(h[(size_t)k] * 0x9E3779B97F4A7C15ull >> 17) & (cap - 1);
            for (;;) {
                if (table[pos].second < 0) {
                    if ((int64_t)reps.size() == max_blocks) return 0;      // too many distinct slices: *nblocks stays -1
                    table[pos] = std::make_pair(h[(size_t)k], (int32_t)reps.size());
                    reps.push_back((int32_t)k);
                    break;
                }
                if (table[pos].first == h[(size_t)k]) break;
                pos = (pos + 1) & (cap - 1);
            }
            id[(size_t)k] = table[pos].second;
        }
    }
    int32_t *dreps = nullptr;
    VEXHIP_TRY(hipMalloc(&dreps, sizeof(int32_t) * reps.size() + sizeof(int)));
    int *dflag = reinterpret_cast<int *>(dreps + reps.size());
    int flag = 1;
    e = hipMemcpyAsync(blocks, id.data(), sizeof(int32_t) * (size_t)nslices, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dreps, reps.data(), sizeof(int32_t) * reps.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(dflag, 0, sizeof(int), s);
    if (e == hipSuccess) {
        slice_verify_kernel<<<grid, 256, 0, s>>>(nslices, stride, words, static_cast<const unsigned *>(buf), blocks, dreps, dflag);
        slice_pool_kernel<<<(unsigned)reps.size(), 256, 0, s>>>(stride, words, static_cast<const unsigned *>(buf), dreps, static_cast<unsigned *>(pool));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dreps);
    VEXHIP_TRY(e);
    if (flag == 0) *nblocks = (int64_t)reps.size();                        // flag != 0: two different slices share a hash
    return 0;
}

template <typename P>
int sell8_analyze(int dev, void *stream, int64_t n, const P *ptr, const int32_t *col,
        int64_t w, int32_t *deltas, int *ndeltas)
{
    VEXHIP_REQUIRE(ndeltas && deltas, "NULL output");
    *ndeltas = -1;
    if (n <= 0 || w < 1) return 0;
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    int *d = nullptr;
    VEXHIP_TRY(hipMalloc(&d, sizeof(int) * (HASH_SLOTS + 2)));
    std::vector<int> host(HASH_SLOTS + 2, EMPTY);
    host[HASH_SLOTS] = 0; host[HASH_SLOTS + 1] = 0;
    VEXHIP_TRY(hipMemcpyAsync(d, host.data(), sizeof(int) * host.size(), hipMemcpyHostToDevice, s));
    delta_collect_kernel<P><<<grid_for(dev, n), 256, 0, s>>>(n, (int)std::min<int64_t>(w, INT_MAX), ptr, col, d, d + HASH_SLOTS);
    VEXHIP_TRY(hipMemcpyAsync(host.data(), d, sizeof(int) * host.size(), hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    VEXHIP_TRY(hipFree(d));
    if (env(ENV_VEXHIP_DEBUG)) std::fprintf(stderr, "sell8 analyze: count %d overflow %d\n", host[HASH_SLOTS], host[HASH_SLOTS + 1]);
    if (host[HASH_SLOTS + 1] != 0 || host[HASH_SLOTS] > 254 || host[HASH_SLOTS] < 1) return 0;   // not a banded matrix (codes 254 and 255 are padding)
    std::vector<int> table;
    for (int k = 0; k < HASH_SLOTS; ++k) if (host[k] != EMPTY) table.push_back(host[k]);
    std::sort(table.begin(), table.end());
    if ((int)table.size() != host[HASH_SLOTS]) return fail(__FILE__, __LINE__, "diagonal set is inconsistent");
    table.resize(256, INT_MAX);
    VEXHIP_TRY(hipMemcpyAsync(deltas, table.data(), sizeof(int) * 256, hipMemcpyHostToDevice, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    *ndeltas = host[HASH_SLOTS];
    return 0;
}


} // namespace

int sell8_apply_halo(int dev, hipStream_t s, long long own_rows, double alpha, int append, int w, bool vcoded, const void *buf, const void *pool,
        const int *blocks, const int *deltas, const double *values, const double *x, double *y, halo_dev H)
{ return sell8_apply_halo_impl<double>(dev, s, own_rows, alpha, append, w, vcoded, buf, pool, blocks, deltas, values, x, y, H); }

// ---- runs of three diagonals (above): the plan of a dictionary's blocks, and the product (spmat.hip) ----
// desc_out: device array of nblocks x 4 waves x RUNS_STRIDE ints (hipFree), NULL when no wave of any block holds a fast group
template <typename V>
static int sell8v_runs_plan_impl(int dev, void *stream, const void *pool, int64_t nblocks, int64_t w, const int *deltas, const V *values, int **desc_out)
{
    VEXHIP_REQUIRE(desc_out, "NULL output");
    *desc_out = nullptr;
    if (!pool || !deltas || !values || nblocks < 1 || w < 10 || w > 3 * RUNS_GROUPS) return 0;
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    int *desc = nullptr;
    const size_t ints = (size_t)nblocks * 4 * RUNS_STRIDE;
    VEXHIP_TRY(hipMalloc(reinterpret_cast<void **>(&desc), sizeof(int) * (ints + 2)));
    int *total = desc + ints;                                                          // {columns in fast groups, columns that hold entries} over all waves of all blocks
    hipError_t e = hipMemsetAsync(desc, 0, sizeof(int) * (ints + 2), s);
    if (e == hipSuccess) { sell8v_runs_plan_kernel<V><<<(unsigned)nblocks, 256, 0, s>>>(static_cast<const char *>(pool), (int)w, deltas, values, desc, total); e = hipGetLastError(); }
    int found[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(found, total, sizeof(found), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    // Worth it where most columns are fast: the codes' path of this kernel takes its columns one at a time (the any-width kernel eight at a
    // time).  A 19-point row -- five triples and four single columns, the triples at columns 1, 5, 8, 11, 15: not aligned -- is slower through
    // it (320^3: 0.478 ms against 0.339) and keeps the any-width kernel; a form with triples at any column and single fast columns (a state
    // per column instead of straight-line groups) lost on BOTH (0.458 and 0.470 ms, profiles/r06_runs_of_three.md).
    if (e != hipSuccess || found[0] * 4 < found[1] * 3) { (void)hipFree(desc); return e == hipSuccess ? 0 : check(e, __FILE__, __LINE__); }
    *desc_out = desc;
    return 0;
}
int sell8v_runs_plan(int dev, void *stream, const void *pool, int64_t nblocks, int64_t w, const int *deltas, const double *values, int **desc_out)
{ return sell8v_runs_plan_impl<double>(dev, stream, pool, nblocks, w, deltas, values, desc_out); }
int sell8v_runs_plan(int dev, void *stream, const void *pool, int64_t nblocks, int64_t w, const int *deltas, const float *values, int **desc_out)
{ return sell8v_runs_plan_impl<float>(dev, stream, pool, nblocks, w, deltas, values, desc_out); }

template <typename V>
static int sell8v_runs_apply_impl(int dev, void *stream, int64_t n, V alpha, int append, int64_t w, const void *pool, const int *blocks,
        const int *deltas, const V *values, const int *cp, const int *cc, const V *cv, const V *x, V *y, const vexhip_traversal *tr, const int *desc, long long x_last)
{
    VEXHIP_REQUIRE(n >= 0 && w >= 10 && w <= 3 * RUNS_GROUPS && pool && blocks && deltas && values && desc && x_last >= 0, "bad arguments of the runs product");
    if (n == 0) return 0;
    VEXHIP_REQUIRE(x && y, "NULL vector");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    const long long ns = (n + S8_ROWS - 1) / S8_ROWS;
    long long grid = 0;
    trav_dev t8 = make_traversal(tr, ns, &grid);
    if (!t8.order && t8.chunk == 0) grid = 8 * ((ns + 7) / 8);                        // (the kernel deals the slices to the XCDs itself)
    VEXHIP_REQUIRE(grid < (1ll << 31), "matrix too large for one launch");
    t8 = with_addend(t8);
    sell8v_runs_kernel<V><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, (int)w, static_cast<const char *>(pool), deltas, values, cp, cc, cv, x, y, t8, blocks, desc, x_last);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}
int sell8v_runs_apply(int dev, void *stream, int64_t n, double alpha, int append, int64_t w, const void *pool, const int *blocks, const int *deltas, const double *values,
        const int *cp, const int *cc, const double *cv, const double *x, double *y, const vexhip_traversal *tr, const int *desc, long long x_last)
{ return sell8v_runs_apply_impl<double>(dev, stream, n, alpha, append, w, pool, blocks, deltas, values, cp, cc, cv, x, y, tr, desc, x_last); }
int sell8v_runs_apply(int dev, void *stream, int64_t n, float alpha, int append, int64_t w, const void *pool, const int *blocks, const int *deltas, const float *values,
        const int *cp, const int *cc, const float *cv, const float *x, float *y, const vexhip_traversal *tr, const int *desc, long long x_last)
{ return sell8v_runs_apply_impl<float>(dev, stream, n, alpha, append, w, pool, blocks, deltas, values, cp, cc, cv, x, y, tr, desc, x_last); }


// ---- 64-bit row pointers (a device may hold 2^31 entries or more; columns stay 32-bit): internal entry points used by
//      spmat.hip -- vexhip_spmat_create_*_p64 is the C-ABI door (reference: size_t row pointers, vexcl/spmat.hpp:56-57)
int analyze_fused_p32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const double *val, int64_t w, int32_t *deltas, int *ndeltas, double *values, int *nvalues)
{ return analyze_fused<double, int32_t>(dev, stream, n, ptr, col, val, w, deltas, ndeltas, values, nvalues); }
int analyze_fused_p32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const float *val, int64_t w, int32_t *deltas, int *ndeltas, float *values, int *nvalues)
{ return analyze_fused<float, int32_t>(dev, stream, n, ptr, col, val, w, deltas, ndeltas, values, nvalues); }
int analyze_fused_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const double *val, int64_t w, int32_t *deltas, int *ndeltas, double *values, int *nvalues)
{ return analyze_fused<double, long long>(dev, stream, n, ptr, col, val, w, deltas, ndeltas, values, nvalues); }
int analyze_fused_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const float *val, int64_t w, int32_t *deltas, int *ndeltas, float *values, int *nvalues)
{ return analyze_fused<float, long long>(dev, stream, n, ptr, col, val, w, deltas, ndeltas, values, nvalues); }
void clear_max_col_hint() { g_max_col_hint = -1; g_hint_ptr = nullptr; g_hint_n = -1; }
// the largest ELL column the fused analysis of THIS matrix saw on this thread (-1: no such analysis): the direct grid build (grid.hip)
long long analysis_max_col(const void *ptr, long long n) { return (g_max_col_hint >= 0 && g_hint_ptr == ptr && g_hint_n == n) ? g_max_col_hint : -1; }
int sell8_analyze_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, int64_t w, int32_t *deltas, int *ndeltas)
{ return sell8_analyze<long long>(dev, stream, n, ptr, col, w, deltas, ndeltas); }
int sell8v_analyze_p64(int dev, void *stream, int64_t n, const long long *ptr, const double *val, int64_t w, double *values, int *nvalues)
{ return sell8v_analyze<double, long long>(dev, stream, n, ptr, val, w, values, nvalues); }
int sell8v_analyze_p64(int dev, void *stream, int64_t n, const long long *ptr, const float *val, int64_t w, float *values, int *nvalues)
{ return sell8v_analyze<float, long long>(dev, stream, n, ptr, val, w, values, nvalues); }
int sell8v_fill_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const double *val, int64_t w,
        const int32_t *deltas, int ndeltas, const double *values, int nvalues, void *buf, vexhip_traversal *trav)
{ return sell8v_fill<double, long long>(dev, stream, n, ptr, col, val, w, deltas, ndeltas, values, nvalues, buf, trav); }
int sell8v_fill_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const float *val, int64_t w,
        const int32_t *deltas, int ndeltas, const float *values, int nvalues, void *buf, vexhip_traversal *trav)
{ return sell8v_fill<float, long long>(dev, stream, n, ptr, col, val, w, deltas, ndeltas, values, nvalues, buf, trav); }
int sell8_fill_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const double *val, int64_t w,
        const int32_t *deltas, int ndeltas, void *buf, vexhip_traversal *trav)
{ return sell8_fill<double, long long>(dev, stream, n, ptr, col, val, w, deltas, ndeltas, buf, trav); }
int sell8_fill_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const float *val, int64_t w,
        const int32_t *deltas, int ndeltas, void *buf, vexhip_traversal *trav)
{ return sell8_fill<float, long long>(dev, stream, n, ptr, col, val, w, deltas, ndeltas, buf, trav); }

template <typename P>
int csr_traversal_impl(int dev, void *stream, int64_t n, const P *ptr, const int32_t *col,
        int rows_per_block, vexhip_traversal *traversal)
{
    VEXHIP_REQUIRE(traversal && rows_per_block > 0, "bad argument");
    std::memset(traversal, 0, sizeof(*traversal));
    if (n <= 0) return 0;
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    int *deltas = nullptr;
    VEXHIP_TRY(hipMalloc(&deltas, sizeof(int) * 256 + sizeof(unsigned long long) * 256));
    unsigned long long *dcounts = reinterpret_cast<unsigned long long *>(deltas + 256);
    int nd = -1;
    const int w = 64;                                    // the first 64 entries of every row decide
    int rc = sell8_analyze<P>(dev, stream, n, ptr, col, w, deltas, &nd);
    if (rc == 0 && nd > 0) {
        std::vector<unsigned long long> counts(256);
        std::vector<int> table(nd);
        hipError_t e = hipMemsetAsync(dcounts, 0, sizeof(unsigned long long) * 256, s);
        if (e == hipSuccess) {
            csr_delta_count_kernel<P><<<grid_for(dev, n), 256, 0, s>>>(n, w, nd, ptr, col, deltas, dcounts);
            e = hipMemcpyAsync(counts.data(), dcounts, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost, s);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(table.data(), deltas, sizeof(int) * nd, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { (void)hipFree(deltas); return check(e, __FILE__, __LINE__); }
        strip_traversal(n, table, counts, traversal, rows_per_block);
    }
    VEXHIP_TRY(hipFree(deltas));
    return rc;
}

// 64-bit row pointers: the strips of a CSR matrix kept in CSR with 2^31 entries or more (spmat.hip)
int csr_traversal_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, int rows_per_block, vexhip_traversal *traversal)
{ return csr_traversal_impl<long long>(dev, stream, n, ptr, col, rows_per_block, traversal); }

} // namespace vexhip

using namespace vexhip;

extern "C" {

int64_t vexhip_sell8_bytes(int64_t n, int64_t w, int value_bytes) {
    return (n + S8_ROWS - 1) / S8_ROWS * slice_bytes(w, value_bytes);
}

int vexhip_sell8_analyze_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col,
        int64_t w, int32_t *deltas, int *ndeltas)
{ return sell8_analyze<int32_t>(dev, stream, n, ptr, col, w, deltas, ndeltas); }

int vexhip_spmv_sell8_set_variant(int variant) { g_sell8_variant = variant; return 0; }

int64_t vexhip_sell8v_bytes(int64_t n, int64_t w) { return (n + S8_ROWS - 1) / S8_ROWS * ((w + 1) / 2) * 2048; }

int vexhip_sell8v_analyze_f64_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const double *val, int64_t w, double *values, int *nvalues)
{ return sell8v_analyze<double, int32_t>(dev, stream, n, ptr, val, w, values, nvalues); }
int vexhip_sell8v_analyze_f32_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const float *val, int64_t w, float *values, int *nvalues)
{ return sell8v_analyze<float, int32_t>(dev, stream, n, ptr, val, w, values, nvalues); }

int vexhip_sell8v_fill_f64_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const double *val, int64_t w,
        const int32_t *deltas, int ndeltas, const double *values, int nvalues, void *buf, vexhip_traversal *traversal)
{ return sell8v_fill<double, int32_t>(dev, stream, n, ptr, col, val, w, deltas, ndeltas, values, nvalues, buf, traversal); }
int vexhip_sell8v_fill_f32_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const float *val, int64_t w,
        const int32_t *deltas, int ndeltas, const float *values, int nvalues, void *buf, vexhip_traversal *traversal)
{ return sell8v_fill<float, int32_t>(dev, stream, n, ptr, col, val, w, deltas, ndeltas, values, nvalues, buf, traversal); }

int vexhip_spmv_sell8v_f64_i32(int dev, void *stream, int64_t n, double alpha, int append, int64_t w, const void *buf,
        const int32_t *deltas, const double *values, const int32_t *cp, const int32_t *cc, const double *cv,
        const double *x, double *y, const vexhip_traversal *traversal)
{ return spmv_sell8v<double>(dev, stream, n, alpha, append, w, buf, deltas, values, cp, cc, cv, x, y, traversal); }
int vexhip_spmv_sell8v_f32_i32(int dev, void *stream, int64_t n, float alpha, int append, int64_t w, const void *buf,
        const int32_t *deltas, const float *values, const int32_t *cp, const int32_t *cc, const float *cv,
        const float *x, float *y, const vexhip_traversal *traversal)
{ return spmv_sell8v<float>(dev, stream, n, alpha, append, w, buf, deltas, values, cp, cc, cv, x, y, traversal); }

int vexhip_spmv_sell8v_dict_f64_i32(int dev, void *stream, int64_t n, double alpha, int append, int64_t w, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const double *values, const int32_t *cp, const int32_t *cc, const double *cv,
        const double *x, double *y, const vexhip_traversal *traversal)
{ return spmv_sell8v<double>(dev, stream, n, alpha, append, w, pool, deltas, values, cp, cc, cv, x, y, traversal, blocks); }
int vexhip_spmv_sell8v_dict_f32_i32(int dev, void *stream, int64_t n, float alpha, int append, int64_t w, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const float *values, const int32_t *cp, const int32_t *cc, const float *cv,
        const float *x, float *y, const vexhip_traversal *traversal)
{ return spmv_sell8v<float>(dev, stream, n, alpha, append, w, pool, deltas, values, cp, cc, cv, x, y, traversal, blocks); }

int vexhip_slice_dictionary(int dev, void *stream, int64_t nslices, int64_t stride_bytes, int64_t slice_bytes, const void *buf, int64_t max_blocks,
        int32_t *blocks, void *pool, int64_t *nblocks)
{ return slice_dictionary(dev, stream, nslices, stride_bytes, slice_bytes, buf, max_blocks, blocks, pool, nblocks); }

int vexhip_spmv_sell8_dict_f64_i32(int dev, void *stream, int64_t n, double alpha, int append, int64_t w, const void *buf, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const int32_t *cp, const int32_t *cc, const double *cv,
        const double *x, double *y, const vexhip_traversal *traversal)
{ return spmv_sell8<double>(dev, stream, n, alpha, append, w, buf, deltas, cp, cc, cv, x, y, traversal, pool, blocks); }
int vexhip_spmv_sell8_dict_f32_i32(int dev, void *stream, int64_t n, float alpha, int append, int64_t w, const void *buf, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const int32_t *cp, const int32_t *cc, const float *cv,
        const float *x, float *y, const vexhip_traversal *traversal)
{ return spmv_sell8<float>(dev, stream, n, alpha, append, w, buf, deltas, cp, cc, cv, x, y, traversal, pool, blocks); }

// ---- march plan -------------------------------------------------------------------------------------------------------
int vexhip_sell8_march_plan(int dev, void *stream, const int32_t *deltas, int ndeltas, const int32_t *blocks, int64_t nslices,
        int value_bytes, const vexhip_traversal *traversal, int64_t x_last, vexhip_march *out)
{
    reload_env();
    VEXHIP_REQUIRE(out, "NULL output");
    std::memset(out, 0, sizeof(*out));
    if (ndeltas < 1 || ndeltas > 254 || !deltas || !blocks || nslices < 8 || x_last < 0) return 0;
    if (traversal && traversal->order) return 0;
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    std::vector<int> table((size_t)ndeltas), id((size_t)nslices);
    VEXHIP_TRY(hipMemcpyAsync(table.data(), deltas, sizeof(int) * (size_t)ndeltas, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipMemcpyAsync(id.data(), blocks, sizeof(int) * (size_t)nslices, hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    // the code block must rarely change from one slice to the next (a change costs a dependent load + decode)
    int64_t changes = 0;
    for (int64_t k = 1; k < nslices; ++k) changes += id[(size_t)k] != id[(size_t)k - 1];
    if (changes * 2 > nslices) return 0;
    // near diagonals: as many as 48 KiB of LDS hold (ring + mirror + two far slots + tables: three workgroups per CU; the 512^3 Poisson
    // matrix needs 35 KiB: four), grown from 0 outwards
    int lo = 0, hi = 0;
    std::vector<int> by_abs(table);
    std::sort(by_abs.begin(), by_abs.end(), [](int a, int b) { return std::llabs((long long)a) < std::llabs((long long)b); });
    for (int dlt : by_abs) {
        const int nlo = std::min(lo, dlt), nhi = std::max(hi, dlt);
        if ((long long)nhi - nlo > (1 << 20) || march_lds_bytes(nlo, nhi, value_bytes) > 48 * 1024) break;
        if (march_span_bytes(nlo, nhi, value_bytes) / value_bytes + S8_ROWS > 2048) break;       // the first window: <= 4 pairs per lane
        lo = nlo; hi = nhi;
    }
    // Slices per workgroup.  The prologue of a run costs about two slices (v8b), so long runs -- but a CU holds four workgroups
    // and wants at least two rounds of them: 512^3 (262 144 slices) run 16 / 32 / 64 = 0.540 / 0.469 / 0.484 ms; 256^3 (32 768
    // slices, the local matrix of one of eight ranks) run 8 / 16 / 32 = 0.068 / 0.057 / 0.075 ms, pair kernel 0.070.
    int run = 32;
    for (const long long want = nslices / (8ll * std::max(1, info(dev).cus)); run > 4 && run > want; ) run >>= 1;
    if (const char *e = env(ENV_VEXHIP_MARCH_RUN)) run = std::max(1, std::atoi(e));
    if (traversal && traversal->grid_blocks > 0 && traversal->chunk > 0) {
        while (run > 1 && traversal->chunk % run != 0) --run;
    }
    if (run < 2) return 0;
    // the (up to two) far diagonals are requested one slice ahead
    // A THIRD far diagonal has no slot: decode() would send every wave of every slice through the per-entry loop (the hot loop is
    // never entered, the ring traffic is still paid) -- slower than the pair product such a matrix had before.  Decline.
    out->nfar = 0;
    int nfar_all = 0;
    for (int dlt : by_abs)
        if (dlt < lo || dlt > hi) { if (out->nfar < 2) out->far[out->nfar++] = dlt; ++nfar_all; }
    out->far[2] = 0;
    if (nfar_all > 2) { std::memset(out, 0, sizeof(*out)); return 0; }
    out->lo = lo; out->hi = hi; out->run = run; out->x_last = x_last; out->usable = 1;
    return 0;
}

int64_t vexhip_sell8_last_fill_max_col(void) { return g_fill_max_col; }

int vexhip_spmv_sell8v_march_f64_i32(int dev, void *stream, int64_t n, double alpha, int append, int64_t w, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const double *values, const int32_t *cp, const int32_t *cc, const double *cv,
        const double *x, double *y, const vexhip_traversal *traversal, const vexhip_march *march)
{ return spmv_sell8v<double>(dev, stream, n, alpha, append, w, pool, deltas, values, cp, cc, cv, x, y, traversal, blocks, (march && march->usable) ? march : nullptr); }
int vexhip_spmv_sell8v_march_f32_i32(int dev, void *stream, int64_t n, float alpha, int append, int64_t w, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const float *values, const int32_t *cp, const int32_t *cc, const float *cv,
        const float *x, float *y, const vexhip_traversal *traversal, const vexhip_march *march)
{ return spmv_sell8v<float>(dev, stream, n, alpha, append, w, pool, deltas, values, cp, cc, cv, x, y, traversal, blocks, (march && march->usable) ? march : nullptr); }
int vexhip_csr_traversal_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col,
        int rows_per_block, vexhip_traversal *traversal)
{ return csr_traversal_impl<int32_t>(dev, stream, n, ptr, col, rows_per_block, traversal); }

int vexhip_sell8_fill_f64_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const double *val,
        int64_t w, const int32_t *deltas, int ndeltas, void *buf, vexhip_traversal *traversal)
{ return sell8_fill<double, int32_t>(dev, stream, n, ptr, col, val, w, deltas, ndeltas, buf, traversal); }

int vexhip_sell8_fill_f32_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const float *val,
        int64_t w, const int32_t *deltas, int ndeltas, void *buf, vexhip_traversal *traversal)
{ return sell8_fill<float, int32_t>(dev, stream, n, ptr, col, val, w, deltas, ndeltas, buf, traversal); }

int vexhip_spmv_sell8_f64_i32(int dev, void *stream, int64_t n, double alpha, int append, int64_t w,
        const void *buf, const int32_t *deltas, const int32_t *cp, const int32_t *cc, const double *cv,
        const double *x, double *y, const vexhip_traversal *traversal)
{ return spmv_sell8<double>(dev, stream, n, alpha, append, w, buf, deltas, cp, cc, cv, x, y, traversal); }

int vexhip_spmv_sell8_f32_i32(int dev, void *stream, int64_t n, float alpha, int append, int64_t w,
        const void *buf, const int32_t *deltas, const int32_t *cp, const int32_t *cc, const float *cv,
        const float *x, float *y, const vexhip_traversal *traversal)
{ return spmv_sell8<float>(dev, stream, n, alpha, append, w, buf, deltas, cp, cc, cv, x, y, traversal); }

} // extern "C"

VEXHIP_WARM_TU(sell8)
