// Device-side partition set-up of vex::SpMat: the LOCAL / REMOTE split of one device's row strip and its sorted ghost
// set, without touching the host.
//
// The reference does this on the host for every device (vexcl/spmat.hpp:291-378: a std::set of remote columns per
// device, then per-entry loops that fill the local and remote CSR; spmat/csr.inl:92-131 renumbers local columns to
// c - col_begin, hybrid_ell.inl:132-136 ghosts to their rank in the sorted set) -- O(nnz log) host work and two more
// host copies of the matrix.  Here, for a strip already in HBM (global column ids):
//   count   per row: entries with a column inside / outside [col_begin, col_end)            (one pass over col)
//   scan    exclusive scans of the two count arrays -> row pointers                          (scan.hip)
//   fill    local CSR (columns - col_begin) and the remote entries with GLOBAL columns, CSR order kept
//   rows    the rows that have remote entries, compacted (row-subset CSR: 2 planes of 64 for 512^3 over 8 GPUs)
//   ghosts  sort (sort.hip) + unique of the remote columns -> the sorted ghost set; then every remote column is
//           replaced by its rank in that set (binary search)
#include "common.hpp"

#include <vector>

namespace vexhip {
namespace {

__global__ __launch_bounds__(256)
void split_count_kernel(long long n, const int *__restrict__ ptr, const int *__restrict__ col, int c0, int c1,
        int *__restrict__ lcnt, int *__restrict__ rcnt)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int l = 0, r = 0;
        for (int j = ptr[i], e = ptr[i + 1]; j < e; ++j) { const int c = col[j]; if (c >= c0 && c < c1) ++l; else ++r; }
        lcnt[i] = l; rcnt[i] = r;
    }
}

template <typename V>
__global__ __launch_bounds__(256)
void split_fill_kernel(long long n, const int *__restrict__ ptr, const int *__restrict__ col, const V *__restrict__ val,
        int c0, int c1, const int *__restrict__ lptr, int *__restrict__ lcol, V *__restrict__ lval,
        const int *__restrict__ rptr, int *__restrict__ rcol, V *__restrict__ rval)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int l = lptr[i], r = rptr[i];
        for (int j = ptr[i], e = ptr[i + 1]; j < e; ++j) {
            const int c = col[j];
            if (c >= c0 && c < c1) { lcol[l] = c - c0; lval[l] = val[j]; ++l; }
            else { rcol[r] = c; rval[r] = val[j]; ++r; }
        }
    }
}

// flag[i] = 1 where a[i] starts a new run (a sorted); flag[i] = 1 where cnt[i] > 0
__global__ __launch_bounds__(256)
void run_start_kernel(long long n, const int *__restrict__ a, int *__restrict__ flag) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        flag[i] = (i == 0 || a[i] != a[i - 1]) ? 1 : 0;
}
__global__ __launch_bounds__(256)
void nonzero_kernel(long long n, const int *__restrict__ cnt, int *__restrict__ flag) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        flag[i] = cnt[i] > 0 ? 1 : 0;
}
// out[pos[i]] = a[i] where flag[i]
__global__ __launch_bounds__(256)
void compact_kernel(long long n, const int *__restrict__ a, const int *__restrict__ flag, const int *__restrict__ pos, int *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        if (flag[i]) out[pos[i]] = a[i];
}
// rows with remote entries: rows[pos[i]] = i, cptr[pos[i]] = rptr[i]   (cptr[nr] = rnnz is written by the host side)
__global__ __launch_bounds__(256)
void compact_rows_kernel(long long n, const int *__restrict__ flag, const int *__restrict__ pos, const int *__restrict__ rptr,
        int *__restrict__ rows, int *__restrict__ cptr) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        if (flag[i]) { rows[pos[i]] = (int)i; cptr[pos[i]] = rptr[i]; }
}
// c[k] <- rank of c[k] in the sorted set g[0..ng)
__global__ __launch_bounds__(256)
void rank_kernel(long long n, int *__restrict__ c, const int *__restrict__ g, int ng) {
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        const int v = c[k];
        int lo = 0, hi = ng;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (g[mid] < v) lo = mid + 1; else hi = mid; }
        c[k] = lo;
    }
}

// The strip with its ghost planes as ONE square matrix (transport "halo"): lo empty rows, the strip's rows, hi empty rows; columns
// counted from the first element of the lower ghost plane.  out_of_range counts the columns outside [0, lo + n + hi).
__global__ __launch_bounds__(256)
void extend_ptr_kernel(long long n, long long lo, long long hi, const int *__restrict__ ptr, int *__restrict__ ptr_ext) {
    const long long total = lo + n + hi + 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        ptr_ext[i] = i < lo ? 0 : (i <= lo + n ? ptr[i - lo] : ptr[n]);
}
__global__ __launch_bounds__(256)
void extend_col_kernel(long long nnz, const int *__restrict__ col, long long shift, long long ncols, int *__restrict__ col_ext, unsigned long long *out_of_range) {
    unsigned long long bad = 0;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += (long long)gridDim.x * blockDim.x) {
        const long long c = (long long)col[j] - shift;
        if (c < 0 || c >= ncols) ++bad;
        col_ext[j] = (int)c;
    }
    if (bad) atomicAdd(out_of_range, bad);
}

inline int grid_for(int dev, int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, (int64_t)info(dev).cus * 16)); }

struct scratch {
    std::vector<void *> p;
    ~scratch() { for (void *q : p) (void)hipFree(q); }
    template <typename T> int get(T **out, size_t count) {
        *out = nullptr;
        if (!count) count = 1;
        void *q = nullptr;
        if (int rc = check(hipMalloc(&q, count * sizeof(T)), __FILE__, __LINE__)) return rc;
        p.push_back(q); *out = static_cast<T *>(q);
        return 0;
    }
};

int exclusive_scan_i32(int dev, void *stream, const int *in, int *out, int64_t n, scratch &S) {
    void *tmp = nullptr;
    if (int rc = S.get(reinterpret_cast<char **>(&tmp), vexhip_scan_tmp_bytes(VEXHIP_I32, n))) return rc;
    const int zero = 0;
    return vexhip_scan(dev, stream, VEXHIP_I32, 1, &zero, in, out, n, tmp);
}

template <typename V>
int split(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const V *val, int64_t c0, int64_t c1,
        int phase, int64_t *sizes, int32_t *lptr, int32_t *lcol, V *lval,
        int32_t *rem_rows, int32_t *rem_ptr, int32_t *rem_col, V *rem_val, int32_t *ghosts)
{
    VEXHIP_REQUIRE(n >= 0 && c0 >= 0 && c1 >= c0 && c1 < (1ll << 31), "bad argument");
    VEXHIP_REQUIRE(sizes, "NULL sizes");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    if (n == 0) { sizes[0] = sizes[1] = sizes[2] = sizes[3] = 0; return 0; }
    VEXHIP_REQUIRE(ptr, "NULL row pointers");          // col / val may be NULL for a strip without entries
    scratch S;
    int *lcnt, *rcnt, *lp, *rp;
    if (int rc = S.get(&lcnt, (size_t)n + 1)) return rc;
    if (int rc = S.get(&rcnt, (size_t)n + 1)) return rc;
    if (int rc = S.get(&lp, (size_t)n + 1)) return rc;
    if (int rc = S.get(&rp, (size_t)n + 1)) return rc;
    VEXHIP_TRY(hipMemsetAsync(lcnt + n, 0, sizeof(int), s));
    VEXHIP_TRY(hipMemsetAsync(rcnt + n, 0, sizeof(int), s));
    split_count_kernel<<<grid_for(dev, n), 256, 0, s>>>(n, ptr, col, (int)c0, (int)c1, lcnt, rcnt);
    VEXHIP_LAUNCH_CHECK();
    if (int rc = exclusive_scan_i32(dev, stream, lcnt, lp, n + 1, S)) return rc;     // lp[n] = local nnz
    if (int rc = exclusive_scan_i32(dev, stream, rcnt, rp, n + 1, S)) return rc;
    int tot[2] = {0, 0};
    VEXHIP_TRY(hipMemcpyAsync(&tot[0], lp + n, sizeof(int), hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipMemcpyAsync(&tot[1], rp + n, sizeof(int), hipMemcpyDeviceToHost, s));
    // rows with remote entries
    int *flag, *pos;
    if (int rc = S.get(&flag, (size_t)n + 1)) return rc;
    if (int rc = S.get(&pos, (size_t)n + 1)) return rc;
    VEXHIP_TRY(hipMemsetAsync(flag + n, 0, sizeof(int), s));
    nonzero_kernel<<<grid_for(dev, n), 256, 0, s>>>(n, rcnt, flag);
    VEXHIP_LAUNCH_CHECK();
    if (int rc = exclusive_scan_i32(dev, stream, flag, pos, n + 1, S)) return rc;
    int nr = 0;
    VEXHIP_TRY(hipMemcpyAsync(&nr, pos + n, sizeof(int), hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    const int64_t lnnz = tot[0], rnnz = tot[1];
    sizes[0] = lnnz; sizes[1] = rnnz; sizes[2] = nr;
    if (phase == 0) { sizes[3] = -1; return 0; }            // ghost count comes with the fill

    VEXHIP_REQUIRE(lptr && (lnnz == 0 || (lcol && lval)), "NULL local arrays");
    VEXHIP_REQUIRE(rnnz == 0 || (rem_rows && rem_ptr && rem_col && rem_val && ghosts && val), "NULL remote arrays");
    VEXHIP_TRY(hipMemcpyAsync(lptr, lp, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToDevice, s));
    split_fill_kernel<V><<<grid_for(dev, n), 256, 0, s>>>(n, ptr, col, val, (int)c0, (int)c1, lp, lcol, lval, rp, rem_col, rem_val);
    VEXHIP_LAUNCH_CHECK();
    sizes[3] = 0;
    if (rnnz) {
        compact_rows_kernel<<<grid_for(dev, n), 256, 0, s>>>(n, flag, pos, rp, rem_rows, rem_ptr);
        VEXHIP_LAUNCH_CHECK();
        const int last = (int)rnnz;
        VEXHIP_TRY(hipMemcpyAsync(rem_ptr + nr, &last, sizeof(int), hipMemcpyHostToDevice, s));
        // sorted ghost set: sort a copy of the remote columns, keep the first of every run
        int *keys, *ktmp, *kflag, *kpos; void *stmp;
        if (int rc = S.get(&keys, (size_t)rnnz)) return rc;
        if (int rc = S.get(&ktmp, (size_t)rnnz)) return rc;
        if (int rc = S.get(&kflag, (size_t)rnnz + 1)) return rc;
        if (int rc = S.get(&kpos, (size_t)rnnz + 1)) return rc;
        if (int rc = S.get(reinterpret_cast<char **>(&stmp), vexhip_sort_tmp_bytes(VEXHIP_I32, rnnz))) return rc;
        VEXHIP_TRY(hipMemcpyAsync(keys, rem_col, sizeof(int) * (size_t)rnnz, hipMemcpyDeviceToDevice, s));
        if (int rc = vexhip_sort(dev, stream, VEXHIP_I32, 0, keys, ktmp, 0, nullptr, nullptr, rnnz, stmp)) return rc;
        VEXHIP_TRY(hipMemsetAsync(kflag + rnnz, 0, sizeof(int), s));
        run_start_kernel<<<grid_for(dev, rnnz), 256, 0, s>>>(rnnz, keys, kflag);
        VEXHIP_LAUNCH_CHECK();
        if (int rc = exclusive_scan_i32(dev, stream, kflag, kpos, rnnz + 1, S)) return rc;
        int ng = 0;
        VEXHIP_TRY(hipMemcpyAsync(&ng, kpos + rnnz, sizeof(int), hipMemcpyDeviceToHost, s));
        compact_kernel<<<grid_for(dev, rnnz), 256, 0, s>>>(rnnz, keys, kflag, kpos, ghosts);
        VEXHIP_LAUNCH_CHECK();
        VEXHIP_TRY(hipStreamSynchronize(s));
        sizes[3] = ng;
        rank_kernel<<<grid_for(dev, rnnz), 256, 0, s>>>(rnnz, rem_col, ghosts, ng);
        VEXHIP_LAUNCH_CHECK();
    }
    VEXHIP_TRY(hipStreamSynchronize(s));
    return 0;
}

} // namespace
} // namespace vexhip

using namespace vexhip;

extern "C" {

int vexhip_csr_extend_halo_i32(int dev, void *stream, int64_t n, int64_t nnz, const int32_t *ptr, const int32_t *col, int64_t col_begin,
        int64_t lo, int64_t hi, int32_t *ptr_ext, int32_t *col_ext, int64_t *out_of_range)
{
    VEXHIP_REQUIRE(n >= 0 && nnz >= 0 && lo >= 0 && hi >= 0 && col_begin >= 0 && lo + n + hi < (1ll << 31), "bad argument");
    VEXHIP_REQUIRE(ptr && ptr_ext && out_of_range && (nnz == 0 || (col && col_ext)), "NULL argument");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    scratch S;
    unsigned long long *bad = nullptr, hbad = 0;
    if (int rc = S.get(&bad, 1)) return rc;
    VEXHIP_TRY(hipMemsetAsync(bad, 0, sizeof(*bad), s));
    extend_ptr_kernel<<<grid_for(dev, lo + n + hi + 1), 256, 0, s>>>(n, lo, hi, ptr, ptr_ext);
    VEXHIP_LAUNCH_CHECK();
    if (nnz) {
        extend_col_kernel<<<grid_for(dev, nnz), 256, 0, s>>>(nnz, col, col_begin - lo, lo + n + hi, col_ext, bad);
        VEXHIP_LAUNCH_CHECK();
    }
    VEXHIP_TRY(hipMemcpyAsync(&hbad, bad, sizeof(hbad), hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    *out_of_range = (int64_t)hbad;
    return 0;
}

int vexhip_csr_split_sizes_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col,
        int64_t col_begin, int64_t col_end, int64_t *sizes)
{ return split<double>(dev, stream, n, ptr, col, nullptr, col_begin, col_end, 0, sizes, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr); }

int vexhip_csr_split_f64_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const double *val,
        int64_t col_begin, int64_t col_end, int64_t *sizes, int32_t *lptr, int32_t *lcol, double *lval,
        int32_t *rem_rows, int32_t *rem_ptr, int32_t *rem_col, double *rem_val, int32_t *ghosts)
{ return split<double>(dev, stream, n, ptr, col, val, col_begin, col_end, 1, sizes, lptr, lcol, lval, rem_rows, rem_ptr, rem_col, rem_val, ghosts); }

int vexhip_csr_split_f32_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const float *val,
        int64_t col_begin, int64_t col_end, int64_t *sizes, int32_t *lptr, int32_t *lcol, float *lval,
        int32_t *rem_rows, int32_t *rem_ptr, int32_t *rem_col, float *rem_val, int32_t *ghosts)
{ return split<float>(dev, stream, n, ptr, col, val, col_begin, col_end, 1, sizes, lptr, lcol, lval, rem_rows, rem_ptr, rem_col, rem_val, ghosts); }

} // extern "C"

VEXHIP_WARM_TU(split)
