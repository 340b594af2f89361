// vexhip_spmat: ONE device-resident sparse matrix object behind the C ABI.
//
// It owns what vex::SpMat decides at construction (reference: vexcl/spmat.hpp:84-104 picks
// SpMatCSR for CPU devices and SpMatHELL for GPUs; spmat/hybrid_ell.inl:60-216 builds the
// hybrid ELL part): here the storage selection
//     hybrid-ELL width / CSR tail  ->  <= 254 diagonals? 1-byte diagonal codes (SELL8)
//                                  ->  <= 255 values?    1-byte value codes   (SELL8V)
//                                  ->  otherwise 32-bit columns (SELL), or plain CSR when the ELL part is empty
// lives in this file once, instead of once in vexcl/spmat.hpp and once in vexcl_amd/ops.py (round 1).
// Both front ends now call create / apply / destroy.  Built from DEVICE CSR arrays: no host staging.
#include "common.hpp"
#include "halo.hpp"
#include "traversal.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>

namespace vexhip {

// 64-bit row pointers (sell8.hip, spmv.hip, misc.hip): set-up and the CSR product for matrices with 2^31 entries or more
int hell_analyze_p64(int dev, void *stream, int64_t n, const long long *ptr, int64_t *ell_width, int64_t *tail_nnz);
int hell_tail_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const double *val, int64_t w, int32_t *csr_ptr, int32_t *csr_col, double *csr_val);
int hell_tail_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const float *val, int64_t w, int32_t *csr_ptr, int32_t *csr_col, float *csr_val);
int sell8_analyze_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, int64_t w, int32_t *deltas, int *ndeltas);
void clear_max_col_hint();
extern int g_sell8_variant;            // sell8.hip (vexhip_spmv_sell8_set_variant): 0 = default products
// diagonals + values + largest ELL column in one pass over the CSR arrays (the fill that follows skips its own column pass)
int analyze_fused_p32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const double *val, int64_t w, int32_t *deltas, int *ndeltas, double *values, int *nvalues);
int analyze_fused_p32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const float *val, int64_t w, int32_t *deltas, int *ndeltas, float *values, int *nvalues);
int analyze_fused_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const double *val, int64_t w, int32_t *deltas, int *ndeltas, double *values, int *nvalues);
int analyze_fused_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const float *val, int64_t w, int32_t *deltas, int *ndeltas, float *values, int *nvalues);
int sell8v_analyze_p64(int dev, void *stream, int64_t n, const long long *ptr, const double *val, int64_t w, double *values, int *nvalues);
int sell8v_analyze_p64(int dev, void *stream, int64_t n, const long long *ptr, const float *val, int64_t w, float *values, int *nvalues);
int sell8v_fill_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const double *val, int64_t w, const int32_t *deltas, int ndeltas, const double *values, int nvalues, void *buf, vexhip_traversal *trav);
int sell8v_fill_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const float *val, int64_t w, const int32_t *deltas, int ndeltas, const float *values, int nvalues, void *buf, vexhip_traversal *trav);
int sell8_fill_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const double *val, int64_t w, const int32_t *deltas, int ndeltas, void *buf, vexhip_traversal *trav);
int sell8_fill_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const float *val, int64_t w, const int32_t *deltas, int ndeltas, void *buf, vexhip_traversal *trav);
int sell_fill_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const double *val, int64_t w, void *sell);
int sell_fill_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, const float *val, int64_t w, void *sell);
int spmv_csr_p64(int dev, void *stream, int64_t n, double alpha, int append, const long long *ptr, const int32_t *col, const double *val, const double *x, double *y, const vexhip_traversal *tr);
int spmv_csr_p64(int dev, void *stream, int64_t n, float alpha, int append, const long long *ptr, const int32_t *col, const float *val, const float *x, float *y, const vexhip_traversal *tr);
int csr_traversal_p64(int dev, void *stream, int64_t n, const long long *ptr, const int32_t *col, int rows_per_block, vexhip_traversal *traversal);
int grid_build_p32(int dev, void *stream, int64_t rows, const int32_t *ptr, const int32_t *col, const double *val,
        int32_t *deltas, double *values, int *ndeltas, int *nvalues, int64_t *ell_width, int64_t *x_last, vexhip_grid *out, int64_t min_cols);
int grid_build_p64(int dev, void *stream, int64_t rows, const long long *ptr, const int32_t *col, const double *val,
        int32_t *deltas, double *values, int *ndeltas, int *nvalues, int64_t *ell_width, int64_t *x_last, vexhip_grid *out, int64_t min_cols);
int grid_build_p32(int dev, void *stream, int64_t rows, const int32_t *ptr, const int32_t *col, const float *val,
        int32_t *deltas, float *values, int *ndeltas, int *nvalues, int64_t *ell_width, int64_t *x_last, vexhip_grid *out, int64_t min_cols);
int grid_build_p64(int dev, void *stream, int64_t rows, const long long *ptr, const int32_t *col, const float *val,
        int32_t *deltas, float *values, int *ndeltas, int *nvalues, int64_t *ell_width, int64_t *x_last, vexhip_grid *out, int64_t min_cols);
int plane_plan_from_grid(int dev, const vexhip_grid *grid, int64_t rows, vexhip_plane *out);
int plane_apply_halo(int dev, hipStream_t s, int64_t n_ext, double alpha, int append, int64_t w, const void *pool, const int32_t *blocks,
        const int32_t *deltas, const double *values, const double *x, double *y, const vexhip_plane *plane, halo_dev H);
int plane_apply_axpby(int dev, void *stream, int64_t n, double alpha, int zm, const double *zs, double beta, int64_t w, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const double *values, const double *x, double *y, const vexhip_plane *plane);
int grid_apply_axpby(int dev, void *stream, int64_t n, double alpha, int zm, const double *zs, double beta, const double *values,
        const double *x, double *y, const vexhip_grid *g);
int plane32_apply_axpby(int dev, void *stream, int64_t n, float alpha, int zm, const float *zs, float beta, int64_t w, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const float *values, const float *x, float *y, const vexhip_plane *plane);
int grid32_apply_axpby(int dev, void *stream, int64_t n, float alpha, int zm, const float *zs, float beta, const float *values,
        const float *x, float *y, const vexhip_grid *g);
int grid_apply_halo(int dev, hipStream_t s, int64_t n_ext, double alpha, int append, const double *values, const double *x, double *y,
        const vexhip_grid *g, halo_dev H);
int sell8v_runs_plan(int dev, void *stream, const void *pool, int64_t nblocks, int64_t w, const int *deltas, const double *values, int **desc_out);
int sell8v_runs_plan(int dev, void *stream, const void *pool, int64_t nblocks, int64_t w, const int *deltas, const float *values, int **desc_out);
int sell8v_runs_apply(int dev, void *stream, int64_t n, double alpha, int append, int64_t w, const void *pool, const int *blocks, const int *deltas, const double *values,
        const int *cp, const int *cc, const double *cv, const double *x, double *y, const vexhip_traversal *tr, const int *desc, long long x_last);
int sell8v_runs_apply(int dev, void *stream, int64_t n, float alpha, int append, int64_t w, const void *pool, const int *blocks, const int *deltas, const float *values,
        const int *cp, const int *cc, const float *cv, const float *x, float *y, const vexhip_traversal *tr, const int *desc, long long x_last);
int sell8_apply_halo(int dev, hipStream_t s, long long own_rows, double alpha, int append, int w, bool vcoded, const void *buf, const void *pool,
        const int *blocks, const int *deltas, const double *values, const double *x, double *y, halo_dev H);
int plane32_apply_halo(int dev, hipStream_t s, int64_t n_ext, float alpha, int append, int64_t w, const void *pool, const int32_t *blocks,
        const int32_t *deltas, const float *values, const float *x, float *y, const vexhip_plane *plane, halo_dev H);

namespace {

struct spmat {
    int dev = 0;
    int value_type = VEXHIP_F64;
    int format = VEXHIP_SPMAT_CSR;
    int64_t n = 0, nnz = 0, ell_w = 0, tail = 0;
    int64_t ell_max_col = -1;                                      // SELL8V: the largest column of the ELL part (or columns the caller vouched for): how far x reaches
    int *runs = nullptr;                                           // wide value-coded slices in a dictionary, decoded per wave and column (sell8.hip sell8v_runs_plan)
    void *sell = nullptr; int64_t sell_bytes = 0;
    // slice dictionary (sell8.hip): the codes of slice s are block blocks[s] of `pool` (dict_blocks distinct code blocks of
    // code_bytes each).  SELL8V: the slices are nothing but codes -- `sell` is freed; SELL8: `sell` keeps the values.
    int32_t *blocks = nullptr; void *pool = nullptr; int64_t dict_blocks = 0, code_bytes = 0;
    int32_t *deltas = nullptr; int ndeltas = -1;
    void *values = nullptr; int nvalues = -1;
    int32_t *csr_ptr = nullptr, *csr_col = nullptr; void *csr_val = nullptr;   // CSR tail, or the whole matrix (format CSR)
    long long *csr_ptr64 = nullptr;        // format CSR built from 64-bit row pointers (csr_ptr is NULL then)
    bool owns_csr = false;
    vexhip_traversal trav = {0, 0, 0, 0, nullptr};
    vexhip_march march = {0, 0, 0, 0, 0, 0, {0, 0, 0}};      // march product (sell8.hip): usable when the slices repeat in runs and the near diagonals fit a ring
    vexhip_plane plane = {0, 0, 0, 0, 0, 0, 0, 0, 0};              // plane product (plane.hip): 7-point pattern on 512-point lines; preferred to the march product
    vexhip_grid grid = {};                                         // grid product (grid.hip): 7-point pattern on lines of any length, where the plane product does not apply
    bool direct = false;                                           // stored by grid line straight from the CSR arrays (grid.hip grid_build): no SELL-512 slices, no dictionary
    char why[200] = {0};                                           // why this storage (the branch of build() that was taken)
};

template <typename T> int dmalloc(T **p, size_t count) {
    *p = nullptr;
    if (!count) return 0;
    return check(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)), __FILE__, __LINE__);
}

void release(spmat *A) {
    if (!A) return;
    (void)hipSetDevice(A->dev);
    if (A->sell) (void)hipFree(A->sell);
    if (A->blocks) (void)hipFree(A->blocks);
    if (A->runs) (void)hipFree(A->runs);
    if (A->pool) (void)hipFree(A->pool);
    if (A->deltas) (void)hipFree(A->deltas);
    if (A->values) (void)hipFree(A->values);
    (void)vexhip_sell8_grid_release(A->dev, &A->grid);
    if (A->owns_csr) {
        if (A->csr_ptr) (void)hipFree(A->csr_ptr);
        if (A->csr_ptr64) (void)hipFree(A->csr_ptr64);       // (both only when owned: a borrowed CSR matrix keeps the caller's arrays)
        if (A->csr_col) (void)hipFree(A->csr_col);
        if (A->csr_val) (void)hipFree(A->csr_val);
    }
    delete A;
}

// ---- THE selection of the product (round 6: one table instead of conditions spread over apply / get_info / the front ends) ----
// What a matrix's product launches, by storage and plan, first match wins:
//   value codes (SELL8V)   plane product      a plane plan: 7-point pattern on 512-point lines (fp64: x and y 16-byte aligned; fp32: any)
//                          grid product       a grid plan: the same pattern on lines of any length
//                          march product      a slice dictionary (the x window of the near diagonals in an LDS ring along runs of slices)
//                          pair product       otherwise
//   diagonal codes (SELL8) pair product       with or without a slice dictionary
//   32-bit columns (SELL)  pair / row kernel  (the slices dealt by XCD for unstructured matrices)
//   CSR                    row-block streaming kernel (32- or 64-bit row pointers)
enum product_kind { P_NONE, P_ZERO_FILL, P_PLANE64, P_PLANE32, P_GRID64, P_GRID32, P_MARCH, P_PAIR_CODES, P_PAIR_DICT_VALUES, P_PAIR_VALUES, P_SELL32, P_CSR32, P_CSR64 };
struct product_choice { product_kind kind; const char *kernel; const char *reason; };

inline product_choice select_product(const spmat *A, const void *x, const void *y)
{
    if (A->n == 0) return {P_NONE, "none", "no rows"};
    if (A->nnz == 0) return {P_ZERO_FILL, "hipMemsetAsync", "no entries: '=' zero-fills y (csr.inl:186-200)"};
    const bool f64 = A->value_type == VEXHIP_F64;
    const bool coded_by_line = (A->blocks || A->direct) && (g_sell8_variant == 0 || A->direct) && !A->tail;
    switch (A->format) {
        case VEXHIP_SPMAT_SELL8V:
            if (coded_by_line && A->plane.usable) {
                if (!f64) return {P_PLANE32, "sell8_plane_f32_kernel", "plane plan (512-point lines), float: four rows per lane"};
                if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) return {P_PLANE64, "sell8_plane_kernel", "plane plan (512-point lines)"};
            }
            if (A->grid.usable && (g_sell8_variant == 0 || A->direct) && !A->tail)
                return f64 ? product_choice{P_GRID64, "sell8_grid_kernel", A->plane.usable ? "plane plan, but x or y is not 16-byte aligned: the grid product addresses by element" : "grid plan (lines of any length)"}
                           : product_choice{P_GRID32, "sell8_grid_f32_kernel", "grid plan (lines of any length), float"};
            if (A->blocks && A->runs && A->ell_max_col >= 0 && g_sell8_variant == 0) return {P_MARCH, "sell8v_runs_kernel", "slice dictionary; entries decoded once per distinct slice, runs of three consecutive diagonals share a request"};
            if (A->blocks) return {P_MARCH, A->march.usable ? "sell8_march_kernel" : (A->ell_w <= 8 ? "sell8_pair_kernel" : "sell8v_kernel"), A->march.usable ? "slice dictionary + march plan" : "slice dictionary, no march plan: the codes of the distinct slices from the pool"};
            return {P_PAIR_CODES, A->ell_w <= 8 ? "sell8_pair_kernel" : "sell8v_kernel", "value codes, one code block per slice"};
        case VEXHIP_SPMAT_SELL8:
            if (A->blocks) return {P_PAIR_DICT_VALUES, "sell8_pair_kernel", "diagonal codes from the slice dictionary, values streamed"};
            return {P_PAIR_VALUES, A->ell_w <= 8 ? "sell8_pair_kernel" : "sell8_kernel", "diagonal codes + values streamed per slice"};
        case VEXHIP_SPMAT_SELL:
            return {P_SELL32, A->ell_w <= 8 ? "sell_pair_kernel" : "sell_kernel", A->trav.grid_blocks ? "32-bit columns, slices in the plan's order (strips of a banded matrix / an eighth per XCD)" : "32-bit columns, slices in storage order"};
        default:
            return A->csr_ptr64 ? product_choice{P_CSR64, "csr_stream2_kernel", "CSR arrays, 64-bit row pointers"} : product_choice{P_CSR32, "csr_stream2_kernel", "CSR arrays"};
    }
}

template <typename V>
__global__ __launch_bounds__(256) void scale_into_kernel(V *__restrict__ y, const V *z, V beta, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = beta * z[i];                 // (z may be y itself)
}

template <typename V> struct api;
template <> struct api<double> {
    static constexpr int type = VEXHIP_F64;
    static int hell_fill(int d, void *s, int64_t n, const int32_t *p, const int32_t *c, const double *v, int64_t w, int64_t pitch, int32_t *cp, int32_t *cc, double *cv)
    { return vexhip_hell_fill_f64_i32(d, s, n, p, c, v, w, pitch, nullptr, nullptr, cp, cc, cv); }
    static int v_analyze(int d, void *s, int64_t n, const int32_t *p, const double *v, int64_t w, double *vals, int *nv) { return vexhip_sell8v_analyze_f64_i32(d, s, n, p, v, w, vals, nv); }
    static int v_fill(int d, void *s, int64_t n, const int32_t *p, const int32_t *c, const double *v, int64_t w, const int32_t *dl, int nd, const double *vals, int nv, void *b, vexhip_traversal *t)
    { return vexhip_sell8v_fill_f64_i32(d, s, n, p, c, v, w, dl, nd, vals, nv, b, t); }
    static int d_fill(int d, void *s, int64_t n, const int32_t *p, const int32_t *c, const double *v, int64_t w, const int32_t *dl, int nd, void *b, vexhip_traversal *t)
    { return vexhip_sell8_fill_f64_i32(d, s, n, p, c, v, w, dl, nd, b, t); }
    static int s_fill(int d, void *s, int64_t n, const int32_t *p, const int32_t *c, const double *v, int64_t w, void *b) { return vexhip_sell_fill_f64_i32(d, s, n, p, c, v, w, b); }
    static int mul_v(int d, void *s, int64_t n, double a, int ap, int64_t w, const void *b, const int32_t *dl, const void *vals, const int32_t *cp, const int32_t *cc, const void *cv, const double *x, double *y, const vexhip_traversal *t)
    { return vexhip_spmv_sell8v_f64_i32(d, s, n, a, ap, w, b, dl, (const double *)vals, cp, cc, (const double *)cv, x, y, t); }
    static int mul_dd(int d, void *s, int64_t n, double a, int ap, int64_t w, const void *b, const void *pl, const int32_t *bl, const int32_t *dl, const int32_t *cp, const int32_t *cc, const void *cv, const double *x, double *y, const vexhip_traversal *t)
    { return vexhip_spmv_sell8_dict_f64_i32(d, s, n, a, ap, w, b, pl, bl, dl, cp, cc, (const double *)cv, x, y, t); }
    static int mm_dd(int d, void *s, int64_t n, int k, double a, int ap, int64_t w, const void *b, const void *pl, const int32_t *bl, const int32_t *dl, const int32_t *cp, const int32_t *cc, const void *cv, const double *const *x, double *const *y, const vexhip_traversal *t)
    { return vexhip_spmm_sell8_dict_f64_i32(d, s, n, k, a, ap, w, b, pl, bl, dl, cp, cc, (const double *)cv, x, y, t); }
    static int mul_vd(int d, void *s, int64_t n, double a, int ap, int64_t w, const void *b, const int32_t *bl, const int32_t *dl, const void *vals, const int32_t *cp, const int32_t *cc, const void *cv, const double *x, double *y, const vexhip_traversal *t, const vexhip_march *m)
    { return vexhip_spmv_sell8v_march_f64_i32(d, s, n, a, ap, w, b, bl, dl, (const double *)vals, cp, cc, (const double *)cv, x, y, t, m); }
    static int mm_vd(int d, void *s, int64_t n, int k, double a, int ap, int64_t w, const void *b, const int32_t *bl, const int32_t *dl, const void *vals, const int32_t *cp, const int32_t *cc, const void *cv, const double *const *x, double *const *y, const vexhip_traversal *t)
    { return vexhip_spmm_sell8v_dict_f64_i32(d, s, n, k, a, ap, w, b, bl, dl, (const double *)vals, cp, cc, (const double *)cv, x, y, t); }
    static int mul_d(int d, void *s, int64_t n, double a, int ap, int64_t w, const void *b, const int32_t *dl, const int32_t *cp, const int32_t *cc, const void *cv, const double *x, double *y, const vexhip_traversal *t)
    { return vexhip_spmv_sell8_f64_i32(d, s, n, a, ap, w, b, dl, cp, cc, (const double *)cv, x, y, t); }
    static int mul_s(int d, void *s, int64_t n, double a, int ap, int64_t w, const void *b, const int32_t *cp, const int32_t *cc, const void *cv, const double *x, double *y, const vexhip_traversal *t)
    { return vexhip_spmv_sell_f64_i32(d, s, n, a, ap, w, b, cp, cc, (const double *)cv, x, y, t); }
    static int mul_c(int d, void *s, int64_t n, double a, int ap, const int32_t *p, const int32_t *c, const void *v, const double *x, double *y, const vexhip_traversal *t)
    { return vexhip_spmv_csr_ordered_f64_i32(d, s, n, a, ap, p, c, (const double *)v, x, y, t); }
    static int mm_v(int d, void *s, int64_t n, int k, double a, int ap, int64_t w, const void *b, const int32_t *dl, const void *vals, const int32_t *cp, const int32_t *cc, const void *cv, const double *const *x, double *const *y, const vexhip_traversal *t)
    { return vexhip_spmm_sell8v_f64_i32(d, s, n, k, a, ap, w, b, dl, (const double *)vals, cp, cc, (const double *)cv, x, y, t); }
    static int mm_d(int d, void *s, int64_t n, int k, double a, int ap, int64_t w, const void *b, const int32_t *dl, const int32_t *cp, const int32_t *cc, const void *cv, const double *const *x, double *const *y, const vexhip_traversal *t)
    { return vexhip_spmm_sell8_f64_i32(d, s, n, k, a, ap, w, b, dl, cp, cc, (const double *)cv, x, y, t); }
    static int mm_s(int d, void *s, int64_t n, int k, double a, int ap, int64_t w, const void *b, const int32_t *cp, const int32_t *cc, const void *cv, const double *const *x, double *const *y, const vexhip_traversal *t)
    { return vexhip_spmm_sell_f64_i32(d, s, n, k, a, ap, w, b, cp, cc, (const double *)cv, x, y, t); }
};
template <> struct api<float> {
    static constexpr int type = VEXHIP_F32;
    static int hell_fill(int d, void *s, int64_t n, const int32_t *p, const int32_t *c, const float *v, int64_t w, int64_t pitch, int32_t *cp, int32_t *cc, float *cv)
    { return vexhip_hell_fill_f32_i32(d, s, n, p, c, v, w, pitch, nullptr, nullptr, cp, cc, cv); }
    static int v_analyze(int d, void *s, int64_t n, const int32_t *p, const float *v, int64_t w, float *vals, int *nv) { return vexhip_sell8v_analyze_f32_i32(d, s, n, p, v, w, vals, nv); }
    static int v_fill(int d, void *s, int64_t n, const int32_t *p, const int32_t *c, const float *v, int64_t w, const int32_t *dl, int nd, const float *vals, int nv, void *b, vexhip_traversal *t)
    { return vexhip_sell8v_fill_f32_i32(d, s, n, p, c, v, w, dl, nd, vals, nv, b, t); }
    static int d_fill(int d, void *s, int64_t n, const int32_t *p, const int32_t *c, const float *v, int64_t w, const int32_t *dl, int nd, void *b, vexhip_traversal *t)
    { return vexhip_sell8_fill_f32_i32(d, s, n, p, c, v, w, dl, nd, b, t); }
    static int s_fill(int d, void *s, int64_t n, const int32_t *p, const int32_t *c, const float *v, int64_t w, void *b) { return vexhip_sell_fill_f32_i32(d, s, n, p, c, v, w, b); }
    static int mul_v(int d, void *s, int64_t n, float a, int ap, int64_t w, const void *b, const int32_t *dl, const void *vals, const int32_t *cp, const int32_t *cc, const void *cv, const float *x, float *y, const vexhip_traversal *t)
    { return vexhip_spmv_sell8v_f32_i32(d, s, n, a, ap, w, b, dl, (const float *)vals, cp, cc, (const float *)cv, x, y, t); }
    static int mul_dd(int d, void *s, int64_t n, float a, int ap, int64_t w, const void *b, const void *pl, const int32_t *bl, const int32_t *dl, const int32_t *cp, const int32_t *cc, const void *cv, const float *x, float *y, const vexhip_traversal *t)
    { return vexhip_spmv_sell8_dict_f32_i32(d, s, n, a, ap, w, b, pl, bl, dl, cp, cc, (const float *)cv, x, y, t); }
    static int mm_dd(int d, void *s, int64_t n, int k, float a, int ap, int64_t w, const void *b, const void *pl, const int32_t *bl, const int32_t *dl, const int32_t *cp, const int32_t *cc, const void *cv, const float *const *x, float *const *y, const vexhip_traversal *t)
    { return vexhip_spmm_sell8_dict_f32_i32(d, s, n, k, a, ap, w, b, pl, bl, dl, cp, cc, (const float *)cv, x, y, t); }
    static int mul_vd(int d, void *s, int64_t n, float a, int ap, int64_t w, const void *b, const int32_t *bl, const int32_t *dl, const void *vals, const int32_t *cp, const int32_t *cc, const void *cv, const float *x, float *y, const vexhip_traversal *t, const vexhip_march *m)
    { return vexhip_spmv_sell8v_march_f32_i32(d, s, n, a, ap, w, b, bl, dl, (const float *)vals, cp, cc, (const float *)cv, x, y, t, m); }
    static int mm_vd(int d, void *s, int64_t n, int k, float a, int ap, int64_t w, const void *b, const int32_t *bl, const int32_t *dl, const void *vals, const int32_t *cp, const int32_t *cc, const void *cv, const float *const *x, float *const *y, const vexhip_traversal *t)
    { return vexhip_spmm_sell8v_dict_f32_i32(d, s, n, k, a, ap, w, b, bl, dl, (const float *)vals, cp, cc, (const float *)cv, x, y, t); }
    static int mul_d(int d, void *s, int64_t n, float a, int ap, int64_t w, const void *b, const int32_t *dl, const int32_t *cp, const int32_t *cc, const void *cv, const float *x, float *y, const vexhip_traversal *t)
    { return vexhip_spmv_sell8_f32_i32(d, s, n, a, ap, w, b, dl, cp, cc, (const float *)cv, x, y, t); }
    static int mul_s(int d, void *s, int64_t n, float a, int ap, int64_t w, const void *b, const int32_t *cp, const int32_t *cc, const void *cv, const float *x, float *y, const vexhip_traversal *t)
    { return vexhip_spmv_sell_f32_i32(d, s, n, a, ap, w, b, cp, cc, (const float *)cv, x, y, t); }
    static int mul_c(int d, void *s, int64_t n, float a, int ap, const int32_t *p, const int32_t *c, const void *v, const float *x, float *y, const vexhip_traversal *t)
    { return vexhip_spmv_csr_ordered_f32_i32(d, s, n, a, ap, p, c, (const float *)v, x, y, t); }
    static int mm_v(int d, void *s, int64_t n, int k, float a, int ap, int64_t w, const void *b, const int32_t *dl, const void *vals, const int32_t *cp, const int32_t *cc, const void *cv, const float *const *x, float *const *y, const vexhip_traversal *t)
    { return vexhip_spmm_sell8v_f32_i32(d, s, n, k, a, ap, w, b, dl, (const float *)vals, cp, cc, (const float *)cv, x, y, t); }
    static int mm_d(int d, void *s, int64_t n, int k, float a, int ap, int64_t w, const void *b, const int32_t *dl, const int32_t *cp, const int32_t *cc, const void *cv, const float *const *x, float *const *y, const vexhip_traversal *t)
    { return vexhip_spmm_sell8_f32_i32(d, s, n, k, a, ap, w, b, dl, cp, cc, (const float *)cv, x, y, t); }
    static int mm_s(int d, void *s, int64_t n, int k, float a, int ap, int64_t w, const void *b, const int32_t *cp, const int32_t *cc, const void *cv, const float *const *x, float *const *y, const vexhip_traversal *t)
    { return vexhip_spmm_sell_f32_i32(d, s, n, k, a, ap, w, b, cp, cc, (const float *)cv, x, y, t); }
};

// Slice dictionary (sell8.hip): do the code blocks of the slices repeat?  Up to 128 distinct blocks (<= 1 MiB for width 7-8:
// L1 / L2 resident) replace the code stream; anything less regular keeps it.  whole_slice: the slice is nothing but codes
// (SELL8V) -- the per-slice storage is released; otherwise (SELL8) the values stay where they are.
int make_dictionary(spmat *A, void *stream, int flags, int64_t code_bytes, bool whole_slice)
{
    const int64_t ns = (A->n + 511) / 512, stride = A->sell_bytes / ns;
    // (round 6: also wider than eight columns -- a constant-coefficient 27-point operator keeps 28 KiB of codes per DISTINCT slice instead of
    //  54 bytes per row; the march and plane plans are for the 7-point pattern)
    if ((flags & VEXHIP_SPMAT_NO_DICTIONARY) || ns < 64 || A->ell_w > 32 || (A->ell_w > 8 && !whole_slice)) return 0;
    hipStream_t s = as_stream(stream);
    const int64_t cap = 128;
    void *big = nullptr; int64_t nb = -1;
    if (int rc = dmalloc(&A->blocks, (size_t)ns)) return rc;
    VEXHIP_TRY(hipMalloc(&big, (size_t)(cap * code_bytes)));
    int rc = vexhip_slice_dictionary(A->dev, stream, ns, stride, code_bytes, A->sell, cap, A->blocks, big, &nb);
    if (rc == 0 && nb > 0 && nb * 4 <= ns) {                                   // worth it only if the slices really repeat
        hipError_t e = hipMalloc(&A->pool, (size_t)(nb * code_bytes));          // the pool at its real size
        if (e == hipSuccess) e = hipMemcpyAsync(A->pool, big, (size_t)(nb * code_bytes), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipFree(big);
        if (e != hipSuccess) return check(e, __FILE__, __LINE__);
        A->dict_blocks = nb; A->code_bytes = code_bytes;
        if (whole_slice) { (void)hipFree(A->sell); A->sell = nullptr; A->sell_bytes = 0; }
        if (whole_slice && A->ell_w > 9 && g_sell8_variant == 0) {            // 19- / 27-point rows: decoded once per distinct slice (sell8.hip, runs of three diagonals)
            int rc2 = A->value_type == VEXHIP_F64 ? sell8v_runs_plan(A->dev, stream, A->pool, nb, A->ell_w, A->deltas, (const double *)A->values, &A->runs)
                                                  : sell8v_runs_plan(A->dev, stream, A->pool, nb, A->ell_w, A->deltas, (const float *)A->values, &A->runs);
            if (rc2) return rc2;
        }
        if (whole_slice && A->ell_w <= 8 && !(flags & VEXHIP_SPMAT_NO_MARCH)) {
            const int vb = A->value_type == VEXHIP_F64 ? 8 : 4;
            if (int rc2 = vexhip_sell8_march_plan(A->dev, stream, A->deltas, A->ndeltas, A->blocks, ns, vb,
                                                  &A->trav, std::max<int64_t>(vexhip_sell8_last_fill_max_col(), ((flags & VEXHIP_SPMAT_SQUARE) ? A->n : 0) - 1), &A->march)) return rc2;
            if (!(flags & VEXHIP_SPMAT_NO_PLANE) && !env(ENV_VEXHIP_NO_PLANE512))      // (A/B: the grid product on 512-point lines)
                if (int rc2 = vexhip_sell8_plane_plan(A->dev, stream, A->deltas, A->ndeltas, A->blocks, ns, A->pool, nb, A->ell_w, A->n, A->tail, vb,
                                                      std::max<int64_t>(vexhip_sell8_last_fill_max_col(), ((flags & VEXHIP_SPMAT_SQUARE) ? A->n : 0) - 1), &A->plane)) return rc2;
        }
        return 0;
    }
    (void)hipFree(big); (void)hipFree(A->blocks); A->blocks = nullptr;
    return rc;
}

// set-up steps by row-pointer type
template <typename V> struct setup32 {
    typedef api<V> F;
    static int analyze(int d, void *s, int64_t n, const int32_t *p, int64_t *w, int64_t *t) { return vexhip_hell_analyze_i32(d, s, n, p, w, t); }
    static int tail(int d, void *s, int64_t n, const int32_t *p, const int32_t *c, const V *v, int64_t w, int32_t *cp, int32_t *cc, V *cv) { return F::hell_fill(d, s, n, p, c, v, w, (n + 15) / 16 * 16, cp, cc, cv); }
    static int d_analyze(int d, void *s, int64_t n, const int32_t *p, const int32_t *c, int64_t w, int32_t *dl, int *nd) { return vexhip_sell8_analyze_i32(d, s, n, p, c, w, dl, nd); }
    static int v_analyze(int d, void *s, int64_t n, const int32_t *p, const V *v, int64_t w, V *vals, int *nv) { return F::v_analyze(d, s, n, p, v, w, vals, nv); }
    static int v_fill(int d, void *s, int64_t n, const int32_t *p, const int32_t *c, const V *v, int64_t w, const int32_t *dl, int nd, const V *vals, int nv, void *b, vexhip_traversal *t) { return F::v_fill(d, s, n, p, c, v, w, dl, nd, vals, nv, b, t); }
    static int d_fill(int d, void *s, int64_t n, const int32_t *p, const int32_t *c, const V *v, int64_t w, const int32_t *dl, int nd, void *b, vexhip_traversal *t) { return F::d_fill(d, s, n, p, c, v, w, dl, nd, b, t); }
    static int s_fill(int d, void *s, int64_t n, const int32_t *p, const int32_t *c, const V *v, int64_t w, void *b) { return F::s_fill(d, s, n, p, c, v, w, b); }
    static int fused(int d, void *s, int64_t n, const int32_t *p, const int32_t *c, const V *v, int64_t w, int32_t *dl, int *nd, V *vals, int *nv) { return analyze_fused_p32(d, s, n, p, c, v, w, dl, nd, vals, nv); }
};
template <typename V> struct setup64 {
    static int analyze(int d, void *s, int64_t n, const long long *p, int64_t *w, int64_t *t) { return hell_analyze_p64(d, s, n, p, w, t); }
    static int tail(int d, void *s, int64_t n, const long long *p, const int32_t *c, const V *v, int64_t w, int32_t *cp, int32_t *cc, V *cv) { return hell_tail_p64(d, s, n, p, c, v, w, cp, cc, cv); }
    static int d_analyze(int d, void *s, int64_t n, const long long *p, const int32_t *c, int64_t w, int32_t *dl, int *nd) { return sell8_analyze_p64(d, s, n, p, c, w, dl, nd); }
    static int v_analyze(int d, void *s, int64_t n, const long long *p, const V *v, int64_t w, V *vals, int *nv) { return sell8v_analyze_p64(d, s, n, p, v, w, vals, nv); }
    static int v_fill(int d, void *s, int64_t n, const long long *p, const int32_t *c, const V *v, int64_t w, const int32_t *dl, int nd, const V *vals, int nv, void *b, vexhip_traversal *t) { return sell8v_fill_p64(d, s, n, p, c, v, w, dl, nd, vals, nv, b, t); }
    static int d_fill(int d, void *s, int64_t n, const long long *p, const int32_t *c, const V *v, int64_t w, const int32_t *dl, int nd, void *b, vexhip_traversal *t) { return sell8_fill_p64(d, s, n, p, c, v, w, dl, nd, b, t); }
    static int s_fill(int d, void *s, int64_t n, const long long *p, const int32_t *c, const V *v, int64_t w, void *b) { return sell_fill_p64(d, s, n, p, c, v, w, b); }
    static int fused(int d, void *s, int64_t n, const long long *p, const int32_t *c, const V *v, int64_t w, int32_t *dl, int *nd, V *vals, int *nv) { return analyze_fused_p64(d, s, n, p, c, v, w, dl, nd, vals, nv); }
};

// P = int32_t or long long (row pointers); columns are 32-bit either way
template <typename V, typename P>
int build(spmat *A, void *stream, int64_t n, const P *ptr, const int32_t *col, const V *val, int format, int flags)
{
    typedef api<V> F;
    typedef typename std::conditional<sizeof(P) == 4, setup32<V>, setup64<V>>::type S;
    constexpr bool p64 = sizeof(P) == 8;
    const int dev = A->dev;
    A->n = n; A->value_type = F::type;
    clear_max_col_hint();                      // a hint left by a build that failed half-way must not reach this matrix's fill
    if (n == 0) { A->format = VEXHIP_SPMAT_CSR; std::snprintf(A->why, sizeof A->why, "no rows"); return 0; }
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    setup_trace trace(s);
    P last = 0;
    VEXHIP_TRY(hipMemcpyAsync(&last, ptr + n, sizeof(P), hipMemcpyDeviceToHost, s));
    VEXHIP_TRY(hipStreamSynchronize(s));
    A->nnz = (int64_t)last;
    trace.mark("entry count");
    // VEXHIP_SPMAT_SQUARE: x has at least as many elements as the matrix has rows, whatever the largest column that occurs
    const int64_t min_cols = (flags & VEXHIP_SPMAT_SQUARE) ? n : 0;

    // A 7-point pattern on a grid with a handful of distinct values: stored by grid line in ONE pass over the CSR arrays (grid.hip
    // grid_build: no ELL analysis, no table pass, no per-slice codes, no dictionary, no plans from read-backs).  Declined
    // (usable = 0, after a probe of a few thousand rows in most cases): the SELL-512 set-up below.
    // (fp32, round 5: the same storage; its products are plane32.hip on 512-point lines and grid32.hip on lines of any length)
    if ((format == VEXHIP_SPMAT_AUTO || format == VEXHIP_SPMAT_SELL8V) && A->nnz > 0
        && !(flags & (VEXHIP_SPMAT_NO_DICTIONARY | VEXHIP_SPMAT_NO_MARCH | VEXHIP_SPMAT_NO_PLANE | VEXHIP_SPMAT_NO_GRID_BUILD))) {
        V *vals = nullptr;
        if (int rc = dmalloc(&A->deltas, 256)) return rc;
        if (int rc = dmalloc(&vals, 256)) return rc;
        A->values = vals;
        int nd = -1, nv = -1; int64_t gw = 0, x_last = -1;
        int rc;
        if constexpr (p64) rc = grid_build_p64(dev, stream, n, ptr, col, val, A->deltas, vals, &nd, &nv, &gw, &x_last, &A->grid, min_cols);
        else rc = grid_build_p32(dev, stream, n, ptr, col, val, A->deltas, vals, &nd, &nv, &gw, &x_last, &A->grid, min_cols);
        if (rc) return rc;
        trace.mark("grid build");
        if (A->grid.usable) {
            A->ndeltas = nd; A->nvalues = nv; A->ell_w = gw; A->tail = 0; A->format = VEXHIP_SPMAT_SELL8V; A->direct = true;
            std::snprintf(A->why, sizeof A->why, "stored by grid line in one pass over the CSR arrays: a 7-point pattern on %d-point lines (%d lines per plane), %d distinct values, %d classes of lines",
                          (int)A->grid.nx, (int)A->grid.lines_per_plane, nv, (int)A->grid.classes);
            if (!env(ENV_VEXHIP_NO_PLANE512))
                if (int rc2 = plane_plan_from_grid(dev, &A->grid, n, &A->plane)) return rc2;       // 512-point lines: the plane kernel reads the same tables
            trace.mark("plane plan");
            return 0;
        }
        (void)hipFree(A->deltas); A->deltas = nullptr;
        (void)hipFree(A->values); A->values = nullptr;
    }
    int64_t w = 0, tail = 0;
    if (format != VEXHIP_SPMAT_CSR && A->nnz > 0)
        if (int rc = S::analyze(dev, stream, n, ptr, &w, &tail)) return rc;
    trace.mark("ELL width");
    if (format == VEXHIP_SPMAT_CSR || w == 0) {
        // the CSR arrays as they are: borrowed (the caller keeps them alive) or copied
        A->format = VEXHIP_SPMAT_CSR;
        std::snprintf(A->why, sizeof A->why, format == VEXHIP_SPMAT_CSR ? "CSR was asked for" : "the hybrid-ELL rule (hybrid_ell.inl:103-110) gives width 0: every row goes to the CSR part");
        P *own_ptr = nullptr;
        if (flags & VEXHIP_SPMAT_BORROW_CSR) {
            own_ptr = const_cast<P *>(ptr); A->csr_col = const_cast<int32_t *>(col); A->csr_val = const_cast<V *>(val);
        } else {
            A->owns_csr = true;
            V *cv = nullptr;
            if (int rc = dmalloc(&own_ptr, (size_t)n + 1)) return rc;
            if constexpr (p64) A->csr_ptr64 = own_ptr; else A->csr_ptr = own_ptr;      // owned from here on (release frees it)
            if (int rc = dmalloc(&A->csr_col, (size_t)A->nnz)) return rc;
            if (int rc = dmalloc(&cv, (size_t)A->nnz)) return rc;
            A->csr_val = cv;
            VEXHIP_TRY(hipMemcpyAsync(own_ptr, ptr, sizeof(P) * ((size_t)n + 1), hipMemcpyDeviceToDevice, s));
            if (A->nnz) {
                VEXHIP_TRY(hipMemcpyAsync(A->csr_col, col, sizeof(int32_t) * (size_t)A->nnz, hipMemcpyDeviceToDevice, s));
                VEXHIP_TRY(hipMemcpyAsync(cv, val, sizeof(V) * (size_t)A->nnz, hipMemcpyDeviceToDevice, s));
            }
        }
        if constexpr (p64) A->csr_ptr64 = own_ptr; else A->csr_ptr = own_ptr;
        if (A->nnz) {                             // strips for banded matrices
            if constexpr (p64) (void)csr_traversal_p64(dev, stream, n, A->csr_ptr64, A->csr_col, 256, &A->trav);
            else (void)vexhip_csr_traversal_i32(dev, stream, n, A->csr_ptr, A->csr_col, 256, &A->trav);
        }
        VEXHIP_TRY(hipStreamSynchronize(s));
        return 0;
    }

    A->ell_w = w; A->tail = tail;
    if (tail) {      // rows wider than the ELL width keep their tail in CSR (hybrid_ell.inl:166-198)
        VEXHIP_REQUIRE(tail < (1ll << 31), "the CSR tail of the hybrid-ELL storage holds 2^31 or more entries");
        A->owns_csr = true;
        V *cv = nullptr;
        if (int rc = dmalloc(&A->csr_ptr, (size_t)n + 1)) return rc;
        if (int rc = dmalloc(&A->csr_col, (size_t)tail)) return rc;
        if (int rc = dmalloc(&cv, (size_t)tail)) return rc;
        A->csr_val = cv;
        if (int rc = S::tail(dev, stream, n, ptr, col, val, w, A->csr_ptr, A->csr_col, cv)) return rc;
    }
    int nd = -1, nv = -1;
    if (format == VEXHIP_SPMAT_AUTO || format == VEXHIP_SPMAT_SELL8V) {
        // both analyses are wanted: ONE pass over the CSR arrays finds the diagonals, the values and the largest ELL column
        V *vals = nullptr;
        if (int rc = dmalloc(&A->deltas, 256)) return rc;
        if (int rc = dmalloc(&vals, 256)) return rc;
        A->values = vals;
        if (int rc = S::fused(dev, stream, n, ptr, col, val, w, A->deltas, &nd, vals, &nv)) return rc;
        trace.mark("diagonals + values");
    } else if (format != VEXHIP_SPMAT_SELL) {
        if (int rc = dmalloc(&A->deltas, 256)) return rc;
        if (int rc = S::d_analyze(dev, stream, n, ptr, col, w, A->deltas, &nd)) return rc;
    }
    if (nd > 0) {
        A->ndeltas = nd;
        if (nv > 0) {
            A->nvalues = nv; A->format = VEXHIP_SPMAT_SELL8V;
            std::snprintf(A->why, sizeof A->why, "SELL-512 with 1-byte diagonal and value codes: ELL width %lld, %d diagonals (<= 254), %d distinct values (<= 255), %lld entries in the CSR tail",
                          (long long)w, nd, nv, (long long)tail);
            A->sell_bytes = vexhip_sell8v_bytes(n, w);
            VEXHIP_TRY(hipMalloc(&A->sell, (size_t)A->sell_bytes));
            trace.mark("allocate slices");
            if (int rc = S::v_fill(dev, stream, n, ptr, col, val, w, A->deltas, nd, (const V *)A->values, nv, A->sell, &A->trav)) return rc;
            A->ell_max_col = std::max<int64_t>(vexhip_sell8_last_fill_max_col(), min_cols - 1);
            trace.mark("fill");
            if (int rc = make_dictionary(A, stream, flags, A->sell_bytes / ((n + 511) / 512), true)) return rc;
            trace.mark("dictionary + plans");
            // a 7-point pattern on grid lines of another length than 512: the matrix by grid line (grid.hip), from the slices'
            // codes wherever they are now (the pool of a dictionary, or the per-slice buffer)
            if (!A->plane.usable && !tail
                && !(flags & (VEXHIP_SPMAT_NO_DICTIONARY | VEXHIP_SPMAT_NO_MARCH | VEXHIP_SPMAT_NO_PLANE)))
                if (int rc = vexhip_sell8_grid_plan(dev, stream, A->deltas, nd, A->blocks ? A->pool : A->sell, A->blocks, w, n, tail, (int)sizeof(V),
                                                    std::max<int64_t>(vexhip_sell8_last_fill_max_col(), min_cols - 1), &A->grid)) return rc;
            trace.mark("grid plan");
        } else {
            if (A->values) { (void)hipFree(A->values); A->values = nullptr; }
            A->format = VEXHIP_SPMAT_SELL8;
            std::snprintf(A->why, sizeof A->why, "SELL-512 with 1-byte diagonal codes and stored values: ELL width %lld, %d diagonals (<= 254), more than 255 distinct values%s, %lld entries in the CSR tail",
                          (long long)w, nd, format == VEXHIP_SPMAT_SELL8 ? " (or format SELL8 asked for)" : "", (long long)tail);
            A->sell_bytes = vexhip_sell8_bytes(n, w, (int)sizeof(V));
            VEXHIP_TRY(hipMalloc(&A->sell, (size_t)A->sell_bytes));
            if (int rc = S::d_fill(dev, stream, n, ptr, col, val, w, A->deltas, nd, A->sell, &A->trav)) return rc;
            if (int rc = make_dictionary(A, stream, flags, ((w + 1) / 2) * 1024, false)) return rc;
        }
    } else {
        if (A->deltas) { (void)hipFree(A->deltas); A->deltas = nullptr; }
        if (A->values) { (void)hipFree(A->values); A->values = nullptr; }
        A->format = VEXHIP_SPMAT_SELL;
        std::snprintf(A->why, sizeof A->why, "SELL-512 with 32-bit columns: ELL width %lld, %s, %lld entries in the CSR tail", (long long)w,
                      format == VEXHIP_SPMAT_SELL ? "format SELL asked for" : "more than 254 distinct diagonals (an unstructured matrix)", (long long)tail);
        A->sell_bytes = vexhip_sell_bytes(n, w, (int)sizeof(V));
        VEXHIP_TRY(hipMalloc(&A->sell, (size_t)A->sell_bytes));
        if (int rc = S::s_fill(dev, stream, n, ptr, col, val, w, A->sell)) return rc;
        if (int rc = vexhip_sell_order_i32(dev, stream, n, w, (int)sizeof(V), A->sell, 0, nullptr, 0, &A->trav)) return rc;
        // Round 6 -- no constant bands (an unstructured matrix): every XCD walks ONE contiguous eighth of the slices instead of every
        // eighth slice.  Where the columns of a row lie near the row (a mesh-ordered operator) the lines of x a slice gathers from are
        // the ones its neighbours use: walked in order by one XCD they stay in ITS L2; dealt round-robin, all eight L2s fetch all of
        // them (banded16 of bench.py: x traffic 39 x its size).  Matrices without locality lose nothing; a large CSR tail (rows of very
        // different cost) keeps the plain order, which the dispatcher balances.
        const int64_t ns = (n + 511) / 512;
        if (A->trav.grid_blocks == 0 && ns >= 4096 && tail * 8 <= A->nnz && !(flags & VEXHIP_SPMAT_PLAIN_ORDER)) {
            const int64_t per = (ns + 7) / 8;
            A->trav.grid_blocks = 8 * per; A->trav.chunk = per; A->trav.planes = 1; A->trav.plane_blocks = ns; A->trav.order = nullptr;
        }
    }
    clear_max_col_hint();
    VEXHIP_TRY(hipStreamSynchronize(s));
    return 0;
}

template <typename V, typename P>
int create(int dev, void *stream, int64_t n, const P *ptr, const int32_t *col, const V *val, int format, int flags, vexhip_spmat **out)
{
    VEXHIP_REQUIRE(out, "NULL output");
    *out = nullptr;
    reload_env();                              // an object is created: the switches are read now (common.hpp)
    VEXHIP_REQUIRE(n >= 0 && format >= VEXHIP_SPMAT_AUTO && format <= VEXHIP_SPMAT_CSR, "bad argument");
    VEXHIP_REQUIRE(n == 0 || ptr, "NULL row pointers");
    spmat *A = new (std::nothrow) spmat;
    VEXHIP_REQUIRE(A, "out of host memory");
    A->dev = dev;
    if (int rc = build<V, P>(A, stream, n, ptr, col, val, format, flags)) { release(A); return rc; }
    if (env(ENV_VEXHIP_DEBUG)) {
        const product_choice pc = select_product(A, nullptr, nullptr);
        std::fprintf(stderr, "[vexhip] matrix of %lld rows, %lld entries: %s; product: %s (%s)\n", (long long)A->n, (long long)A->nnz, A->why, pc.kernel, pc.reason);
    }
    *out = reinterpret_cast<vexhip_spmat *>(A);
    return 0;
}

template <typename V>
int apply(const spmat *A, void *stream, V alpha, int append, const V *x, V *y)
{
    typedef api<V> F;
    VEXHIP_REQUIRE(A && A->value_type == F::type, "matrix and vector value types differ");
    const product_choice pc = select_product(A, x, y);
    const int32_t *cp = A->tail ? A->csr_ptr : nullptr;
    const void *tables = A->direct ? A->grid.table : A->pool;
    const int32_t *classes = A->direct ? A->grid.line_class : A->blocks;
    switch (pc.kind) {
        case P_NONE: return 0;
        case P_ZERO_FILL:
            if (!append) { VEXHIP_SET_DEVICE(A->dev); VEXHIP_TRY(hipMemsetAsync(y, 0, sizeof(V) * (size_t)A->n, as_stream(stream))); }
            return 0;
        case P_PLANE64:
            if constexpr (std::is_same<V, double>::value)
                return vexhip_spmv_sell8v_plane_f64_i32(A->dev, stream, A->n, alpha, append, A->ell_w, tables, classes, A->deltas, (const double *)A->values, x, y, &A->plane);
            break;
        case P_PLANE32:
            if constexpr (std::is_same<V, float>::value)
                return vexhip_spmv_sell8v_plane_f32_i32(A->dev, stream, A->n, alpha, append, A->ell_w, tables, classes, A->deltas, (const float *)A->values, x, y, &A->plane);
            break;
        case P_GRID64:
            if constexpr (std::is_same<V, double>::value) return vexhip_spmv_sell8v_grid_f64(A->dev, stream, A->n, alpha, append, (const double *)A->values, x, y, &A->grid);
            break;
        case P_GRID32:
            if constexpr (std::is_same<V, float>::value) return vexhip_spmv_sell8v_grid_f32(A->dev, stream, A->n, alpha, append, (const float *)A->values, x, y, &A->grid);
            break;
        case P_MARCH:
            if (A->runs && A->ell_max_col >= 0 && g_sell8_variant == 0)
                return sell8v_runs_apply(A->dev, stream, A->n, alpha, append, A->ell_w, A->pool, A->blocks, A->deltas, (const V *)A->values, cp, A->csr_col, (const V *)A->csr_val, x, y, &A->trav, A->runs, A->ell_max_col);
            return F::mul_vd(A->dev, stream, A->n, alpha, append, A->ell_w, A->pool, A->blocks, A->deltas, A->values, cp, A->csr_col, A->csr_val, x, y, &A->trav, &A->march);
        case P_PAIR_CODES: return F::mul_v(A->dev, stream, A->n, alpha, append, A->ell_w, A->sell, A->deltas, A->values, cp, A->csr_col, A->csr_val, x, y, &A->trav);
        case P_PAIR_DICT_VALUES: return F::mul_dd(A->dev, stream, A->n, alpha, append, A->ell_w, A->sell, A->pool, A->blocks, A->deltas, cp, A->csr_col, A->csr_val, x, y, &A->trav);
        case P_PAIR_VALUES: return F::mul_d(A->dev, stream, A->n, alpha, append, A->ell_w, A->sell, A->deltas, cp, A->csr_col, A->csr_val, x, y, &A->trav);
        case P_SELL32: return F::mul_s(A->dev, stream, A->n, alpha, append, A->ell_w, A->sell, cp, A->csr_col, A->csr_val, x, y, &A->trav);
        case P_CSR64: return spmv_csr_p64(A->dev, stream, A->n, alpha, append, A->csr_ptr64, A->csr_col, (const V *)A->csr_val, x, y, &A->trav);
        case P_CSR32: return F::mul_c(A->dev, stream, A->n, alpha, append, A->csr_ptr, A->csr_col, A->csr_val, x, y, &A->trav);
    }
    return fail(__FILE__, __LINE__, "the selected product does not exist for this value type");
}

template <typename V>
int apply_multi(const spmat *A, void *stream, int k, V alpha, int append, const V *const *x, V *const *y)
{
    typedef api<V> F;
    VEXHIP_REQUIRE(A && A->value_type == F::type, "matrix and vector value types differ");
    VEXHIP_REQUIRE(k >= 1 && x && y, "bad argument");
    if (A->n == 0) return 0;
    if (A->nnz == 0 || A->format == VEXHIP_SPMAT_CSR || A->direct) {          // no multi-vector kernel (stored by grid line: k plane / grid products beat it): one product per component
        for (int c = 0; c < k; ++c) if (int rc = apply<V>(A, stream, alpha, append, x[c], y[c])) return rc;
        return 0;
    }
    const int32_t *cp = A->tail ? A->csr_ptr : nullptr;
    switch (A->format) {
        case VEXHIP_SPMAT_SELL8V:
            if (A->blocks) return F::mm_vd(A->dev, stream, A->n, k, alpha, append, A->ell_w, A->pool, A->blocks, A->deltas, A->values, cp, A->csr_col, A->csr_val, x, y, &A->trav);
            return F::mm_v(A->dev, stream, A->n, k, alpha, append, A->ell_w, A->sell, A->deltas, A->values, cp, A->csr_col, A->csr_val, x, y, &A->trav);
        case VEXHIP_SPMAT_SELL8:
            if (A->blocks) return F::mm_dd(A->dev, stream, A->n, k, alpha, append, A->ell_w, A->sell, A->pool, A->blocks, A->deltas, cp, A->csr_col, A->csr_val, x, y, &A->trav);
            return F::mm_d(A->dev, stream, A->n, k, alpha, append, A->ell_w, A->sell, A->deltas, cp, A->csr_col, A->csr_val, x, y, &A->trav);
        default:                  return F::mm_s(A->dev, stream, A->n, k, alpha, append, A->ell_w, A->sell, cp, A->csr_col, A->csr_val, x, y, &A->trav);
    }
}

} // namespace

// The stored strip of one rank (its rows, plus an empty ghost plane in front of / behind them where it has a neighbour) as the
// operand of the one-launch product step (halo.hpp, comm.hip): must be stored for the plane product.  planes / lines_per_plane
// tell the caller the geometry it has to match (ghost plane = one plane of the stored grid).
int spmat_halo_geometry(const vexhip_spmat *h, int *planes, int *lines_per_plane, int *line_length, int *value_type) {
    const spmat *A = reinterpret_cast<const spmat *>(h);
    VEXHIP_REQUIRE(A && planes && lines_per_plane && line_length && value_type, "NULL argument");
    *planes = 0; *lines_per_plane = 0; *line_length = 0; *value_type = A->value_type;
    if (A->format != VEXHIP_SPMAT_SELL8V || A->tail) return 0;
    if ((A->blocks || A->direct) && A->plane.usable) { *planes = A->plane.planes; *lines_per_plane = A->plane.lines_per_plane; *line_length = 512; }
    else if (A->grid.usable && A->value_type == VEXHIP_F64) { *planes = A->grid.planes; *lines_per_plane = A->grid.lines_per_plane; *line_length = A->grid.nx; }      // lines of any length: the pull form, fp64
    return 0;
}
// ... or ANY matrix stored with diagonal codes (round 6; the pair product's role, sell8.hip): fp64, ELL width <= 8, no CSR tail, whole slices,
// every diagonal within one ghost range.  *reach = the largest |diagonal| (0: not such a matrix).
int spmat_halo_general(const vexhip_spmat *h, int64_t halo, int64_t rows_ext, int *reach, int *value_type) {
    const spmat *A = reinterpret_cast<const spmat *>(h);
    VEXHIP_REQUIRE(A && reach && value_type, "NULL argument");
    *reach = 0; *value_type = A->value_type;
    if (!(A->format == VEXHIP_SPMAT_SELL8 || A->format == VEXHIP_SPMAT_SELL8V) || A->tail || A->value_type != VEXHIP_F64 || A->ell_w > 8 || A->ndeltas < 1) return 0;
    if (A->n != rows_ext || halo % 512 || A->n % 512 || !A->deltas) return 0;
    if (A->format == VEXHIP_SPMAT_SELL8V ? !(A->sell || (A->blocks && A->pool)) : !A->sell) return 0;
    VEXHIP_SET_DEVICE(A->dev);
    int table[256];
    VEXHIP_TRY(hipMemcpy(table, A->deltas, sizeof(int) * (size_t)A->ndeltas, hipMemcpyDeviceToHost));
    long long far = 0;
    for (int k = 0; k < A->ndeltas; ++k) far = std::max<long long>(far, std::llabs((long long)table[k]));
    if (far == 0 || far > halo) return 0;
    *reach = (int)far;
    return 0;
}
int spmat_device(const vexhip_spmat *h, int *dev) {
    VEXHIP_REQUIRE(h && dev, "NULL argument");
    *dev = reinterpret_cast<const spmat *>(h)->dev;
    return 0;
}
int spmat_apply_halo(const vexhip_spmat *h, hipStream_t s, double alpha, int append, const void *x, void *y, const halo_dev &H) {
    const spmat *A = reinterpret_cast<const spmat *>(h);
    if (A && H.lo_planes > 0 && H.hi_planes == -1) {           // the pair product's role (comm.hip marks it: hi_planes = -1, lo_planes = reach)
        VEXHIP_REQUIRE(A->value_type == VEXHIP_F64 && !A->tail && H.pull, "the one-launch step of a matrix with diagonal codes: fp64, no CSR tail, shares read in place");
        halo_dev G = H; G.hi_planes = 0;
        const bool vcoded = A->format == VEXHIP_SPMAT_SELL8V;
        return sell8_apply_halo(A->dev, s, (long long)(H.z1 - H.z0) * H.halo, alpha, append, (int)A->ell_w, vcoded, A->sell, A->pool, A->blocks, A->deltas,
                                (const double *)A->values, static_cast<const double *>(x), static_cast<double *>(y), G);
    }
    VEXHIP_REQUIRE(A && A->format == VEXHIP_SPMAT_SELL8V && !A->tail, "the one-launch step needs a matrix stored for the plane or the grid product");
    if (A->value_type == VEXHIP_F32) {
        VEXHIP_REQUIRE((A->blocks || A->direct) && A->plane.usable && H.pull, "the one-launch step of a float matrix needs the plane product and shares read in place");
        return plane32_apply_halo(A->dev, s, A->n, (float)alpha, append, A->ell_w, A->direct ? A->grid.table : A->pool,
                                  A->direct ? A->grid.line_class : A->blocks, A->deltas, (const float *)A->values, static_cast<const float *>(x), static_cast<float *>(y), &A->plane, H);
    }
    if ((A->blocks || A->direct) && A->plane.usable)
        return plane_apply_halo(A->dev, s, A->n, alpha, append, A->ell_w, A->direct ? A->grid.table : A->pool,
                                A->direct ? A->grid.line_class : A->blocks, A->deltas, (const double *)A->values, static_cast<const double *>(x), static_cast<double *>(y), &A->plane, H);
    VEXHIP_REQUIRE(A->grid.usable && H.pull, "the one-launch step needs a matrix stored for the plane product (pushed shares) or the grid product (shares read in place)");
    return grid_apply_halo(A->dev, s, A->n, alpha, append, (const double *)A->values, static_cast<const double *>(x), static_cast<double *>(y), &A->grid, H);
}
} // namespace vexhip

using namespace vexhip;

namespace {
// products whose launcher hands the addend to its kernels' store_pair (traversal.hpp): the pair / any-width products of the coded storages
// and of the 32-bit columns, the CSR kernels and the march product in their own epilogues
inline bool addend_by_store_pair(const spmat *A, product_kind k) {
    return k == P_PAIR_CODES || k == P_PAIR_DICT_VALUES || k == P_PAIR_VALUES || k == P_SELL32 || k == P_MARCH || k == P_CSR32 || k == P_CSR64;
}
template <typename V>
int apply_with_addend(const spmat *A, void *stream, V alpha, const V *x, V beta, const V *z, V *y) {
    pending_addend &a = next_addend();
    a.z = z; a.beta = (double)beta; a.taken = false;
    const int rc = apply<V>(A, stream, alpha, 0, x, y);
    const bool taken = a.taken;
    a = pending_addend();
    if (rc) return rc;
    if (!taken) return fail(__FILE__, __LINE__, "y = alpha A x + beta z: the product that ran took no addend (select_product and the launchers disagree)");
    return 0;
}

} // namespace

extern "C" {

int vexhip_spmat_create_f64_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const double *val,
        int format, int flags, vexhip_spmat **out)
{ return create<double, int32_t>(dev, stream, n, ptr, col, val, format, flags, out); }

int vexhip_spmat_create_f32_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const float *val,
        int format, int flags, vexhip_spmat **out)
{ return create<float, int32_t>(dev, stream, n, ptr, col, val, format, flags, out); }

int vexhip_spmat_create_f64_p64(int dev, void *stream, int64_t n, const int64_t *ptr, const int32_t *col, const double *val,
        int format, int flags, vexhip_spmat **out)
{ return create<double, long long>(dev, stream, n, reinterpret_cast<const long long *>(ptr), col, val, format, flags, out); }

int vexhip_spmat_create_f32_p64(int dev, void *stream, int64_t n, const int64_t *ptr, const int32_t *col, const float *val,
        int format, int flags, vexhip_spmat **out)
{ return create<float, long long>(dev, stream, n, reinterpret_cast<const long long *>(ptr), col, val, format, flags, out); }

int vexhip_spmat_destroy(vexhip_spmat *A) { release(reinterpret_cast<spmat *>(A)); return 0; }

int vexhip_spmat_apply_f64(const vexhip_spmat *A, void *stream, double alpha, int append, const double *x, double *y)
{ return apply<double>(reinterpret_cast<const spmat *>(A), stream, alpha, append, x, y); }
int vexhip_spmat_apply_f32(const vexhip_spmat *A, void *stream, float alpha, int append, const float *x, float *y)
{ return apply<float>(reinterpret_cast<const spmat *>(A), stream, alpha, append, x, y); }

// y = alpha A x + beta z in ONE pass where the matrix's product can take the addend (the plane product: from z, or -- z == x -- from the
// registers that hold x anyway), else as y = beta z followed by y += alpha A x: the same two roundings and one addition per element either way.
int vexhip_spmat_apply_axpby_f64(const vexhip_spmat *h, void *stream, double alpha, const double *x, double beta, const double *z, double *y)
{
    const spmat *A = reinterpret_cast<const spmat *>(h);
    VEXHIP_REQUIRE(A && A->value_type == VEXHIP_F64, "matrix and vector value types differ");
    VEXHIP_REQUIRE(x && y && z, "NULL argument");
    VEXHIP_REQUIRE(static_cast<const void *>(x) != static_cast<const void *>(y), "y = alpha A x + beta z: x and y are the same vector");
    if (A->n == 0) return 0;
    const product_choice pc = select_product(A, x, y);
    if (pc.kind == P_PLANE64 && (z == x || (reinterpret_cast<uintptr_t>(z) & 15) == 0))
        return plane_apply_axpby(A->dev, stream, A->n, alpha, z == x ? 2 : 1, z, beta, A->ell_w, A->direct ? A->grid.table : A->pool,
                                 A->direct ? A->grid.line_class : A->blocks, A->deltas, (const double *)A->values, x, y, &A->plane);
    if (pc.kind == P_GRID64 && (reinterpret_cast<uintptr_t>(z) & 7) == 0)
        return grid_apply_axpby(A->dev, stream, A->n, alpha, z == x ? 2 : 1, z, beta, (const double *)A->values, x, y, &A->grid);
    if (addend_by_store_pair(A, pc.kind)) return apply_with_addend<double>(A, stream, alpha, x, beta, z, y);
    if (!(z == y && beta == 1.0)) {
        VEXHIP_SET_DEVICE(A->dev);
        const long long grid = (A->n + 255) / 256;
        VEXHIP_REQUIRE(grid < (1ll << 31), "vector too large for one launch");
        scale_into_kernel<double><<<(unsigned)grid, 256, 0, as_stream(stream)>>>(y, z, beta, (long long)A->n);
        VEXHIP_LAUNCH_CHECK();
    }
    return apply<double>(A, stream, alpha, 1, x, y);
}

int vexhip_spmat_apply_axpby_f32(const vexhip_spmat *h, void *stream, float alpha, const float *x, float beta, const float *z, float *y)
{
    const spmat *A = reinterpret_cast<const spmat *>(h);
    VEXHIP_REQUIRE(A && A->value_type == VEXHIP_F32, "matrix and vector value types differ");
    VEXHIP_REQUIRE(x && y && z, "NULL argument");
    VEXHIP_REQUIRE(static_cast<const void *>(x) != static_cast<const void *>(y), "y = alpha A x + beta z: x and y are the same vector");
    if (A->n == 0) return 0;
    const product_choice pc = select_product(A, x, y);
    if (pc.kind == P_PLANE32)
        return plane32_apply_axpby(A->dev, stream, A->n, alpha, z == x ? 2 : 1, z, beta, A->ell_w, A->direct ? A->grid.table : A->pool,
                                   A->direct ? A->grid.line_class : A->blocks, A->deltas, (const float *)A->values, x, y, &A->plane);
    if (pc.kind == P_GRID32)
        return grid32_apply_axpby(A->dev, stream, A->n, alpha, z == x ? 2 : 1, z, beta, (const float *)A->values, x, y, &A->grid);
    if (addend_by_store_pair(A, pc.kind)) return apply_with_addend<float>(A, stream, alpha, x, beta, z, y);
    if (!(z == y && beta == 1.0f)) {
        VEXHIP_SET_DEVICE(A->dev);
        const long long grid = (A->n + 255) / 256;
        VEXHIP_REQUIRE(grid < (1ll << 31), "vector too large for one launch");
        scale_into_kernel<float><<<(unsigned)grid, 256, 0, as_stream(stream)>>>(y, z, beta, (long long)A->n);
        VEXHIP_LAUNCH_CHECK();
    }
    return apply<float>(A, stream, alpha, 1, x, y);
}

// 1: vexhip_spmat_apply_axpby_f64 on these vectors runs as ONE pass (the product takes the addend); 0: as y = beta z, y += alpha A x
int vexhip_spmat_axpby_fused(const vexhip_spmat *h, const void *x, const void *z, const void *y)
{
    const spmat *A = reinterpret_cast<const spmat *>(h);
    if (!A || !x || !y || !z || x == y || A->n == 0) return 0;
    const product_kind k = select_product(A, x, y).kind;
    if (addend_by_store_pair(A, k)) return 1;
    if (A->value_type == VEXHIP_F32) return k == P_PLANE32 || k == P_GRID32 ? 1 : 0;
    return (k == P_PLANE64 && (z == x || (reinterpret_cast<uintptr_t>(z) & 15) == 0)) || (k == P_GRID64 && (reinterpret_cast<uintptr_t>(z) & 7) == 0) ? 1 : 0;
}

int vexhip_spmat_apply_multi_f64(const vexhip_spmat *A, void *stream, int nrhs, double alpha, int append, const double *const *x, double *const *y)
{ return apply_multi<double>(reinterpret_cast<const spmat *>(A), stream, nrhs, alpha, append, x, y); }
int vexhip_spmat_apply_multi_f32(const vexhip_spmat *A, void *stream, int nrhs, float alpha, int append, const float *const *x, float *const *y)
{ return apply_multi<float>(reinterpret_cast<const spmat *>(A), stream, nrhs, alpha, append, x, y); }

int vexhip_spmat_get_info(const vexhip_spmat *h, vexhip_spmat_info *o) {
    VEXHIP_REQUIRE(h && o, "NULL argument");
    const spmat *A = reinterpret_cast<const spmat *>(h);
    std::memset(o, 0, sizeof(*o));
    o->format = A->format; o->value_type = A->value_type; o->device = A->dev;
    o->rows = A->n; o->nnz = A->nnz; o->ell_width = A->ell_w; o->tail_nnz = A->tail;
    o->ndeltas = A->ndeltas; o->nvalues = A->nvalues;
    // value-coded storage with a dictionary: the pool takes the place of the per-slice buffer (make_inline reads `sell`)
    o->sell = A->sell ? A->sell : A->pool; o->sell_bytes = A->sell_bytes; o->deltas = A->deltas; o->values = A->values;
    o->csr_ptr = A->csr_ptr; o->csr_col = A->csr_col; o->csr_val = A->csr_val;
    o->traversal = A->trav;
    o->slice_blocks = A->blocks; o->code_pool = A->pool; o->dictionary_blocks = A->dict_blocks;
    o->march = A->march;
    o->plane = A->plane;
    o->grid = A->grid;
    // bytes one product moves through HBM at least: the stored matrix + x once + y once (+ y read for "+=" not counted)
    const int64_t vb = A->value_type == VEXHIP_F64 ? 8 : 4;
    int64_t m = A->sell_bytes;
    if (A->blocks) {                        // codes come from the pool: the per-slice code parts (if still stored) are not read
        const int64_t ns = (A->n + 511) / 512;
        m += A->dict_blocks * A->code_bytes + ns * 4 - (A->sell ? ns * A->code_bytes : 0);
    }
    if (A->format == VEXHIP_SPMAT_CSR) m = A->nnz * (4 + vb) + (A->n + 1) * (A->csr_ptr64 ? 8 : 4);
    else if (A->tail) m += A->tail * (4 + vb) + (A->n + 1) * 4;
    if (A->grid.usable)                     // the grid product reads the matrix by grid line: a class per line + the class tables
        m = (A->n / A->grid.nx) * 4 + (int64_t)A->grid.classes * 7 * A->grid.pitch;
    o->matrix_bytes = m;
    // what a product of this matrix launches and why (the table above; for vectors at 16-byte addresses), and why this storage
    const product_choice pc = select_product(A, nullptr, nullptr);
    std::snprintf(o->product, sizeof o->product, "%s", pc.kernel);
    std::snprintf(o->reason, sizeof o->reason, "%s; product: %s", A->why, pc.reason);
    return 0;
}

} // extern "C"
