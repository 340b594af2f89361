#ifndef VEXCL_SPMAT_HPP
#define VEXCL_SPMAT_HPP
// vex::SpMat<val_t, col_t, idx_t>: sparse matrix partitioned by rows across the
// GPUs of a context (reference: vexcl/spmat.hpp:56-185 class + apply, :291-378
// setup_exchange; per-device parts spmat/csr.inl:34-256, spmat/hybrid_ell.inl:34-401;
// inline form spmat/inline_spmv.hpp).
//
// Each device keeps a LOCAL part (columns it owns, renumbered c - col_begin) and
// a REMOTE part (ghost columns renumbered to their rank in the device's sorted
// ghost set).  Device storage is int32-indexed; the local part is hybrid ELL
// (the reference's GPU choice, spmat.hpp:98-103), the sparse remote part CSR.
// Kernels: libvexhip (vexhip_spmv_hell_* / vexhip_spmv_csr_*).
//
// Ghost exchange: the reference stages ghosts device -> host -> device in five
// finish()-fenced phases (spmat.hpp:125-183).  Here every owner packs, per
// consumer, exactly the values that consumer needs (gather kernel on the
// primary queue) and the consumer pulls them with one peer copy per
// (owner, consumer) pair on its secondary queue -- xGMI, no host hop -- while
// the local product runs; the remote product waits on an event.
#include <memory>
#include <set>
#include <unordered_map>
#include <vector>

#include "operations.hpp"
#include "vector.hpp"
#include "multivector.hpp"
#include "spmat/ccsr.hpp"

namespace vex {

template <typename val_t, typename col_t = size_t, typename idx_t = size_t>
class SpMat {
    public:
        typedef val_t value_type;
        typedef val_t scalar_type;

        SpMat() : nrows(0), ncols(0), nnz(0) {}

        /// Partitions the host CSR matrix across the queues (spmat.hpp:71-106).
        SpMat(const std::vector<backend::command_queue> &queue, size_t n, size_t m,
              const idx_t *row, const col_t *col, const val_t *val)
            : queue(queue), part(vex::partition(n, queue)), col_part(vex::partition(m, queue)),
              nrows(n), ncols(m), nnz(static_cast<size_t>(row[n])), mtx(queue.size())
        {
            static_assert(std::is_same<val_t, double>::value || std::is_same<val_t, float>::value,
                    "SpMat value type must be float or double");
            for (const auto &q : queue) squeue.push_back(backend::duplicate_queue(q));   // spmat.hpp:81-82

            std::vector<std::vector<col_t>> ghosts(queue.size());
            for (unsigned d = 0; d < queue.size(); ++d)
                mtx[d] = std::make_shared<device_part>(queue[d], row + part[d], row + part[d + 1], col, val,
                        col_part[d], col_part[d + 1], ghosts[d]);
            if (queue.size() > 1) setup_exchange(ghosts);
        }

        size_t rows() const { return nrows; }
        size_t cols() const { return ncols; }
        size_t nonzeros() const { return nnz; }

        /// y = alpha * A * x  or  y += alpha * A * x  (spmat.hpp:120-185).
        template <class T>
        void apply(const vex::vector<T> &x, vex::vector<T> &y, scalar_type alpha = 1, bool append = false) const {
            static_assert(std::is_same<T, val_t>::value, "vector and matrix value types differ");
            precondition(x.size() == ncols && y.size() == nrows, "SpMat::apply: incompatible sizes");
            const bool exchange = queue.size() > 1 && !pairs.empty();
            if (exchange) start_exchange(x);
            for (unsigned d = 0; d < queue.size(); ++d)
                if (part[d + 1] > part[d]) mtx[d]->mul_local(queue[d], x(d), y(d), alpha, append);
            if (exchange) finish_exchange(y, alpha);
        }

        /// Y = alpha * A * X for a multivector (spmat.hpp:388-398).  Without a ghost exchange
        /// (one device, or block-diagonal partitions) every device reads its matrix ONCE for
        /// up to four components (vexhip_spmm_*); otherwise component by component.
        template <class T, size_t N, class... Ts>
        void apply(const multivector<T, N> &x, detail::multi_target<Ts...> &y, scalar_type alpha = 1, bool append = false) const {
            static_assert(sizeof...(Ts) == N, "multivector product: component count mismatch");
            static_assert(std::is_same<T, val_t>::value, "vector and matrix value types differ");
            apply_components(x, y, alpha, append, std::make_index_sequence<N>());
        }
        /// One vector on the right, several on the left: the same product lands in every component.
        template <class T, class... Ts>
        void apply(const vex::vector<T> &x, detail::multi_target<Ts...> &y, scalar_type alpha = 1, bool append = false) const {
            detail::tuple_for_each(y.v, [&](auto &yk, size_t) { this->apply(x, yk, alpha, append); });
        }

        // ---- pieces used by make_inline (spmat.hpp:195-230) -------------------------
        struct device_part;
        const device_part &part_of(unsigned d) const { return *mtx[d]; }
        const std::vector<backend::command_queue> &queue_list() const { return queue; }
        const std::vector<size_t> &row_partition() const { return part; }

        // ELL part in SELL-512 storage (include/vexhip.h): slice-major, one slice = the 512
        // rows of one workgroup; element (r, j) of slice s at s*w*512 + j*512 + r.
        struct matrix_arrays {
            size_t n = 0, nnz = 0;
            long ell_w = 0;
            backend::device_vector<char> sell;                              // columns (or 1-byte diagonal codes) then values, per slice
            backend::device_vector<int> deltas; int ndeltas = -1;           // SELL8: sorted diagonal table (vexhip.h)
            backend::device_vector<val_t> values; int nvalues = -1;         // SELL8V: sorted value table, values coded too
            vexhip_traversal trav = {0, 0, 0, 0, nullptr};                  // traversal order (grid 0 = plain)
            backend::device_vector<int> csr_ptr, csr_col; backend::device_vector<val_t> csr_val;
            size_t csr_nnz = 0;
            bool empty() const { return nnz == 0; }
        };

        struct device_part {
            matrix_arrays loc, rem;
            size_t n;

            device_part(const backend::command_queue &q, const idx_t *row_begin, const idx_t *row_end,
                    const col_t *col, const val_t *val, size_t col_begin, size_t col_end, std::vector<col_t> &ghost_cols)
                : n(row_end - row_begin)
            {
                // split rows into local / remote CSR (csr.inl:92-131, hybrid_ell.inl:166-216)
                std::set<col_t> gset;
                for (auto r = row_begin; r != row_end; ++r)
                    for (idx_t j = r[0]; j < r[1]; ++j)
                        if (!(static_cast<size_t>(col[j]) >= col_begin && static_cast<size_t>(col[j]) < col_end)) gset.insert(col[j]);
                ghost_cols.assign(gset.begin(), gset.end());
                std::unordered_map<col_t, int> r2l(2 * ghost_cols.size() + 1);
                for (size_t g = 0; g < ghost_cols.size(); ++g) r2l[ghost_cols[g]] = static_cast<int>(g);

                std::vector<int> lptr(1, 0), lcol, rptr(1, 0), rcol;
                std::vector<val_t> lval, rval;
                lptr.reserve(n + 1); rptr.reserve(n + 1);
                for (auto r = row_begin; r != row_end; ++r) {
                    for (idx_t j = r[0]; j < r[1]; ++j) {
                        size_t c = static_cast<size_t>(col[j]);
                        if (c >= col_begin && c < col_end) { lcol.push_back(static_cast<int>(c - col_begin)); lval.push_back(val[j]); }
                        else { rcol.push_back(r2l[col[j]]); rval.push_back(val[j]); }
                    }
                    precondition(lcol.size() < (1ull << 31) && rcol.size() < (1ull << 31), "SpMat: more than 2^31 nonzeros on one device");
                    lptr.push_back(static_cast<int>(lcol.size()));
                    rptr.push_back(static_cast<int>(rcol.size()));
                }
                precondition(col_end - col_begin < (1ull << 31), "SpMat: more than 2^31 columns on one device");
                upload(q, loc, lptr, lcol, lval, use_ell());
                // remote part: only the rows that reach a ghost column are stored (row list + compact CSR)
                rem.n = n; rem.nnz = rcol.size();
                if (rem.nnz) {
                    std::vector<int> rows, cptr(1, 0);
                    for (size_t i = 0; i < n; ++i)
                        if (rptr[i + 1] > rptr[i]) { rows.push_back(static_cast<int>(i)); cptr.push_back(rptr[i + 1]); }
                    rem_rows = backend::device_vector<int>(q, rows.size(), rows.data());
                    rem.csr_ptr = backend::device_vector<int>(q, cptr.size(), cptr.data());
                    rem.csr_col = backend::device_vector<int>(q, rcol.size(), rcol.data());
                    rem.csr_val = backend::device_vector<val_t>(q, rval.size(), rval.data());
                    rem.csr_nnz = rem.nnz;
                }
            }
            backend::device_vector<int> rem_rows;

            static bool use_ell() {
#ifdef VEXCL_SPMAT_CSR
                return false;
#else
                return true;
#endif
            }

            void upload(const backend::command_queue &q, matrix_arrays &A, const std::vector<int> &ptr,
                    const std::vector<int> &col, const std::vector<val_t> &val, bool ell)
            {
                A.n = n; A.nnz = col.size();
                if (!A.nnz || !n) return;
                backend::device_vector<int> dptr(q, ptr.size(), ptr.data());
                backend::device_vector<int> dcol(q, col.size(), col.data());
                backend::device_vector<val_t> dval(q, val.size(), val.data());
                if (!ell) { A.csr_ptr = dptr; A.csr_col = dcol; A.csr_val = dval; A.csr_nnz = A.nnz; return; }
                int dev = q.device_ordinal();
                int64_t w = 0, tail = 0;
                backend::check(vexhip_hell_analyze_i32(dev, q.raw(), (int64_t)n, dptr.raw(), &w, &tail));
                if (w == 0) { A.csr_ptr = dptr; A.csr_col = dcol; A.csr_val = dval; A.csr_nnz = A.nnz; return; }
                A.ell_w = (long)w; A.csr_nnz = (size_t)tail;
                if (tail) {     // rows wider than the ELL width keep their tail in CSR (hybrid_ell.inl:166-198)
                    A.csr_ptr = backend::device_vector<int>(q, n + 1); A.csr_col = backend::device_vector<int>(q, tail); A.csr_val = backend::device_vector<val_t>(q, tail);
                    backend::check(fill(dev, q.raw(), (int64_t)n, dptr.raw(), dcol.raw(), dval.raw(), w, (int64_t)alignup(n, 16),
                                nullptr, nullptr, A.csr_ptr.raw(), A.csr_col.raw(), A.csr_val.raw()));
                }
                // banded / stencil matrices (<= 255 distinct diagonals): 1-byte diagonal codes instead of 32-bit columns
                backend::device_vector<int> deltas(q, 256);
                int nd = -1;
                backend::check(vexhip_sell8_analyze_i32(dev, q.raw(), (int64_t)n, dptr.raw(), dcol.raw(), w, deltas.raw(), &nd));
                if (nd > 0) {
                    A.deltas = deltas; A.ndeltas = nd;
                    // ... and at most 255 distinct values (constant-coefficient stencils): 1-byte value codes too
                    backend::device_vector<val_t> values(q, 256);
                    int nv = -1;
                    backend::check(sell8v_analyze(dev, q.raw(), (int64_t)n, dptr.raw(), dval.raw(), w, values.raw(), &nv));
                    if (nv > 0) {
                        A.values = values; A.nvalues = nv;
                        A.sell = backend::device_vector<char>(q, (size_t)vexhip_sell8v_bytes((int64_t)n, w));
                        backend::check(sell8v_fill(dev, q.raw(), (int64_t)n, dptr.raw(), dcol.raw(), dval.raw(), w, deltas.raw(), nd,
                                    values.raw(), nv, A.sell.raw(), &A.trav));
                    } else {
                        A.sell = backend::device_vector<char>(q, (size_t)vexhip_sell8_bytes((int64_t)n, w, (int)sizeof(val_t)));
                        backend::check(sell8_fill(dev, q.raw(), (int64_t)n, dptr.raw(), dcol.raw(), dval.raw(), w, deltas.raw(), nd, A.sell.raw(), &A.trav));
                    }
                } else {
                    A.sell = backend::device_vector<char>(q, (size_t)vexhip_sell_bytes((int64_t)n, w, (int)sizeof(val_t)));
                    backend::check(sell_fill(dev, q.raw(), (int64_t)n, dptr.raw(), dcol.raw(), dval.raw(), w, A.sell.raw()));
                    backend::check(vexhip_sell_order_i32(dev, q.raw(), (int64_t)n, w, (int)sizeof(val_t), A.sell.raw(), 0, nullptr, 0, &A.trav));
                }
                q.finish();
            }

            static int sell8v_analyze(int dev, void *s, int64_t n, const int *p, const double *v, int64_t w, double *vals, int *nv) { return vexhip_sell8v_analyze_f64_i32(dev, s, n, p, v, w, vals, nv); }
            static int sell8v_analyze(int dev, void *s, int64_t n, const int *p, const float *v, int64_t w, float *vals, int *nv) { return vexhip_sell8v_analyze_f32_i32(dev, s, n, p, v, w, vals, nv); }
            static int sell8v_fill(int dev, void *s, int64_t n, const int *p, const int *c, const double *v, int64_t w, const int *d, int nd, const double *vals, int nv, void *sl, vexhip_traversal *t) { return vexhip_sell8v_fill_f64_i32(dev, s, n, p, c, v, w, d, nd, vals, nv, sl, t); }
            static int sell8v_fill(int dev, void *s, int64_t n, const int *p, const int *c, const float *v, int64_t w, const int *d, int nd, const float *vals, int nv, void *sl, vexhip_traversal *t) { return vexhip_sell8v_fill_f32_i32(dev, s, n, p, c, v, w, d, nd, vals, nv, sl, t); }
            static int sell8_fill(int dev, void *s, int64_t n, const int *p, const int *c, const double *v, int64_t w, const int *d, int nd, void *sl, vexhip_traversal *t) { return vexhip_sell8_fill_f64_i32(dev, s, n, p, c, v, w, d, nd, sl, t); }
            static int sell8_fill(int dev, void *s, int64_t n, const int *p, const int *c, const float *v, int64_t w, const int *d, int nd, void *sl, vexhip_traversal *t) { return vexhip_sell8_fill_f32_i32(dev, s, n, p, c, v, w, d, nd, sl, t); }
            static int sell_fill(int dev, void *s, int64_t n, const int *p, const int *c, const double *v, int64_t w, void *sl) { return vexhip_sell_fill_f64_i32(dev, s, n, p, c, v, w, sl); }
            static int sell_fill(int dev, void *s, int64_t n, const int *p, const int *c, const float *v, int64_t w, void *sl) { return vexhip_sell_fill_f32_i32(dev, s, n, p, c, v, w, sl); }

            static int fill(int dev, void *s, int64_t n, const int *p, const int *c, const double *v, int64_t w, int64_t pitch,
                    int *ec, double *ev, int *cp, int *cc, double *cv) { return vexhip_hell_fill_f64_i32(dev, s, n, p, c, v, w, pitch, ec, ev, cp, cc, cv); }
            static int fill(int dev, void *s, int64_t n, const int *p, const int *c, const float *v, int64_t w, int64_t pitch,
                    int *ec, float *ev, int *cp, int *cc, float *cv) { return vexhip_hell_fill_f32_i32(dev, s, n, p, c, v, w, pitch, ec, ev, cp, cc, cv); }

            static int spmv(int dev, void *s, int64_t n, double a, int app, const matrix_arrays &A, const double *x, double *y) {
                if (A.ell_w == 0 && A.csr_nnz)      // plain CSR storage: LDS-staged CSR kernel
                    return vexhip_spmv_csr_f64_i32(dev, s, n, a, app, A.csr_ptr.raw(), A.csr_col.raw(), A.csr_val.raw(), x, y);
                if (A.nvalues > 0)
                    return vexhip_spmv_sell8v_f64_i32(dev, s, n, a, app, A.ell_w, A.sell.raw(), A.deltas.raw(), A.values.raw(),
                            A.csr_nnz ? A.csr_ptr.raw() : nullptr, A.csr_col.raw(), A.csr_val.raw(), x, y, &A.trav);
                if (A.ndeltas > 0)
                    return vexhip_spmv_sell8_f64_i32(dev, s, n, a, app, A.ell_w, A.sell.raw(), A.deltas.raw(),
                            A.csr_nnz ? A.csr_ptr.raw() : nullptr, A.csr_col.raw(), A.csr_val.raw(), x, y, &A.trav);
                return vexhip_spmv_sell_f64_i32(dev, s, n, a, app, A.ell_w, A.sell.raw(),
                        A.csr_nnz ? A.csr_ptr.raw() : nullptr, A.csr_col.raw(), A.csr_val.raw(), x, y, &A.trav);
            }
            static int spmv(int dev, void *s, int64_t n, float a, int app, const matrix_arrays &A, const float *x, float *y) {
                if (A.ell_w == 0 && A.csr_nnz)
                    return vexhip_spmv_csr_f32_i32(dev, s, n, a, app, A.csr_ptr.raw(), A.csr_col.raw(), A.csr_val.raw(), x, y);
                if (A.nvalues > 0)
                    return vexhip_spmv_sell8v_f32_i32(dev, s, n, a, app, A.ell_w, A.sell.raw(), A.deltas.raw(), A.values.raw(),
                            A.csr_nnz ? A.csr_ptr.raw() : nullptr, A.csr_col.raw(), A.csr_val.raw(), x, y, &A.trav);
                if (A.ndeltas > 0)
                    return vexhip_spmv_sell8_f32_i32(dev, s, n, a, app, A.ell_w, A.sell.raw(), A.deltas.raw(),
                            A.csr_nnz ? A.csr_ptr.raw() : nullptr, A.csr_col.raw(), A.csr_val.raw(), x, y, &A.trav);
                return vexhip_spmv_sell_f32_i32(dev, s, n, a, app, A.ell_w, A.sell.raw(),
                        A.csr_nnz ? A.csr_ptr.raw() : nullptr, A.csr_col.raw(), A.csr_val.raw(), x, y, &A.trav);
            }

            static int spmm(int dev, void *s, int64_t n, int k, double a, int app, const matrix_arrays &A, const double *const *x, double *const *y) {
                if (A.nvalues > 0)
                    return vexhip_spmm_sell8v_f64_i32(dev, s, n, k, a, app, A.ell_w, A.sell.raw(), A.deltas.raw(), A.values.raw(),
                            A.csr_nnz ? A.csr_ptr.raw() : nullptr, A.csr_col.raw(), A.csr_val.raw(), x, y, &A.trav);
                if (A.ndeltas > 0)
                    return vexhip_spmm_sell8_f64_i32(dev, s, n, k, a, app, A.ell_w, A.sell.raw(), A.deltas.raw(),
                            A.csr_nnz ? A.csr_ptr.raw() : nullptr, A.csr_col.raw(), A.csr_val.raw(), x, y, &A.trav);
                return vexhip_spmm_sell_f64_i32(dev, s, n, k, a, app, A.ell_w, A.sell.raw(),
                        A.csr_nnz ? A.csr_ptr.raw() : nullptr, A.csr_col.raw(), A.csr_val.raw(), x, y, &A.trav);
            }
            static int spmm(int dev, void *s, int64_t n, int k, float a, int app, const matrix_arrays &A, const float *const *x, float *const *y) {
                if (A.nvalues > 0)
                    return vexhip_spmm_sell8v_f32_i32(dev, s, n, k, a, app, A.ell_w, A.sell.raw(), A.deltas.raw(), A.values.raw(),
                            A.csr_nnz ? A.csr_ptr.raw() : nullptr, A.csr_col.raw(), A.csr_val.raw(), x, y, &A.trav);
                if (A.ndeltas > 0)
                    return vexhip_spmm_sell8_f32_i32(dev, s, n, k, a, app, A.ell_w, A.sell.raw(), A.deltas.raw(),
                            A.csr_nnz ? A.csr_ptr.raw() : nullptr, A.csr_col.raw(), A.csr_val.raw(), x, y, &A.trav);
                return vexhip_spmm_sell_f32_i32(dev, s, n, k, a, app, A.ell_w, A.sell.raw(),
                        A.csr_nnz ? A.csr_ptr.raw() : nullptr, A.csr_col.raw(), A.csr_val.raw(), x, y, &A.trav);
            }

            /// Local part times several vectors at once (pointers of this device's segments).
            void mul_local_multi(const backend::command_queue &q, int k, const val_t *const *x, val_t *const *y,
                    val_t alpha, bool append) const
            {
                const int dev = q.device_ordinal();
                if (loc.empty()) {
                    if (!append) for (int c = 0; c < k; ++c) backend::check(vexhip_memset(dev, y[c], 0, n * sizeof(val_t), q.raw()));
                    return;
                }
                if (loc.ell_w == 0) {           // CSR-only storage: no multi-vector kernel, one product per component
                    for (int c = 0; c < k; ++c) backend::check(spmv(dev, q.raw(), (int64_t)n, alpha, append ? 1 : 0, loc, x[c], y[c]));
                    return;
                }
                backend::check(spmm(dev, q.raw(), (int64_t)n, k, alpha, append ? 1 : 0, loc, x, y));
            }

            static int spmv_rows(int dev, void *s, int64_t nr, double a, const int *rows, const int *p, const int *c, const double *v, const double *x, double *y) {
                return vexhip_spmv_csr_rows_f64_i32(dev, s, nr, a, rows, p, c, v, x, y); }
            static int spmv_rows(int dev, void *s, int64_t nr, float a, const int *rows, const int *p, const int *c, const float *v, const float *x, float *y) {
                return vexhip_spmv_csr_rows_f32_i32(dev, s, nr, a, rows, p, c, v, x, y); }

            /// csr.inl:186-200: an empty local part zero-fills y on SET.
            void mul_local(const backend::command_queue &q, const backend::device_vector<val_t> &x,
                    backend::device_vector<val_t> &y, val_t alpha, bool append) const
            {
                if (loc.empty()) {
                    if (!append) backend::check(vexhip_memset(q.device_ordinal(), y.raw(), 0, n * sizeof(val_t), q.raw()));
                    return;
                }
                backend::check(spmv(q.device_ordinal(), q.raw(), (int64_t)n, alpha, append ? 1 : 0, loc, x.raw(), y.raw()));
            }
            void mul_remote(const backend::command_queue &q, const backend::device_vector<val_t> &ghosts,
                    backend::device_vector<val_t> &y, val_t alpha) const
            {
                if (rem.empty()) return;
                backend::check(spmv_rows(q.device_ordinal(), q.raw(), (int64_t)rem_rows.size(), alpha, rem_rows.raw(),
                            rem.csr_ptr.raw(), rem.csr_col.raw(), rem.csr_val.raw(), ghosts.raw(), y.raw()));
            }
        };

    private:
        std::vector<backend::command_queue> queue, squeue;
        std::vector<size_t> part, col_part;
        size_t nrows, ncols, nnz;
        std::vector<std::shared_ptr<device_part>> mtx;

        template <class T, size_t N, class... Ts, size_t... I>
        void apply_components(const multivector<T, N> &x, detail::multi_target<Ts...> &y, scalar_type alpha, bool append,
                std::index_sequence<I...>) const
        {
            const bool exchange = queue.size() > 1 && !pairs.empty();
            if (exchange) {       // the ghost buffers hold one vector: component by component
                int dummy[] = {0, (apply(x(I), std::get<I>(y.v), alpha, append), 0)...};
                (void)dummy;
                return;
            }
            precondition(x.size() == ncols && std::get<0>(y.v).size() == nrows, "SpMat::apply: incompatible sizes");
            for (unsigned d = 0; d < queue.size(); ++d) {
                if (part[d + 1] == part[d]) continue;
                const val_t *xs[] = {x(I)(d).raw()...};
                val_t *ys[] = {std::get<I>(y.v)(d).raw()...};
                mtx[d]->mul_local_multi(queue[d], (int)N, xs, ys, alpha, append);
            }
        }

        // one (owner -> consumer) transfer
        struct pair_t { unsigned owner, consumer; size_t send_off, recv_off, count; };
        std::vector<pair_t> pairs;
        struct exchange_t {
            backend::device_vector<int> send_idx;     // local ids of everything this device sends, grouped by consumer
            backend::device_vector<val_t> send_buf;   // packed values, same order
            backend::device_vector<val_t> ghost_buf;  // what this device receives, ordered by global column
            size_t nsend = 0, nghost = 0;
        };
        mutable std::vector<exchange_t> exc;
        mutable std::vector<backend::event> copies_done;   // per consumer: previous product's peer copies

        static int gather(int dev, void *s, int64_t n, const int *idx, const double *src, double *dst) { return vexhip_gather_f64_i32(dev, s, n, idx, src, dst); }
        static int gather(int dev, void *s, int64_t n, const int *idx, const float *src, float *dst) { return vexhip_gather_f32_i32(dev, s, n, idx, src, dst); }

        /// spmat.hpp:291-378, point-to-point: for every consumer, its sorted ghost
        /// list splits into one contiguous run per owner (owners hold contiguous
        /// column ranges), so owner o packs run (o -> d) and d receives it in place.
        void setup_exchange(const std::vector<std::vector<col_t>> &ghosts) {
            const unsigned nd = static_cast<unsigned>(queue.size());
            exc.resize(nd);
            std::vector<std::vector<int>> send_idx(nd);
            for (unsigned d = 0; d < nd; ++d) {
                const auto &g = ghosts[d];
                exc[d].nghost = g.size();
                size_t i = 0;
                while (i < g.size()) {
                    unsigned o = static_cast<unsigned>(column_owner(static_cast<size_t>(g[i]), col_part));
                    size_t j = i;
                    while (j < g.size() && static_cast<size_t>(g[j]) < col_part[o + 1]) ++j;
                    pair_t p; p.owner = o; p.consumer = d; p.send_off = send_idx[o].size(); p.recv_off = i; p.count = j - i;
                    for (size_t k = i; k < j; ++k) send_idx[o].push_back(static_cast<int>(static_cast<size_t>(g[k]) - col_part[o]));
                    pairs.push_back(p);
                    i = j;
                }
            }
            for (unsigned d = 0; d < nd; ++d) {
                exc[d].nsend = send_idx[d].size();
                if (exc[d].nsend) {
                    exc[d].send_idx = backend::device_vector<int>(queue[d], send_idx[d].size(), send_idx[d].data());
                    exc[d].send_buf = backend::device_vector<val_t>(queue[d], send_idx[d].size());
                }
                if (exc[d].nghost) exc[d].ghost_buf = backend::device_vector<val_t>(queue[d], exc[d].nghost);
            }
        }

        template <class T>
        void start_exchange(const vex::vector<T> &x) const {
            const unsigned nd = static_cast<unsigned>(queue.size());
            // the previous product's copies must have drained the send buffers
            for (unsigned o = 0; o < nd; ++o)
                if (exc[o].nsend && !copies_done.empty()) backend::enqueue_barrier(queue[o], copies_done);
            std::vector<backend::event> packed(nd);
            for (unsigned o = 0; o < nd; ++o) {
                if (!exc[o].nsend) continue;
                backend::check(gather(queue[o].device_ordinal(), queue[o].raw(), (int64_t)exc[o].nsend,
                            exc[o].send_idx.raw(), x(o).raw(), exc[o].send_buf.raw()));
                packed[o] = backend::enqueue_marker(queue[o]);
            }
            copies_done.assign(nd, backend::event());
            for (const auto &p : pairs) {
                const backend::command_queue &sq = squeue[p.consumer];
                backend::enqueue_barrier(sq, backend::wait_list(1, packed[p.owner]));
                backend::check(vexhip_memcpy_peer(sq.device_ordinal(), exc[p.consumer].ghost_buf.raw() + p.recv_off,
                            queue[p.owner].device_ordinal(), exc[p.owner].send_buf.raw() + p.send_off,
                            p.count * sizeof(val_t), sq.raw()));
            }
            for (unsigned d = 0; d < nd; ++d) if (exc[d].nghost) copies_done[d] = backend::enqueue_marker(squeue[d]);
        }

        template <class T>
        void finish_exchange(vex::vector<T> &y, val_t alpha) const {
            for (unsigned d = 0; d < queue.size(); ++d) {
                if (!exc[d].nghost || part[d + 1] == part[d]) continue;
                backend::enqueue_barrier(queue[d], backend::wait_list(1, copies_done[d]));
                mtx[d]->mul_remote(queue[d], exc[d].ghost_buf, y(d), alpha);
            }
        }
};

/// A * x: a term that can only be assigned, added, subtracted or scaled
/// (spmat.hpp:381-386).
template <typename val_t, typename col_t, typename idx_t, typename T>
detail::additive_operator<SpMat<val_t, col_t, idx_t>, vector<T>>
operator*(const SpMat<val_t, col_t, idx_t> &A, const vector<T> &x) {
    return detail::additive_operator<SpMat<val_t, col_t, idx_t>, vector<T>>(A, x);
}

/// A * X with X a multivector: the product is applied to every component
/// (spmat.hpp:388-400; tests/spmv.cpp:262-300).
template <typename val_t, typename col_t, typename idx_t, typename T, size_t N>
detail::additive_operator<SpMat<val_t, col_t, idx_t>, multivector<T, N>>
operator*(const SpMat<val_t, col_t, idx_t> &A, const multivector<T, N> &x) {
    return detail::additive_operator<SpMat<val_t, col_t, idx_t>, multivector<T, N>>(A, x);
}

// ---- make_inline (spmat/inline_spmv.hpp:70-198; device function body
//      hybrid_ell.inl:322-351) ----------------------------------------------------------
namespace detail {
template <class M, class T>
struct inline_spmv : expression_base {
    typedef T value_type;
    const M &A; const vector<T> &x;
    inline_spmv(const M &A, const vector<T> &x) : A(A), x(x) {
        precondition(x.nparts() == 1, "make_inline is only supported for single-device contexts");
    }
    void preamble(gen_context &c) const {
        std::string name = c.next();
        const std::string V = type_name<T>();
        c.src.begin_function(V, name + "_hell_spmv");
        c.src.begin_function_parameters();
        c.src.parameter("long", "ell_w");
        c.src.parameter("const char *", "sell"); c.src.parameter("const int *", "deltas");
        c.src.parameter("const " + V + " *", "values");
        c.src.parameter("const int *", "csr_row"); c.src.parameter("const int *", "csr_col");
        c.src.parameter("const " + V + " *", "csr_val"); c.src.parameter("const " + V + " *", "in");
        c.src.parameter("ulong", "i");
        c.src.end_function_parameters();
        c.src.new_line() << V << " sum = 0;";
        c.src.new_line() << "if (values)";            // SELL8V: diagonal codes and value codes (include/vexhip.h)
        c.src.open("{");
        c.src.new_line() << "const long wp = (ell_w + 1) / 2;";
        c.src.new_line() << "const uint *cw = (const uint *)(sell + (i >> 9) * (wp * 2048)) + ((i & 511) >> 1);";
        c.src.new_line() << "const uint *vw = cw + wp * 256;";
        c.src.new_line() << "for(long j = 0; j < ell_w; ++j)";
        c.src.open("{");
        c.src.new_line() << "const int sh = 8 * ((j & 1) * 2 + (i & 1));";
        c.src.new_line() << "const uint code = (cw[(j >> 1) * 256] >> sh) & 255u;";
        c.src.new_line() << "if (code < 254u) sum += values[(vw[(j >> 1) * 256] >> sh) & 255u] * in[(long)i + deltas[code]];";
        c.src.close("}");
        c.src.close("}");
        c.src.new_line() << "else if (deltas)";       // SELL8: 1-byte diagonal codes
        c.src.open("{");
        c.src.new_line() << "const long wp = (ell_w + 1) / 2;";
        c.src.new_line() << "const char *slice = sell + (i >> 9) * (wp * 1024 + ell_w * 512 * sizeof(" << V << "));";
        c.src.new_line() << "const uint *cw = (const uint *)slice + ((i & 511) >> 1);";
        c.src.new_line() << "const " << V << " *ell_val = (const " << V << " *)(slice + wp * 1024) + (i & 511);";
        c.src.new_line() << "for(long j = 0; j < ell_w; ++j)";
        c.src.open("{");
        c.src.new_line() << "const uint code = (cw[(j >> 1) * 256] >> (8 * ((j & 1) * 2 + (i & 1)))) & 255u;";
        c.src.new_line() << "if (code < 254u) sum += ell_val[j * 512] * in[(long)i + deltas[code]];";
        c.src.close("}");
        c.src.close("}");
        c.src.new_line() << "else";                   // SELL-512 with 32-bit columns
        c.src.open("{");
        c.src.new_line() << "const char *slice = sell + (i >> 9) * (ell_w * 512 * (4 + sizeof(" << V << ")));";
        c.src.new_line() << "const int *ell_col = (const int *)slice + (i & 511);";
        c.src.new_line() << "const " << V << " *ell_val = (const " << V << " *)(slice + ell_w * 2048) + (i & 511);";
        c.src.new_line() << "for(long j = 0; j < ell_w; ++j)";
        c.src.open("{");
        c.src.new_line() << "int c = ell_col[j * 512];";
        c.src.new_line() << "if (c >= 0) sum += ell_val[j * 512] * in[c];";
        c.src.close("}");
        c.src.close("}");
        c.src.new_line() << "if (csr_row)";
        c.src.open("{");
        c.src.new_line() << "for(int j = csr_row[i], e = csr_row[i + 1]; j < e; ++j) sum += csr_val[j] * in[csr_col[j]];";
        c.src.close("}");
        c.src.new_line() << "return sum;";
        c.src.end_function();
    }
    void params(gen_context &c) const {
        std::string name = c.next();
        const std::string V = type_name<T>();
        c.src.parameter("long", name + "_ell_w");
        c.src.parameter("const char *", name + "_sell"); c.src.parameter("const int *", name + "_deltas");
        c.src.parameter("const " + V + " *", name + "_values");
        c.src.parameter("const int *", name + "_csr_row"); c.src.parameter("const int *", name + "_csr_col");
        c.src.parameter("const " + V + " *", name + "_csr_val"); c.src.parameter("const " + V + " *", name + "_vec");
    }
    void local_init(gen_context &c) const { c.next(); }
    void emit(gen_context &c) const {
        std::string n = c.next();
        c.src << n << "_hell_spmv(" << n << "_ell_w, " << n << "_sell, " << n << "_deltas, " << n << "_values, "
              << n << "_csr_row, " << n << "_csr_col, " << n << "_csr_val, " << n << "_vec, idx)";
    }
    void set_args(arg_context &a) const {
        a.next();
        const auto &L = A.part_of(a.device).loc;
        a.krn.push_arg((long)L.ell_w);
        a.krn.push_arg(static_cast<const char *>(L.sell.raw()));
        a.krn.push_arg(static_cast<const int *>(L.ndeltas > 0 ? L.deltas.raw() : nullptr));
        a.krn.push_arg(static_cast<const T *>(L.nvalues > 0 ? L.values.raw() : nullptr));
        a.krn.push_arg(static_cast<const int *>(L.csr_nnz ? L.csr_ptr.raw() : nullptr));
        a.krn.push_arg(static_cast<const int *>(L.csr_col.raw())); a.krn.push_arg(static_cast<const T *>(L.csr_val.raw()));
        a.krn.push_arg(static_cast<const T *>(x(a.device).raw()));
    }
    void get_props(prop_context &p) const {
        if (p.empty()) { p.queue = A.queue_list(); p.part = A.row_partition(); p.size = A.rows(); }
    }
};
} // namespace detail

/// sin(make_inline(A * x)): the product evaluated inside the fused kernel
/// (single device; inline_spmv.hpp:70-76).
template <class M, class T>
detail::inline_spmv<M, T> make_inline(const detail::additive_operator<M, vector<T>> &op) {
    return detail::inline_spmv<M, T>(op.A, op.x);
}

namespace detail {
/// make_inline(A * X), X a multivector: component I is make_inline(A * X(I)).
template <class M, class T, size_t N>
struct mv_inline_spmv : expression_base {
    typedef T value_type;
    const M &A; const multivector<T, N> &x;
    mv_inline_spmv(const M &A, const multivector<T, N> &x) : A(A), x(x) {}
    void get_props(prop_context &p) const {
        if (p.empty()) { p.queue = A.queue_list(); p.part = A.row_partition(); p.size = A.rows(); }
    }
};
template <class M, class T, size_t N> struct mv_dim<mv_inline_spmv<M, T, N>> : std::integral_constant<size_t, N> {};
template <size_t I, class M, class T, size_t N> struct component_of<I, mv_inline_spmv<M, T, N>, void> {
    typedef inline_spmv<M, T> type;
    static type get(const mv_inline_spmv<M, T, N> &p) { return type(p.A, p.x(I)); }
};
}
template <class M, class T, size_t N>
detail::mv_inline_spmv<M, T, N> make_inline(const detail::additive_operator<M, multivector<T, N>> &op) {
    return detail::mv_inline_spmv<M, T, N>(op.A, op.x);
}

} // namespace vex
#endif
