#ifndef VEXCL_SPMAT_HPP
#define VEXCL_SPMAT_HPP
// vex::SpMat<val_t, col_t, idx_t>: sparse matrix partitioned by rows across the
// GPUs of a context (reference: vexcl/spmat.hpp:56-185 class + apply, :291-378
// setup_exchange; per-device parts spmat/csr.inl:34-256, spmat/hybrid_ell.inl:34-401;
// inline form spmat/inline_spmv.hpp).
//
// Each device keeps a LOCAL part (columns it owns, renumbered c - col_begin): a vexhip_spmat, the library
// object that picks the storage (hybrid-ELL width as spmat/hybrid_ell.inl:103-110, then 1-byte diagonal /
// value codes, 32-bit columns or plain CSR; include/vexhip.h) -- and a REMOTE part (ghost columns renumbered
// to their rank in the device's sorted ghost set), a row-subset CSR because it is very sparse.
//
// Ghost exchange: the reference stages ghosts device -> host -> device in five finish()-fenced phases
// (spmat.hpp:125-183).  Here (vexcl/exchange.hpp) every owner packs, per consumer, exactly the values that
// consumer needs (gather kernel on the primary queue) and ONE vexhip_halo_exchange ships them on the
// secondary queues -- grouped ncclSend / ncclRecv over xGMI between GPUs, event-ordered copies between
// logical devices of one GPU -- while the local product runs; the remote product waits on an event.
#include <array>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "operations.hpp"
#include "exchange.hpp"
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

            std::vector<std::vector<col_t>> ghosts(queue.size());
            // every device's strip is uploaded, split and converted by its own host thread (round 4: the set-up of D devices
            // used to run one after the other)
            per_device([&](unsigned d) {
                mtx[d] = std::make_shared<device_part>(queue[d], row + part[d], row + part[d + 1], col, val,
                        col_part[d], col_part[d + 1], ghosts[d], queue.size() == 1);
            });
            if (queue.size() > 1) { exc.setup(queue, col_part, ghosts); setup_halo(ghosts); }
            for (auto &m : mtx) m->release_strip();
        }

        /// From per-device DEVICE strips (round 4): strip d holds rows part[d] .. part[d+1] of the matrix (part =
        /// vex::partition(n, queue), the partition of a vex::vector on this context), resident on device d, with strip-local
        /// row pointers (row[d][0] == 0), GLOBAL column ids and nonzeros[d] entries.  Nothing passes through the host: every
        /// device splits its strip into the local and the remote part and converts the local part itself
        /// (vexhip_csr_split_*, vexhip_spmat_create), all devices at the same time.  The reference builds from one host CSR
        /// (spmat.hpp:71-106: 12 GB of host arrays at 512^3); this is the constructor the multi-device headline runs through
        /// (examples/spmv_headline --devices).  Each strip must stay below 2^31 entries.
        SpMat(const std::vector<backend::command_queue> &queue, size_t n, size_t m,
              const std::vector<backend::device_vector<int>> &row, const std::vector<backend::device_vector<int>> &col,
              const std::vector<backend::device_vector<val_t>> &val, const std::vector<size_t> &nonzeros)
            : queue(queue), part(vex::partition(n, queue)), col_part(vex::partition(m, queue)),
              nrows(n), ncols(m), nnz(0), mtx(queue.size())
        {
            static_assert(std::is_same<val_t, double>::value || std::is_same<val_t, float>::value,
                    "SpMat value type must be float or double");
            const size_t nd = queue.size();
            precondition(row.size() == nd && col.size() == nd && val.size() == nd && nonzeros.size() == nd, "SpMat from device strips: one strip per device");
            precondition(m < (size_t(1) << 31), "SpMat: more than 2^31 columns");
            for (unsigned d = 0; d < nd; ++d) {
                precondition(row[d].size() == part[d + 1] - part[d] + 1 && col[d].size() >= nonzeros[d] && val[d].size() >= nonzeros[d]
                             && nonzeros[d] < (size_t(1) << 31), "SpMat from device strips: inconsistent strip");
                nnz += nonzeros[d];
            }
            std::vector<std::vector<col_t>> ghosts(nd);
            per_device([&](unsigned d) {
                mtx[d] = std::make_shared<device_part>(queue[d], part[d + 1] - part[d], nonzeros[d], row[d], col[d], val[d],
                        col_part[d], col_part[d + 1], ghosts[d], nd == 1);
            });
            if (nd > 1) { exc.setup(queue, col_part, ghosts); setup_halo(ghosts); }
            for (auto &m : mtx) m->release_strip();
        }

        /// From DEVICE CSR arrays (int32 row pointers and columns, `nonzeros` entries): nothing is staged through the
        /// host -- the library converts on the device (vexhip_spmat_create).  The reference only builds from host
        /// arrays (spmat.hpp:71-106: 12 GB of host CSR at 512^3); this is the constructor the headline runs through.
        /// Single-device contexts; a multi-device context takes one strip per device (the constructor below).
        SpMat(const std::vector<backend::command_queue> &queue, size_t n, size_t m, size_t nonzeros,
              const backend::device_vector<int> &row, const backend::device_vector<int> &col, const backend::device_vector<val_t> &val)
            : queue(queue), part(vex::partition(n, queue)), col_part(vex::partition(m, queue)),
              nrows(n), ncols(m), nnz(nonzeros), mtx(queue.size())
        {
            static_assert(std::is_same<val_t, double>::value || std::is_same<val_t, float>::value,
                    "SpMat value type must be float or double");
            precondition(queue.size() == 1, "SpMat from device arrays: single-device contexts only");
            precondition(row.size() == n + 1 && col.size() >= nonzeros && val.size() >= nonzeros, "SpMat: inconsistent CSR arrays");
            mtx[0] = std::make_shared<device_part>(queue[0], n, nonzeros, row, col, val);
        }

        /// The same with 64-bit ROW POINTERS: matrices with 2^31 entries and more on one device (the reference's default
        /// index type is size_t, spmat.hpp:56-57; vexhip_spmat_create_*_p64).  Columns stay 32-bit.
        SpMat(const std::vector<backend::command_queue> &queue, size_t n, size_t m, size_t nonzeros,
              const backend::device_vector<long long> &row, const backend::device_vector<int> &col, const backend::device_vector<val_t> &val)
            : queue(queue), part(vex::partition(n, queue)), col_part(vex::partition(m, queue)),
              nrows(n), ncols(m), nnz(nonzeros), mtx(queue.size())
        {
            static_assert(std::is_same<val_t, double>::value || std::is_same<val_t, float>::value,
                    "SpMat value type must be float or double");
            precondition(queue.size() == 1, "SpMat from device arrays: single-device contexts only");
            precondition(row.size() == n + 1 && col.size() >= nonzeros && val.size() >= nonzeros && m < (size_t(1) << 31), "SpMat: inconsistent CSR arrays");
            mtx[0] = std::make_shared<device_part>(queue[0], n, nonzeros, row, col, val);
        }

        size_t rows() const { return nrows; }
        size_t cols() const { return ncols; }
        size_t nonzeros() const { return nnz; }
        /// Storage the library chose for device d's local part (VEXHIP_SPMAT_*; info.matrix_bytes = bytes a product streams).
        const vexhip_spmat_info &storage_info(unsigned d = 0) const { return halo.active() ? halo.info(d) : mtx[d]->loc.info; }
        /// The library object behind device d's local part (NULL: empty) -- for the C ABI's queries (vexhip_spmat_axpby_fused, ...).
        const vexhip_spmat *storage_handle(unsigned d = 0) const { return mtx[d]->loc.handle.get(); }
        /// How a product on a multi-device context runs: "one launch per device (...)" -- the strip of every device stored with its
        /// two ghost planes, the neighbours' boundary planes of x read in place (vexhip_dist_spmv_create_halo_pull) -- or
        /// "pack / exchange / local / remote" (vexcl/exchange.hpp), or "one device".
        const char *step_kind() const {
            if (halo.active()) return halo.order == VEXHIP_PULL_FLAGS ? "one launch per device (ghost planes read in place behind flags)"
                                                                      : "one launch per device (ghost planes read in place, streams ordered by events)";
            return queue.size() > 1 && exc.active() ? "pack / exchange / local / remote" : (queue.size() > 1 ? "block-diagonal: local parts only" : "one device");
        }
        /// Why the one-launch step was not taken on this multi-device context ("" if it was, or on one device).
        const std::string &halo_declined() const { return halo.why; }

        /// y = alpha * A * x  or  y += alpha * A * x  (spmat.hpp:120-185).
        template <class T>
        void apply(const vex::vector<T> &x, vex::vector<T> &y, scalar_type alpha = 1, bool append = false) const {
            static_assert(std::is_same<T, val_t>::value, "vector and matrix value types differ");
            precondition(x.size() == ncols && y.size() == nrows, "SpMat::apply: incompatible sizes");
            if (halo.active()) { halo.apply(queue, x, y, alpha, append); return; }      // the whole step of a device in ONE launch (below)
            const bool exchange = queue.size() > 1 && exc.active();
            if (exchange) exc.start(x);                  // pack + ship on the secondary queues
            for (unsigned d = 0; d < queue.size(); ++d)   // local part, overlapped with the exchange
                if (part[d + 1] > part[d]) mtx[d]->mul_local(queue[d], x(d), y(d), alpha, append);
            if (exchange)
                for (unsigned d = 0; d < queue.size(); ++d) {
                    if (!exc.ghosts(d) || part[d + 1] == part[d]) continue;
                    exc.finish(d);                        // queue[d] waits for its ghosts
                    mtx[d]->mul_remote(queue[d], exc.ghost_buffer(d), y(d), alpha);
                }
        }

        static constexpr bool has_axpby_product = true;      // (operations.hpp: `y = z - A * x` is offered to apply_axpby)
        /// y = alpha * A * x + beta * z in ONE pass, if this matrix can: one device (no ghost exchange), double or float values, x, y and z three
        /// vectors of one partition with y != x (round 6: include/vexhip.h vexhip_spmat_apply_axpby_f64 -- the plane product takes the addend;
        /// other storages decline).  false: nothing was done, the caller takes the general route.
        /// What ends up here: `y = z - A * x` (a residual), `y = x + 2 * make_inline(A * x)` (detail::assign_any).
        template <class T>
        bool apply_axpby(const vex::vector<T> &x, vex::vector<T> &y, double alpha, const vex::vector<T> &z, double beta) const {
            if constexpr (!std::is_same<T, val_t>::value || !(std::is_same<T, double>::value || std::is_same<T, float>::value)) { (void)x; (void)y; (void)alpha; (void)z; (void)beta; return false; }
            else {
                if (queue.size() != 1 || halo.active() || x.size() != ncols || y.size() != nrows || z.size() != nrows || part[1] == part[0]) return false;
                const device_part &P = *mtx[0];
                if (P.loc.empty() || !P.loc.handle || x(0).raw() == y(0).raw()) return false;
                static const bool off = [] { const char *e = std::getenv("VEXCL_AXPBY"); return e && !std::strcmp(e, "off"); }();
                if (off) return false;
                // (only where the product takes the addend: elsewhere the general route costs the same or less -- a make_inline terminal keeps
                //  its product-into-a-vector form, 2.20 against 2.30 ms on the variable-coefficient 512^3 operator)
                if (!vexhip_spmat_axpby_fused(P.loc.handle.get(), x(0).raw(), z(0).raw(), y(0).raw())) return false;
                if constexpr (std::is_same<T, double>::value)
                    backend::check(vexhip_spmat_apply_axpby_f64(P.loc.handle.get(), queue[0].raw(), alpha, x(0).raw(), beta, z(0).raw(), y(0).raw()));
                else
                    backend::check(vexhip_spmat_apply_axpby_f32(P.loc.handle.get(), queue[0].raw(), (float)alpha, x(0).raw(), (float)beta, z(0).raw(), y(0).raw()));
                return true;
            }
        }

        /// The same product with its phases timed per device (HIP events on the primary queues; the call waits for them):
        /// ms[d] = {whole step, local part, wait for the ghosts after the local part, remote part}.  What the multi-device
        /// headline reports next to its wall time (examples/spmv_headline --devices).
        template <class T>
        void apply_timed(const vex::vector<T> &x, vex::vector<T> &y, std::vector<std::array<float, 4>> &ms, scalar_type alpha = 1, bool append = false) const {
            static_assert(std::is_same<T, val_t>::value, "vector and matrix value types differ");
            precondition(x.size() == ncols && y.size() == nrows, "SpMat::apply: incompatible sizes");
            const unsigned nd = static_cast<unsigned>(queue.size());
            struct events {                        // released on every way out (a failing launch throws past the loop below)
                std::vector<std::array<void *, 4>> e; std::vector<int> dev;
                ~events() { for (size_t d = 0; d < e.size(); ++d) for (void *x : e[d]) if (x) (void)vexhip_event_destroy(dev[d], x); }
                std::array<void *, 4> &operator[](size_t d) { return e[d]; }
            } ev;
            ev.e.assign(nd, std::array<void *, 4>{{nullptr, nullptr, nullptr, nullptr}});
            for (unsigned d = 0; d < nd; ++d) ev.dev.push_back(queue[d].device_ordinal());
            for (unsigned d = 0; d < nd; ++d)
                for (int k = 0; k < 4; ++k) backend::check(vexhip_event_create(queue[d].device_ordinal(), 1, &ev[d][k]));
            auto mark = [&](unsigned d, int k) { backend::check(vexhip_event_record(queue[d].device_ordinal(), ev[d][k], queue[d].raw())); };
            const bool exchange = nd > 1 && exc.active() && !halo.active();
            for (unsigned d = 0; d < nd; ++d) mark(d, 0);
            if (halo.active()) {         // one launch per device: there are no phases to tell apart -- {step, step, 0, 0}
                halo.apply(queue, x, y, alpha, append);
                for (unsigned d = 0; d < nd; ++d) { mark(d, 1); mark(d, 2); mark(d, 3); }
            } else {
            if (exchange) exc.start(x);
            for (unsigned d = 0; d < nd; ++d) {
                if (part[d + 1] > part[d]) mtx[d]->mul_local(queue[d], x(d), y(d), alpha, append);
                mark(d, 1);
            }
            for (unsigned d = 0; d < nd; ++d) {
                const bool rem = exchange && exc.ghosts(d) && part[d + 1] > part[d];
                if (rem) exc.finish(d);
                mark(d, 2);
                if (rem) mtx[d]->mul_remote(queue[d], exc.ghost_buffer(d), y(d), alpha);
                mark(d, 3);
            }
            }
            ms.assign(nd, std::array<float, 4>{{0, 0, 0, 0}});
            for (unsigned d = 0; d < nd; ++d) {
                const int dev = queue[d].device_ordinal();
                backend::check(vexhip_event_sync(dev, ev[d][3]));
                const int pair[4][2] = {{0, 3}, {0, 1}, {1, 2}, {2, 3}};
                for (int k = 0; k < 4; ++k) backend::check(vexhip_event_elapsed_ms(dev, ev[d][pair[k][0]], ev[d][pair[k][1]], &ms[d][k]));
            }
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
        /// A vector of this matrix's rows on device d that holds A * x for ONE make_inline terminal (keyed by the terminal's place in
        /// its expression): where the library's product is a hand-written kernel that a generated row function cannot match, the
        /// terminal is evaluated by that kernel into this vector and read back by the expression kernel (round 6).  Kernels of one
        /// queue run in order: the vector is free again when the next expression needs it.
        backend::device_vector<val_t> &inline_temporary(const std::string &key, unsigned d) const {
            auto &v = inline_tmp[key + "/" + std::to_string(d)];
            const size_t rows = part[d + 1] - part[d];
            if (v.size() != rows) v = backend::device_vector<val_t>(queue[d], rows);
            return v;
        }
        struct device_part;
        const device_part &part_of(unsigned d) const { return *mtx[d]; }
        const std::vector<backend::command_queue> &queue_list() const { return queue; }
        const std::vector<size_t> &row_partition() const { return part; }

        // One device's LOCAL matrix: a vexhip_spmat (include/vexhip.h) -- the library object that owns the storage
        // selection (hybrid-ELL width, diagonal / value codes, 32-bit columns or plain CSR) -- plus what
        // vexhip_spmat_get_info reports about it (make_inline reads the arrays in generated code).
        struct matrix_arrays {
            size_t n = 0, nnz = 0;
            std::shared_ptr<vexhip_spmat> handle;
            vexhip_spmat_info info = vexhip_spmat_info();
            // row-subset CSR of the REMOTE part (and the borrowed arrays of a local part kept in CSR)
            backend::device_vector<int> csr_ptr, csr_col; backend::device_vector<val_t> csr_val;
            size_t csr_nnz = 0;
            bool empty() const { return nnz == 0; }
        };

        static int spmat_create(int dev, void *s, int64_t n, const int *p, const int *c, const double *v, int fmt, int flags, vexhip_spmat **o) { return vexhip_spmat_create_f64_i32(dev, s, n, p, c, v, fmt, flags, o); }
        static int spmat_create(int dev, void *s, int64_t n, const int *p, const int *c, const float *v, int fmt, int flags, vexhip_spmat **o) { return vexhip_spmat_create_f32_i32(dev, s, n, p, c, v, fmt, flags, o); }
        static int spmat_create(int dev, void *s, int64_t n, const long long *p, const int *c, const double *v, int fmt, int flags, vexhip_spmat **o) { return vexhip_spmat_create_f64_p64(dev, s, n, reinterpret_cast<const int64_t *>(p), c, v, fmt, flags, o); }
        static int spmat_create(int dev, void *s, int64_t n, const long long *p, const int *c, const float *v, int fmt, int flags, vexhip_spmat **o) { return vexhip_spmat_create_f32_p64(dev, s, n, reinterpret_cast<const int64_t *>(p), c, v, fmt, flags, o); }
        static int spmat_apply(const vexhip_spmat *A, void *s, double a, int app, const double *x, double *y) { return vexhip_spmat_apply_f64(A, s, a, app, x, y); }
        static int spmat_apply(const vexhip_spmat *A, void *s, float a, int app, const float *x, float *y) { return vexhip_spmat_apply_f32(A, s, a, app, x, y); }
        static int spmat_apply_multi(const vexhip_spmat *A, void *s, int k, double a, int app, const double *const *x, double *const *y) { return vexhip_spmat_apply_multi_f64(A, s, k, a, app, x, y); }
        static int spmat_apply_multi(const vexhip_spmat *A, void *s, int k, float a, int app, const float *const *x, float *const *y) { return vexhip_spmat_apply_multi_f32(A, s, k, a, app, x, y); }

        struct device_part {
            matrix_arrays loc, rem;
            size_t n;

            device_part(const backend::command_queue &q, const idx_t *row_begin, const idx_t *row_end,
                    const col_t *col, const val_t *val, size_t col_begin, size_t col_end, std::vector<col_t> &ghost_cols, bool whole)
                : n(row_end - row_begin)
            {
                // The strip goes to the device as it is (row pointers rebased, indices narrowed to int32: one host pass,
                // no std::set, no second host copy) and is split THERE into the local part, the row-subset remote part
                // and the sorted ghost set (vexhip_csr_split_*; the reference does all of it on the host with a
                // std::set per device: spmat.hpp:291-378, csr.inl:92-131, hybrid_ell.inl:132-136).
                const size_t first = static_cast<size_t>(row_begin[0]), strip_nnz = static_cast<size_t>(row_end[0]) - first;
                if (strip_nnz >= (1ull << 31)) {
                    // 2^31 entries or more on this device: 64-bit row pointers (vexhip_spmat_create_*_p64).  The device-side
                    // split into local / remote parts works on 32-bit pointers, so this needs the device to own every
                    // column (a single-device context); a larger matrix on several devices stays below 2^31 per device.
                    precondition(whole && col_end < (1ull << 31), "SpMat: 2^31 or more nonzeros on one device of a multi-device context, or 2^31 or more columns");
                    std::vector<long long> sptr(n + 1);
                    for (size_t i = 0; i <= n; ++i) sptr[i] = static_cast<long long>(static_cast<size_t>(row_begin[i]) - first);
                    backend::device_vector<long long> dptr(q, n + 1, sptr.data());
                    std::vector<long long>().swap(sptr);
                    backend::device_vector<int> dcol(q, strip_nnz);
                    {   // columns narrowed in chunks: no second host copy of the whole array
                        const size_t chunk = size_t(1) << 26;
                        std::vector<int> buf(std::min(chunk, strip_nnz));
                        for (size_t o = 0; o < strip_nnz; o += chunk) {
                            const size_t k = std::min(chunk, strip_nnz - o);
                            for (size_t j = 0; j < k; ++j) buf[j] = static_cast<int>(col[first + o + j]);
                            dcol.write(q, o, k, buf.data(), true);
                        }
                    }
                    backend::device_vector<val_t> dval(q, strip_nnz, val + first);
                    set_local(q, dptr, dcol, dval, strip_nnz);
                    rem.n = n; rem.nnz = 0;
                    return;
                }
                precondition(col_end < (1ull << 31), "SpMat: more than 2^31 columns on one device");
                std::vector<int> sptr(n + 1), scol(strip_nnz);
                for (size_t i = 0; i <= n; ++i) sptr[i] = static_cast<int>(static_cast<size_t>(row_begin[i]) - first);
                for (size_t j = 0; j < strip_nnz; ++j) {
                    precondition(static_cast<size_t>(col[first + j]) < (1ull << 31), "SpMat: column index beyond 2^31");
                    scol[j] = static_cast<int>(col[first + j]);
                }
                backend::device_vector<int> dptr(q, n + 1, sptr.data()), dcol(q, strip_nnz, scol.data());
                backend::device_vector<val_t> dval(q, strip_nnz, val + first);
                std::vector<int>().swap(sptr); std::vector<int>().swap(scol);
                split_strip(q, dptr, dcol, dval, col_begin, col_end, ghost_cols);
            }

            /// One device's strip given as DEVICE CSR arrays (strip-local row pointers, global columns): split here.
            device_part(const backend::command_queue &q, size_t rows, size_t nonzeros, const backend::device_vector<int> &dptr,
                    const backend::device_vector<int> &dcol, const backend::device_vector<val_t> &dval,
                    size_t col_begin, size_t col_end, std::vector<col_t> &ghost_cols, bool whole)
                : n(rows)
            {
                if (whole && col_begin == 0) { set_local(q, dptr, dcol, dval, nonzeros); rem.n = n; rem.nnz = 0; return; }
                precondition(col_end < (1ull << 31), "SpMat: more than 2^31 columns on one device");
                split_strip(q, dptr, dcol, dval, col_begin, col_end, ghost_cols);
            }

            /// The strip on the device -> local part (a vexhip_spmat), row-subset remote part, sorted ghost set.
            void split_strip(const backend::command_queue &q, const backend::device_vector<int> &dptr, const backend::device_vector<int> &dcol,
                    const backend::device_vector<val_t> &dval, size_t col_begin, size_t col_end, std::vector<col_t> &ghost_cols)
            {
                const int dev = q.device_ordinal();
                strip_ptr = dptr; strip_col = dcol; strip_val = dval;
                int64_t sz[4] = {0, 0, 0, 0};
                backend::check(vexhip_csr_split_sizes_i32(dev, q.raw(), (int64_t)n, dptr.raw(), dcol.raw(), (int64_t)col_begin, (int64_t)col_end, sz));
                backend::device_vector<int> lptr(q, n + 1), lcol(q, (size_t)sz[0]);
                backend::device_vector<val_t> lval(q, (size_t)sz[0]);
                backend::device_vector<int> rrows(q, (size_t)sz[2]), rptr(q, (size_t)sz[2] + 1), rcol(q, (size_t)sz[1]), dghost(q, (size_t)sz[1]);
                backend::device_vector<val_t> rval(q, (size_t)sz[1]);
                backend::check(csr_split(dev, q.raw(), (int64_t)n, dptr.raw(), dcol.raw(), dval.raw(), (int64_t)col_begin, (int64_t)col_end, sz,
                            lptr.raw(), lcol.raw(), lval.raw(), rrows.raw(), rptr.raw(), rcol.raw(), rval.raw(), dghost.raw()));
                set_local(q, lptr, lcol, lval, (size_t)sz[0]);
                rem.n = n; rem.nnz = (size_t)sz[1];
                if (rem.nnz) {
                    rem_rows = rrows; rem.csr_ptr = rptr; rem.csr_col = rcol; rem.csr_val = rval; rem.csr_nnz = rem.nnz;
                    std::vector<int> g((size_t)sz[3]);
                    dghost.read(q, 0, g.size(), g.data(), true);
                    ghost_cols.assign(g.begin(), g.end());
                }
            }

            static int csr_split(int dev, void *s, int64_t n, const int *p, const int *c, const double *v, int64_t c0, int64_t c1, int64_t *sz,
                    int *lp, int *lc, double *lv, int *rr, int *rp, int *rc, double *rv, int *g)
            { return vexhip_csr_split_f64_i32(dev, s, n, p, c, v, c0, c1, sz, lp, lc, lv, rr, rp, rc, rv, g); }
            static int csr_split(int dev, void *s, int64_t n, const int *p, const int *c, const float *v, int64_t c0, int64_t c1, int64_t *sz,
                    int *lp, int *lc, float *lv, int *rr, int *rp, int *rc, float *rv, int *g)
            { return vexhip_csr_split_f32_i32(dev, s, n, p, c, v, c0, c1, sz, lp, lc, lv, rr, rp, rc, rv, g); }

            /// One device holding the whole matrix, given as DEVICE CSR arrays (int32 indices): no host staging.
            device_part(const backend::command_queue &q, size_t rows, size_t nonzeros, const backend::device_vector<int> &dptr,
                    const backend::device_vector<int> &dcol, const backend::device_vector<val_t> &dval)
                : n(rows)
            {
                set_local(q, dptr, dcol, dval, nonzeros);
                rem.n = n; rem.nnz = 0;
            }

            /// ... with 64-bit row pointers
            device_part(const backend::command_queue &q, size_t rows, size_t nonzeros, const backend::device_vector<long long> &dptr,
                    const backend::device_vector<int> &dcol, const backend::device_vector<val_t> &dval)
                : n(rows)
            {
                set_local(q, dptr, dcol, dval, nonzeros);
                rem.n = n; rem.nnz = 0;
            }

            backend::device_vector<int> rem_rows;

            // the strip as it came (device arrays, GLOBAL columns), kept while the constructor decides how the product runs
            backend::device_vector<int> strip_ptr, strip_col; backend::device_vector<val_t> strip_val;
            void release_strip() { strip_ptr = backend::device_vector<int>(); strip_col = backend::device_vector<int>(); strip_val = backend::device_vector<val_t>(); }

            static bool use_ell() {
#ifdef VEXCL_SPMAT_CSR
                return false;
#else
                return true;
#endif
            }

            /// The local part from device CSR arrays (columns already local): the library picks the storage.
            void set_local(const backend::command_queue &q, const backend::device_vector<int> &dptr,
                    const backend::device_vector<int> &dcol, const backend::device_vector<val_t> &dval, size_t nonzeros)
            {
                loc.n = n; loc.nnz = nonzeros;
                if (!loc.nnz || !n) return;
                vexhip_spmat *h = nullptr;
                backend::check(spmat_create(q.device_ordinal(), q.raw(), (int64_t)n, dptr.raw(), dcol.raw(), dval.raw(),
                            use_ell() ? VEXHIP_SPMAT_AUTO : VEXHIP_SPMAT_CSR, VEXHIP_SPMAT_BORROW_CSR, &h));
                loc.handle = std::shared_ptr<vexhip_spmat>(h, [](vexhip_spmat *p) { vexhip_spmat_destroy(p); });
                backend::check(vexhip_spmat_get_info(h, &loc.info));
                if (loc.info.format == VEXHIP_SPMAT_CSR) { loc.csr_ptr = dptr; loc.csr_col = dcol; loc.csr_val = dval; loc.csr_nnz = loc.nnz; }   // borrowed: keep alive
            }
            /// ... 64-bit row pointers: the library keeps its own copy of what it needs (nothing is borrowed)
            void set_local(const backend::command_queue &q, const backend::device_vector<long long> &dptr,
                    const backend::device_vector<int> &dcol, const backend::device_vector<val_t> &dval, size_t nonzeros)
            {
                loc.n = n; loc.nnz = nonzeros;
                if (!loc.nnz || !n) return;
                vexhip_spmat *h = nullptr;
                backend::check(spmat_create(q.device_ordinal(), q.raw(), (int64_t)n, dptr.raw(), dcol.raw(), dval.raw(),
                            use_ell() ? VEXHIP_SPMAT_AUTO : VEXHIP_SPMAT_CSR, 0, &h));
                loc.handle = std::shared_ptr<vexhip_spmat>(h, [](vexhip_spmat *p) { vexhip_spmat_destroy(p); });
                backend::check(vexhip_spmat_get_info(h, &loc.info));
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
                backend::check(spmat_apply_multi(loc.handle.get(), q.raw(), k, alpha, append ? 1 : 0, x, y));
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
                backend::check(spmat_apply(loc.handle.get(), q.raw(), alpha, append ? 1 : 0, x.raw(), y.raw()));
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
        // ---- the product step of a device in ONE launch (round 6; csrc/halo.hpp, plane.hip) -------------------------------------
        // Where every device's remote columns are the plane below its first row and the plane above its last one (a plane partition
        // of a 7-point operator on 512-point grid lines: the headline), the strip is stored ONCE, as a grid matrix that includes its
        // two ghost planes, and the plane product reads those planes from the NEIGHBOURS' segments of x where they lie (peer access)
        // -- no pack, no exchange, no remote part (the five phases of /root/reference/vexcl/spmat.hpp:120-185 and the set-up of
        // :291-378).  The bits are those of the one-device product (a row's entries stay in column order).
        // Distinct GPUs: flags in small uncached windows order the launches (VEXHIP_PULL_FLAGS).  Logical devices that share a GPU
        // (the reference's test fixture): events between the queues (VEXHIP_PULL_EVENTS).  VEXCL_HALO=off|flags|events overrides.
        // Double: 512-point lines through the plane product, lines of any other length through the grid product; float: 512-point lines
        // (plane32.hip).  Anything else -- general matrices, float on other grids -- keeps the exchange of vexcl/exchange.hpp; halo_declined() says why.
        struct halo_steps {
            struct dev_t {
                std::shared_ptr<vexhip_spmat> ext; std::shared_ptr<vexhip_ipc_window> win; std::shared_ptr<vexhip_dist_spmv> step;
                vexhip_spmat_info info = vexhip_spmat_info();
                bool lo = false, hi = false;
                void *ready = nullptr, *done = nullptr; int ordinal = 0;
                ~dev_t() { step.reset(); win.reset(); ext.reset(); if (ready) vexhip_event_destroy(ordinal, ready); if (done) vexhip_event_destroy(ordinal, done); }
            };
            std::vector<std::shared_ptr<dev_t>> dev;
            size_t H = 0; int order = 0; bool on = false;
            std::string why;
            bool active() const { return on; }
            const vexhip_spmat_info &info(unsigned d) const { return dev[d]->info; }

            template <class T>
            void apply(const std::vector<backend::command_queue> &q, const vex::vector<T> &x, vex::vector<T> &y, val_t alpha, bool append) const {
                const unsigned nd = static_cast<unsigned>(q.size());
                auto below = [&](unsigned d) -> const void * { return dev[d]->lo ? static_cast<const void *>(x(d - 1).raw() + (x.part_size(d - 1) - H)) : nullptr; };
                auto above = [&](unsigned d) -> const void * { return dev[d]->hi ? static_cast<const void *>(x(d + 1).raw()) : nullptr; };
                if (order == VEXHIP_PULL_FLAGS) {
                    for (unsigned d = 0; d < nd; ++d)
                        backend::check(vexhip_dist_spmv_apply_pull(dev[d]->step.get(), q[d].raw(), (double)alpha, append ? 1 : 0, x(d).raw(), y(d).raw(), below(d), above(d)));
                    return;
                }
                // events: x of the neighbours is final before a launch reads it; the launch has finished before they go on
                auto record = [&](unsigned d, void *e) { backend::check(vexhip_event_record(q[d].device_ordinal(), e, q[d].raw())); };
                auto wait = [&](unsigned d, void *e) { backend::check(vexhip_stream_wait_event(q[d].device_ordinal(), q[d].raw(), e)); };
                for (unsigned d = 0; d < nd; ++d) record(d, dev[d]->ready);
                for (unsigned d = 0; d < nd; ++d) {
                    if (dev[d]->lo) wait(d, dev[d - 1]->ready);
                    if (dev[d]->hi) wait(d, dev[d + 1]->ready);
                    backend::check(vexhip_dist_spmv_apply_pull(dev[d]->step.get(), q[d].raw(), (double)alpha, append ? 1 : 0, x(d).raw(), y(d).raw(), below(d), above(d)));
                    record(d, dev[d]->done);
                }
                for (unsigned d = 0; d < nd; ++d) {
                    if (dev[d]->lo) wait(d, dev[d - 1]->done);
                    if (dev[d]->hi) wait(d, dev[d + 1]->done);
                }
            }
        };
        halo_steps halo;

        template <class V> static typename std::enable_if<!std::is_same<V, double>::value, bool>::type is_double() { return false; }
        template <class V> static typename std::enable_if<std::is_same<V, double>::value, bool>::type is_double() { return true; }

        /// Decides whether this matrix on this context takes the one-launch step, and builds it (all devices at once).
        void setup_halo(const std::vector<std::vector<col_t>> &ghosts) {
            const unsigned nd = static_cast<unsigned>(queue.size());
            halo = halo_steps();
            int order = 0;
            {
                bool shared = false;
                for (unsigned a = 0; a < nd; ++a) for (unsigned b = a + 1; b < nd; ++b) shared = shared || queue[a].device_ordinal() == queue[b].device_ordinal();
                order = shared ? VEXHIP_PULL_EVENTS : VEXHIP_PULL_FLAGS;
                if (const char *e = std::getenv("VEXCL_HALO")) {
                    if (!std::strcmp(e, "off")) { halo.why = "VEXCL_HALO=off"; return; }
                    if (!std::strcmp(e, "events")) order = VEXHIP_PULL_EVENTS;
                    if (!std::strcmp(e, "flags") && !shared) order = VEXHIP_PULL_FLAGS;
                }
            }
            if (!exc.active()) { halo.why = "no ghost columns"; return; }
            if (nrows != ncols || part != col_part) { halo.why = "rows and columns are partitioned differently"; return; }
            // H = elements of a ghost plane: every device's remote columns lie within H of its own (rounded to whole pairs of 512-point lines)
            size_t reach = 0;
            std::vector<char> lo(nd, 0), hi(nd, 0);
            for (unsigned d = 0; d < nd; ++d) {
                if (part[d + 1] == part[d]) { halo.why = "a device owns no rows"; return; }
                for (const col_t &g : ghosts[d]) {
                    const size_t c = static_cast<size_t>(g);
                    if (c < col_part[d]) { lo[d] = 1; reach = std::max(reach, col_part[d] - c); }
                    else { hi[d] = 1; reach = std::max(reach, c + 1 - col_part[d + 1]); }
                }
            }
            // (a plane of the grid is at least `reach` long -- the first and the last points of a plane are boundary rows without
            //  neighbours -- and every strip is a whole number of planes: the smallest such divisor of the first strip; the plan of the
            //  stored strip decides below whether that IS the grid's plane)
            size_t Hh = 0;
            if (reach) {
                const size_t rows0 = part[1] - part[0];
                for (size_t k = std::min<size_t>(rows0 / reach, 65536); k >= 1 && !Hh; --k) {       // (planes per strip; thinner planes than a 65536th of a strip keep the exchange)
                    if (rows0 % k) continue;
                    const size_t h = rows0 / k;
                    bool whole = h >= reach;
                    for (unsigned d = 0; d < nd && whole; ++d) whole = (part[d + 1] - part[d]) % h == 0;
                    if (whole) Hh = h;
                }
            }
            if (!Hh) { halo.why = reach ? "the strips are not whole numbers of planes" : "no ghost columns"; return; }
            for (unsigned d = 0; d < nd; ++d) {
                const size_t rows = part[d + 1] - part[d];
                if (rows % Hh || rows < Hh) { halo.why = "a strip is not a whole number of planes of " + std::to_string(Hh) + " elements"; return; }
                if ((lo[d] && d == 0) || (hi[d] && d + 1 == nd)) { halo.why = "ghost columns outside the matrix"; return; }
                // symmetric coupling: the flag protocol hands `consumed` back to exactly the neighbours it reads from
                if (d + 1 < nd && hi[d] != lo[d + 1]) { halo.why = "the coupling between neighbouring strips is not symmetric"; return; }
            }
            std::vector<std::shared_ptr<typename halo_steps::dev_t>> dev(nd);
            std::vector<std::string> declined(nd);
            try {
                per_device([&](unsigned d) {
                    auto D = std::make_shared<typename halo_steps::dev_t>();
                    const backend::command_queue &q = queue[d];
                    const int ord = q.device_ordinal();
                    D->ordinal = ord; D->lo = lo[d]; D->hi = hi[d];
                    const device_part &P = *mtx[d];
                    const size_t rows = part[d + 1] - part[d], l = lo[d] ? Hh : 0, h = hi[d] ? Hh : 0, snnz = P.loc.nnz + P.rem.nnz;
                    if (P.strip_ptr.size() != rows + 1) { declined[d] = "the strip is not on the device any more"; return; }
                    backend::device_vector<int> eptr(q, l + rows + h + 1), ecol(q, std::max<size_t>(1, snnz));
                    int64_t bad = 0;
                    backend::check(vexhip_csr_extend_halo_i32(ord, q.raw(), (int64_t)rows, (int64_t)snnz, P.strip_ptr.raw(), P.strip_col.raw(), (int64_t)col_part[d],
                                (int64_t)l, (int64_t)h, eptr.raw(), ecol.raw(), &bad));
                    if (bad) { declined[d] = "columns outside the two ghost planes"; return; }
                    vexhip_spmat *e = nullptr;
                    backend::check(spmat_create(ord, q.raw(), (int64_t)(l + rows + h), eptr.raw(), ecol.raw(), P.strip_val.raw(), VEXHIP_SPMAT_AUTO, VEXHIP_SPMAT_SQUARE, &e));
                    D->ext = std::shared_ptr<vexhip_spmat>(e, [](vexhip_spmat *p) { vexhip_spmat_destroy(p); });
                    backend::check(vexhip_spmat_get_info(e, &D->info));
                    const auto &pl = D->info.plane; const auto &gr = D->info.grid;
                    const bool by_plane = pl.usable && (size_t)pl.lines_per_plane * 512 == Hh && (size_t)pl.planes * Hh == l + rows + h;
                    const bool by_grid = is_double<val_t>() && !pl.usable && gr.usable && D->info.format == VEXHIP_SPMAT_SELL8V && !D->info.tail_nnz
                                         && (size_t)gr.lines_per_plane * (size_t)gr.nx == Hh && (size_t)gr.planes * Hh == l + rows + h;
                    // round 6: or ANY strip stored with diagonal codes (a general banded operator, a coefficient per face) whose diagonals stay
                    // within one ghost range: the pair product's role (csrc/sell8.hip); the library checks the reach when the step is made
                    const bool by_codes = is_double<val_t>() && !by_plane && !by_grid
                                          && (D->info.format == VEXHIP_SPMAT_SELL8 || D->info.format == VEXHIP_SPMAT_SELL8V) && !D->info.tail_nnz && D->info.ell_width <= 8
                                          && Hh % 512 == 0 && rows % 512 == 0;
                    if (!by_plane && !by_grid && !by_codes) { declined[d] = "the strip with its ghost planes is stored neither for the plane / grid product nor with diagonal codes (ELL width <= 8, no CSR tail, whole 512-row slices)"; return; }
                    if (order == VEXHIP_PULL_FLAGS) {
                        vexhip_ipc_window *w = nullptr;
                        backend::check(vexhip_ipc_window_create(ord, (int)d, (int)nd, 0, &w));
                        D->win = std::shared_ptr<vexhip_ipc_window>(w, [](vexhip_ipc_window *p) { vexhip_ipc_window_destroy(p); });
                    } else {
                        backend::check(vexhip_event_create(ord, 0, &D->ready));
                        backend::check(vexhip_event_create(ord, 0, &D->done));
                    }
                    dev[d] = D;
                });
            } catch (const std::exception &e) { halo.why = std::string("set-up failed: ") + e.what(); return; }
            for (unsigned d = 0; d < nd; ++d) if (!dev[d]) { halo.why = "device " + std::to_string(d) + ": " + declined[d]; return; }
            try {
                for (unsigned d = 0; d < nd; ++d) {        // the windows exist on every device: attach the neighbours', then the steps
                    if (order == VEXHIP_PULL_FLAGS) {
                        if (lo[d]) backend::check(vexhip_ipc_window_attach(dev[d]->win.get(), (int)d - 1, dev[d - 1]->win.get()));
                        if (hi[d]) backend::check(vexhip_ipc_window_attach(dev[d]->win.get(), (int)d + 1, dev[d + 1]->win.get()));
                    }
                    vexhip_dist_spmv *st = nullptr;
                    backend::check(vexhip_dist_spmv_create_halo_pull(dev[d]->win.get(), dev[d]->ext.get(), (int64_t)(part[d + 1] - part[d]), (int64_t)Hh,
                                lo[d] ? (int)d - 1 : -1, hi[d] ? (int)d + 1 : -1, order, &st));
                    dev[d]->step = std::shared_ptr<vexhip_dist_spmv>(st, [](vexhip_dist_spmv *p) { vexhip_dist_spmv_destroy(p); });
                }
            } catch (const std::exception &e) { halo.why = std::string("set-up failed: ") + e.what(); return; }
            halo.dev = dev; halo.H = Hh; halo.order = order; halo.on = true;
            // the split parts are not needed any more: the stored strips hold every entry
            for (unsigned d = 0; d < nd; ++d) { mtx[d]->loc = matrix_arrays(); mtx[d]->rem = matrix_arrays(); mtx[d]->rem_rows = backend::device_vector<int>(); }
        }

        /// f(d) for every device, each on its own host thread when there are several (the C ABI keeps its state per thread
        /// and per device); the first exception is rethrown here.
        template <class F>
        void per_device(F f) const {
            const unsigned nd = static_cast<unsigned>(queue.size());
            if (nd <= 1 || std::getenv("VEXCL_SPMAT_SERIAL_SETUP")) { for (unsigned d = 0; d < nd; ++d) f(d); return; }
            std::vector<std::exception_ptr> err(nd);
            std::vector<std::thread> th;
            th.reserve(nd);
            struct joiner {                        // a thread that could not be started (resource limit) must not leave joinable ones behind: std::terminate
                std::vector<std::thread> &t;
                ~joiner() { for (auto &x : t) if (x.joinable()) x.join(); }
            } join_all{th};
            unsigned started = 0;
            try {
                for (; started < nd; ++started) {
                    const unsigned d = started;
                    th.emplace_back([&, d]() { try { f(d); } catch (...) { err[d] = std::current_exception(); } });
                }
            } catch (...) {
                for (unsigned d = started; d < nd; ++d) { try { f(d); } catch (...) { err[d] = std::current_exception(); } }     // the rest on this thread
            }
            for (auto &t : th) t.join();
            for (unsigned d = 0; d < nd; ++d) if (err[d]) std::rethrow_exception(err[d]);
        }

        std::vector<backend::command_queue> queue;
        std::vector<size_t> part, col_part;
        size_t nrows, ncols, nnz;
        std::vector<std::shared_ptr<device_part>> mtx;
        mutable std::map<std::string, backend::device_vector<val_t>> inline_tmp;

        template <class T, size_t N, class... Ts, size_t... I>
        void apply_components(const multivector<T, N> &x, detail::multi_target<Ts...> &y, scalar_type alpha, bool append,
                std::index_sequence<I...>) const
        {
            const bool exchange = queue.size() > 1 && exc.active();
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

        // ghost exchange (vexcl/exchange.hpp): pack kernels + ONE vexhip_halo_exchange per product (RCCL over xGMI
        // between distinct GPUs, event-ordered copies between logical devices of one GPU)
        detail::ghost_exchange<val_t> exc;
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
        c.src.parameter("const " + V + " *", "values"); c.src.parameter("const int *", "blocks"); c.src.parameter("const char *", "pool");
        c.src.parameter("const int *", "csr_row"); c.src.parameter("const int *", "csr_col");
        c.src.parameter("const " + V + " *", "csr_val"); c.src.parameter("const " + V + " *", "in");
        c.src.parameter("long", "grid_nx"); c.src.parameter("long", "grid_far"); c.src.parameter("long", "grid_pitch");
        c.src.parameter("const int *", "line_class"); c.src.parameter("const uchar *", "grid_table");
        c.src.parameter("long", "x_last");
        c.src.parameter("ulong", "i");
        c.src.end_function_parameters();
        // Round 6: every branch below is straight-line code -- ALL requests of a row (codes, values, x) are issued before the first sum
        // needs one.  Until then a missing entry was skipped by a branch around its two loads, and a row's seven entries cost seven
        // round trips to memory one after the other (1.97 ms for the 512^3 Poisson product inside an expression kernel, where the
        // library product takes 0.38: profiles/r05_examples_roofline_cpp.log).  Now the index of a missing entry is clamped into x
        // and the term is dropped by a select (never multiplied: x may hold Inf / NaN there); the sums are formed in storage order,
        // as before (hybrid_ell.inl:322-351).
        c.src.new_line() << V << " sum = 0;";
        c.src.new_line() << "if (line_class)";        // stored by grid line (grid.hip): a class per line, a value code per position and row
        c.src.open("{");
        // (the row's line and place in it: a 32-bit division where the row number allows it -- the 64-bit one by a run-time
        //  divisor is a hundred instructions per row in a kernel that does little else)
        c.src.new_line() << "long line, r;";
        c.src.new_line() << "if (i < 0x100000000ul) { const uint l32 = (uint)i / (uint)grid_nx; line = l32; r = (uint)i - l32 * (uint)grid_nx; }";
        c.src.new_line() << "else { line = (long)i / grid_nx; r = (long)i - line * grid_nx; }";
        c.src.new_line() << "const uchar *tb = grid_table + (long)line_class[line] * 7 * grid_pitch + r;";
        c.src.new_line() << "const long off[7] = {-grid_far, -grid_nx, -1, 0, 1, grid_nx, grid_far};";
        c.src.new_line() << "uint code[7]; " << V << " xv[7], av[7];";
        c.src.new_line() << "for(int p = 0; p < 7; ++p) code[p] = tb[p * grid_pitch];";
        c.src.new_line() << "for(int p = 0; p < 7; ++p) { long j = (long)i + off[p]; j = j < 0 ? 0 : j; j = j > x_last ? x_last : j; xv[p] = in[j]; }";
        c.src.new_line() << "for(int p = 0; p < 7; ++p) av[p] = values[code[p]];";
        c.src.new_line() << "for(int p = 0; p < 7; ++p) sum = code[p] != 255u ? sum + av[p] * xv[p] : sum;";
        c.src.close("}");
        c.src.new_line() << "else if (values && ell_w <= 8)";       // SELL8V: diagonal codes and value codes (include/vexhip.h)
        c.src.open("{");
        c.src.new_line() << "const long wp = (ell_w + 1) / 2;";
        c.src.new_line() << "const long slice = blocks ? (long)blocks[i >> 9] : (long)(i >> 9);";   // slice dictionary (include/vexhip.h)
        c.src.new_line() << "const uint *cw = (const uint *)(sell + slice * (wp * 2048)) + ((i & 511) >> 1);";
        c.src.new_line() << "const uint *vw = cw + wp * 256;";
        c.src.new_line() << "uint cword[4], vword[4]; " << V << " xv[8], av[8]; uint real[8];";
        c.src.new_line() << "for(int u = 0; u < 4; ++u) { const long uu = u < wp ? u : wp - 1; cword[u] = cw[uu * 256]; vword[u] = vw[uu * 256]; }";
        c.src.new_line() << "for(int j = 0; j < 8; ++j)";
        c.src.open("{");
        c.src.new_line() << "const int sh = 8 * ((j & 1) * 2 + (int)(i & 1));";
        c.src.new_line() << "const uint code = (cword[j >> 1] >> sh) & 255u;";
        c.src.new_line() << "real[j] = (j < ell_w && code < 254u) ? 1u : 0u;";
        c.src.new_line() << "long jx = (long)i + deltas[real[j] ? code : 0u]; jx = jx < 0 ? 0 : jx; jx = jx > x_last ? x_last : jx;";
        c.src.new_line() << "xv[j] = in[jx]; av[j] = values[(vword[j >> 1] >> sh) & 255u];";
        c.src.close("}");
        c.src.new_line() << "for(int j = 0; j < 8; ++j) sum = real[j] ? sum + av[j] * xv[j] : sum;";
        c.src.close("}");
        c.src.new_line() << "else if (values)";       // ... wider than eight columns: the loop
        c.src.open("{");
        c.src.new_line() << "const long wp = (ell_w + 1) / 2;";
        c.src.new_line() << "const long slice = blocks ? (long)blocks[i >> 9] : (long)(i >> 9);";
        c.src.new_line() << "const uint *cw = (const uint *)(sell + slice * (wp * 2048)) + ((i & 511) >> 1);";
        c.src.new_line() << "const uint *vw = cw + wp * 256;";
        c.src.new_line() << "for(long j = 0; j < ell_w; ++j)";
        c.src.open("{");
        c.src.new_line() << "const int sh = 8 * ((j & 1) * 2 + (i & 1));";
        c.src.new_line() << "const uint code = (cw[(j >> 1) * 256] >> sh) & 255u;";
        c.src.new_line() << "if (code < 254u) sum += values[(vw[(j >> 1) * 256] >> sh) & 255u] * in[(long)i + deltas[code]];";
        c.src.close("}");
        c.src.close("}");
        c.src.new_line() << "else if (deltas && ell_w <= 8)";       // SELL8: 1-byte diagonal codes
        c.src.open("{");
        c.src.new_line() << "const long wp = (ell_w + 1) / 2;";
        c.src.new_line() << "const char *slice = sell + (i >> 9) * (wp * 1024 + ell_w * 512 * sizeof(" << V << "));";
        c.src.new_line() << "const uint *cw = (const uint *)(blocks ? pool + (long)blocks[i >> 9] * (wp * 1024) : slice) + ((i & 511) >> 1);";
        c.src.new_line() << "const " << V << " *ell_val = (const " << V << " *)(slice + wp * 1024) + (i & 511);";
        c.src.new_line() << "uint cword[4]; " << V << " xv[8], av[8]; uint real[8];";
        c.src.new_line() << "for(int u = 0; u < 4; ++u) { const long uu = u < wp ? u : wp - 1; cword[u] = cw[uu * 256]; }";
        c.src.new_line() << "for(int j = 0; j < 8; ++j)";
        c.src.open("{");
        c.src.new_line() << "const uint code = (cword[j >> 1] >> (8 * ((j & 1) * 2 + (int)(i & 1)))) & 255u;";
        c.src.new_line() << "real[j] = (j < ell_w && code < 254u) ? 1u : 0u;";
        c.src.new_line() << "long jx = (long)i + deltas[real[j] ? code : 0u]; jx = jx < 0 ? 0 : jx; jx = jx > x_last ? x_last : jx;";
        c.src.new_line() << "xv[j] = in[jx]; av[j] = ell_val[(j < ell_w ? j : ell_w - 1) * 512];";
        c.src.close("}");
        c.src.new_line() << "for(int j = 0; j < 8; ++j) sum = real[j] ? sum + av[j] * xv[j] : sum;";
        c.src.close("}");
        c.src.new_line() << "else if (deltas)";
        c.src.open("{");
        c.src.new_line() << "const long wp = (ell_w + 1) / 2;";
        c.src.new_line() << "const char *slice = sell + (i >> 9) * (wp * 1024 + ell_w * 512 * sizeof(" << V << "));";
        c.src.new_line() << "const uint *cw = (const uint *)(blocks ? pool + (long)blocks[i >> 9] * (wp * 1024) : slice) + ((i & 511) >> 1);";
        c.src.new_line() << "const " << V << " *ell_val = (const " << V << " *)(slice + wp * 1024) + (i & 511);";
        c.src.new_line() << "for(long j = 0; j < ell_w; ++j)";
        c.src.open("{");
        c.src.new_line() << "const uint code = (cw[(j >> 1) * 256] >> (8 * ((j & 1) * 2 + (i & 1)))) & 255u;";
        c.src.new_line() << "if (code < 254u) sum += ell_val[j * 512] * in[(long)i + deltas[code]];";
        c.src.close("}");
        c.src.close("}");
        c.src.new_line() << "else";                   // SELL-512 with 32-bit columns: groups of four entries, their requests issued together
        c.src.open("{");
        c.src.new_line() << "const char *slice = sell + (i >> 9) * (ell_w * 512 * (4 + sizeof(" << V << ")));";
        c.src.new_line() << "const int *ell_col = (const int *)slice + (i & 511);";
        c.src.new_line() << "const " << V << " *ell_val = (const " << V << " *)(slice + ell_w * 2048) + (i & 511);";
        c.src.new_line() << "long j = 0;";
        c.src.new_line() << "for(; j + 4 <= ell_w; j += 4)";
        c.src.open("{");
        c.src.new_line() << "int cc[4]; " << V << " av[4], xv[4];";
        c.src.new_line() << "for(int u = 0; u < 4; ++u) { cc[u] = ell_col[(j + u) * 512]; av[u] = ell_val[(j + u) * 512]; }";
        c.src.new_line() << "for(int u = 0; u < 4; ++u) xv[u] = in[cc[u] < 0 ? 0 : cc[u]];";
        c.src.new_line() << "for(int u = 0; u < 4; ++u) sum = cc[u] >= 0 ? sum + av[u] * xv[u] : sum;";
        c.src.close("}");
        c.src.new_line() << "for(; j < ell_w; ++j)";
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
        c.src.parameter("const " + V + " *", name + "_values"); c.src.parameter("const int *", name + "_blocks");
        c.src.parameter("const char *", name + "_pool");
        c.src.parameter("const int *", name + "_csr_row"); c.src.parameter("const int *", name + "_csr_col");
        c.src.parameter("const " + V + " *", name + "_csr_val"); c.src.parameter("const " + V + " *", name + "_vec");
        c.src.parameter("long", name + "_grid_nx"); c.src.parameter("long", name + "_grid_far"); c.src.parameter("long", name + "_grid_pitch");
        c.src.parameter("const int *", name + "_line_class"); c.src.parameter("const uchar *", name + "_grid_table");
        c.src.parameter("long", name + "_x_last");
        c.src.parameter("const " + V + " *", name + "_product");       // != NULL: A * x, evaluated by the library's kernel in front of this one
    }
    void local_init(gen_context &c) const { c.next(); }
    void emit(gen_context &c) const {
        std::string n = c.next();
        c.src << "(" << n << "_product ? " << n << "_product[idx] : " << n << "_hell_spmv(" << n << "_ell_w, " << n << "_sell, " << n << "_deltas, " << n << "_values, " << n << "_blocks, " << n << "_pool, "
              << n << "_csr_row, " << n << "_csr_col, " << n << "_csr_val, " << n << "_vec, "
              << n << "_grid_nx, " << n << "_grid_far, " << n << "_grid_pitch, " << n << "_line_class, " << n << "_grid_table, " << n << "_x_last, idx))";
    }
    /// Where the product comes from (round 6).  The generated row function issues 22 requests per row of a matrix stored by grid line --
    /// class, seven codes, seven values, seven elements of x -- and runs at a fifth of the library's plane product (1.76 against 0.38 ms
    /// at 512^3 even as straight-line code, profiles/r06_roofline_inline.log); the coded storages (SELL8V / SELL8: grid, plane, march,
    /// pair products) therefore evaluate A * x with THEIR kernel into a vector the matrix keeps for this terminal, and the expression
    /// kernel reads that vector -- two launches, 16 bytes per row more, still 2 x faster.  Matrices kept with 32-bit columns or in CSR
    /// (no hand-written product that beats a row function by much) stay fused.  VEXCL_INLINE_SPMV=fused | product overrides.
    bool through_product(const vexhip_spmat_info &L) const {
        static const int forced = [] { const char *e = std::getenv("VEXCL_INLINE_SPMV"); return !e ? 0 : !std::strcmp(e, "fused") ? 1 : !std::strcmp(e, "product") ? 2 : 0; }();
        if (forced) return forced == 2 && L.nnz > 0;
        return L.nnz > 0 && (L.format == VEXHIP_SPMAT_SELL8V || L.format == VEXHIP_SPMAT_SELL8);
    }
    void set_args(arg_context &a) const {
        const std::string key = a.next();
        const auto &L = A.part_of(a.device).loc.info;      // zero-initialised when the local part is empty
        const bool csr_rows = L.format == VEXHIP_SPMAT_CSR || L.tail_nnz > 0;
        // (a matrix kept in CSR with 64-bit row pointers has no 32-bit pointer array for the generated terminal)
        precondition(!(L.format == VEXHIP_SPMAT_CSR && L.nnz > 0 && !L.csr_ptr), "make_inline: CSR storage with 64-bit row pointers is not supported in generated code");
        a.krn.push_arg((long)L.ell_width);
        a.krn.push_arg(static_cast<const char *>(L.sell));
        a.krn.push_arg(static_cast<const int *>(L.ndeltas > 0 ? L.deltas : nullptr));
        a.krn.push_arg(static_cast<const T *>(L.nvalues > 0 ? L.values : nullptr));
        a.krn.push_arg(static_cast<const int *>(L.slice_blocks));
        a.krn.push_arg(static_cast<const char *>(L.code_pool));
        a.krn.push_arg(static_cast<const int *>(csr_rows ? L.csr_ptr : nullptr));
        a.krn.push_arg(static_cast<const int *>(L.csr_col)); a.krn.push_arg(static_cast<const T *>(L.csr_val));
        a.krn.push_arg(static_cast<const T *>(x(a.device).raw()));
        // stored by grid line (no SELL-512 slices at all): the class of every line and the class tables
        const bool by_line = L.grid.usable && !L.sell && !L.code_pool;
        a.krn.push_arg((long)L.grid.nx); a.krn.push_arg((long)L.grid.nx * (long)L.grid.lines_per_plane); a.krn.push_arg((long)L.grid.pitch);
        a.krn.push_arg(static_cast<const int *>(by_line ? L.grid.line_class : nullptr));
        a.krn.push_arg(static_cast<const unsigned char *>(by_line ? L.grid.table : nullptr));
        a.krn.push_arg((long)x(a.device).size() - 1);        // the last element of x: clamped requests stay inside the vector
        const T *product = nullptr;
        if (through_product(L)) {
            backend::device_vector<T> &tmp = A.inline_temporary(key, a.device);
            A.part_of(a.device).mul_local(A.queue_list()[a.device], x(a.device), tmp, T(1), false);
            product = tmp.raw();
        }
        a.krn.push_arg(product);
    }
    void get_props(prop_context &p) const {
        if (p.empty()) { p.queue = A.queue_list(); p.part = A.row_partition(); p.size = A.rows(); }
    }
};
} // namespace detail

namespace detail {
// (operations.hpp: `y = z + c * make_inline(A * x)` is the shape y = beta z + alpha A x as well)
template <class M, class T> struct axpby_leaf<inline_spmv<M, T>, typename std::enable_if<has_apply_axpby<M>::value>::type> : std::integral_constant<int, 2> {
    template <class Y> static bool apply(const inline_spmv<M, T> &e, Y &y, double alpha, const void *z, double beta) {
        if constexpr (std::is_same<Y, vector<T>>::value) return e.A.apply_axpby(e.x, y, alpha, *static_cast<const Y *>(z), beta);
        else { (void)e; (void)y; (void)alpha; (void)z; (void)beta; return false; }
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
