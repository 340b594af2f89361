#ifndef VEXCL_SPMAT_CCSR_HPP
#define VEXCL_SPMAT_CCSR_HPP
// vex::SpMatCCSR<val_t, col_t, idx_t>: compressed CSR for matrices with few
// UNIQUE rows (stencil operators) -- reference vexcl/spmat/ccsr.hpp:55-280.
//     y[i] = sum_{j in [row[idx[i]], row[idx[i]+1])} val[j] * x[i + col[j]]
// `col` holds column positions relative to the diagonal, `idx[i]` selects the
// unique row used by matrix row i.  Single queue.  A * x is a vector-expression
// terminal: the product is evaluated inside the fused kernel through a device
// function (ccsr.hpp:184-200), so `y += A * x`, `y = x - A * x`, `sum(x * (A * x))`
// all work.  For the 512^3 Poisson operator this removes the 12 B/nnz matrix
// stream: the kernel reads 4 B of idx per row instead of 84 B of (col, val).
// Device storage: idx and row narrowed to 32 bits, col to int32.
// The stand-alone product `y (=|+=) s * (A * x)` is handed to the library's matrix object (build_fast below): the operator
// expanded to CSR on the device is exactly what vex::SpMat stores with 1-byte diagonal / value codes and a slice
// dictionary -- 0.72 instead of 0.83 ms per product at 512^3, same summation order, same bits.
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>
#include "../operations.hpp"
#include "../vector.hpp"

namespace vex {
template <class T, size_t N> class multivector;   // ../multivector.hpp

template <typename val_t, typename col_t = ptrdiff_t, typename idx_t = size_t>
struct SpMatCCSR {
    static_assert(std::is_signed<col_t>::value, "Column type for CCSR format has to be signed.");
    typedef val_t value_type;

    /// n rows, m unique rows (ccsr.hpp:70-87).
    SpMatCCSR(const backend::command_queue &queue, size_t n, size_t m,
            const idx_t *idx, const idx_t *row, const col_t *col, const val_t *val)
        : queue(queue), n(n), m(m)
    {
        precondition(m < (1ull << 32) && static_cast<size_t>(row[m]) < (1ull << 31), "SpMatCCSR: too many unique rows");
        std::vector<unsigned> idx32(idx, idx + n), row32(row, row + m + 1);
        std::vector<int> col32(static_cast<size_t>(row[m]));
        for (size_t j = 0; j < col32.size(); ++j) {
            precondition(col[j] > -(1ll << 31) && col[j] < (1ll << 31), "SpMatCCSR: column offset exceeds 32 bits");
            col32[j] = static_cast<int>(col[j]);
        }
        this->idx = backend::device_vector<unsigned>(queue, n, idx32.data(), backend::MEM_READ_ONLY);
        this->row = backend::device_vector<unsigned>(queue, m + 1, row32.data(), backend::MEM_READ_ONLY);
        this->col = backend::device_vector<int>(queue, col32.size(), col32.data(), backend::MEM_READ_ONLY);
        this->val = backend::device_vector<val_t>(queue, col32.size(), val, backend::MEM_READ_ONLY);
        entries = col32.size();
        far_offset = 0;
        for (int c : col32) far_offset = std::max<long long>(far_offset, c < 0 ? -(long long)c : (long long)c);
        // entries of the expanded operator, in 64 bits (the device-side expansion scans 32-bit row lengths)
        expanded = 0;
        for (size_t i = 0; i < n; ++i) expanded += (unsigned long long)(row32[idx32[i] + 1] - row32[idx32[i]]);
        try { build_fast(); }
        catch (const backend::error &e) {
            // only an allocation that did not fit keeps the CCSR kernel (it needs none of that memory); anything else -- a sticky
            // device fault, a bad argument -- must not disappear here
            if (!e.out_of_memory()) throw;                 // the hipError_t of the failure (vexhip_last_error_code), not a search in its text
            fast.reset();
        }
    }

    size_t rows() const { return n; }

    /// y = alpha * A * x  or  y += alpha * A * x with the hand-written kernel (libvexhip `vexhip_spmv_ccsr_*`).
    void apply(const vector<val_t> &x, vector<val_t> &y, val_t alpha = 1, bool append = false) const {
        precondition(x.nparts() == 1 && y.nparts() == 1 && x.size() == n && y.size() == n, "SpMatCCSR::apply: incompatible vectors");
        if (fast) { backend::check(spmat_apply(fast.get(), queue.raw(), alpha, append ? 1 : 0, x(0).raw(), y(0).raw())); return; }
        backend::check(spmv(queue.device_ordinal(), queue.raw(), (int64_t)n, alpha, append ? 1 : 0, idx.raw(), (int64_t)m,
                    row.raw(), col.raw(), val.raw(), (int64_t)entries, (int64_t)far_offset, x(0).raw(), y(0).raw()));
    }

    /// The operator handed to the library's matrix object (include/vexhip.h vexhip_spmat): expanded to CSR on the device --
    /// entries in table order, i.e. the order this class sums them in -- it becomes what vex::SpMat makes of such a matrix
    /// (1-byte diagonal and value codes, slice dictionary): 0.68-0.70 ms instead of 0.83 ms per product at 512^3, same
    /// bits.  Operators that are not float / double, are small, or refer to columns outside [0, n) keep the CCSR kernel
    /// (as does VEXCL_CCSR_KERNEL=1, for A/B).
    void build_fast() {
        if (!(std::is_same<val_t, double>::value || std::is_same<val_t, float>::value)) return;
        if (n < 32768 || entries == 0 || std::getenv("VEXCL_CCSR_KERNEL")) return;
        // The expansion is a transient CSR matrix (12 or 8 bytes per entry + the row pointers) next to the storage built from
        // it: an operator whose CSR form has 2^31 entries or more, or does not fit in what is free now (with a margin
        // for the SELL storage), simply keeps the CCSR kernel -- being compact is the point of this class.
        if (expanded == 0 || expanded >= (1ull << 31)) return;
        const int dev = queue.device_ordinal();
        uint64_t free_b = 0, total_b = 0;
        backend::check(vexhip_mem_info(dev, &free_b, &total_b));
        const unsigned long long need = expanded * (4 + sizeof(val_t)) + (n + 1) * 4ull + expanded * (1 + sizeof(val_t)) + (64ull << 20);
        if (need > free_b) return;
        backend::device_vector<int> ptr(queue, n + 1);
        int64_t nnz = 0;
        backend::check(to_csr(dev, queue.raw(), (int64_t)n, idx.raw(), row.raw(), col.raw(), val.raw(), ptr.raw(), nullptr, nullptr, &nnz));
        if (nnz <= 0) return;
        backend::device_vector<int> ccol(queue, (size_t)nnz);
        backend::device_vector<val_t> cval(queue, (size_t)nnz);
        int64_t bad = 0;
        backend::check(to_csr(dev, queue.raw(), (int64_t)n, idx.raw(), row.raw(), col.raw(), val.raw(), ptr.raw(), ccol.raw(), cval.raw(), &bad));
        if (bad) return;
        vexhip_spmat *h = nullptr;
        backend::check(spmat_create(dev, queue.raw(), (int64_t)n, ptr.raw(), ccol.raw(), cval.raw(), &h));
        std::shared_ptr<vexhip_spmat> owner(h, [](vexhip_spmat *p) { vexhip_spmat_destroy(p); });
        vexhip_spmat_info info;
        backend::check(vexhip_spmat_get_info(h, &info));
        // without diagonal structure to exploit the CCSR kernel reads less: keep it (the object is released with `owner`)
        if (info.format == VEXHIP_SPMAT_SELL8V || info.format == VEXHIP_SPMAT_SELL8) fast = owner;
    }
    static int to_csr(int d, void *s, int64_t n, const unsigned *idx, const unsigned *row, const int *col, const double *val, int *ptr, int *oc, double *ov, int64_t *nnz)
    { return vexhip_ccsr_to_csr_f64_i32(d, s, n, idx, row, col, val, ptr, oc, ov, nnz); }
    static int to_csr(int d, void *s, int64_t n, const unsigned *idx, const unsigned *row, const int *col, const float *val, int *ptr, int *oc, float *ov, int64_t *nnz)
    { return vexhip_ccsr_to_csr_f32_i32(d, s, n, idx, row, col, val, ptr, oc, ov, nnz); }
    template <class V> static int to_csr(int, void *, int64_t, const unsigned *, const unsigned *, const int *, const V *, int *, int *, V *, int64_t *nnz) { *nnz = 0; return 0; }
    static int spmat_create(int d, void *s, int64_t n, const int *p, const int *c, const double *v, vexhip_spmat **o) { return vexhip_spmat_create_f64_i32(d, s, n, p, c, v, VEXHIP_SPMAT_AUTO, 0, o); }
    static int spmat_create(int d, void *s, int64_t n, const int *p, const int *c, const float *v, vexhip_spmat **o) { return vexhip_spmat_create_f32_i32(d, s, n, p, c, v, VEXHIP_SPMAT_AUTO, 0, o); }
    template <class V> static int spmat_create(int, void *, int64_t, const int *, const int *, const V *, vexhip_spmat **o) { *o = nullptr; return 0; }
    static int spmat_apply(const vexhip_spmat *A, void *s, double a, int app, const double *x, double *y) { return vexhip_spmat_apply_f64(A, s, a, app, x, y); }
    static int spmat_apply(const vexhip_spmat *A, void *s, float a, int app, const float *x, float *y) { return vexhip_spmat_apply_f32(A, s, a, app, x, y); }
    template <class V> static int spmat_apply(const vexhip_spmat *, void *, V, int, const V *, V *) { return 0; }

    static int spmv(int dev, void *s, int64_t n, double a, int app, const unsigned *idx, int64_t m, const unsigned *row,
            const int *col, const double *val, int64_t e, int64_t far, const double *x, double *y) {
        return vexhip_spmv_ccsr_f64(dev, s, n, a, app, idx, m, row, col, val, e, far, x, y); }
    static int spmv(int dev, void *s, int64_t n, float a, int app, const unsigned *idx, int64_t m, const unsigned *row,
            const int *col, const float *val, int64_t e, int64_t far, const float *x, float *y) {
        return vexhip_spmv_ccsr_f32(dev, s, n, a, app, idx, m, row, col, val, e, far, x, y); }

    size_t entries = 0;
    unsigned long long expanded = 0;             // entries of the operator written out as CSR
    long long far_offset = 0;
    std::shared_ptr<vexhip_spmat> fast;           // the same operator as a vexhip_spmat (build_fast); copies of *this share it

    backend::command_queue queue;
    size_t n, m;
    backend::device_vector<unsigned> idx, row;
    backend::device_vector<int> col;
    backend::device_vector<val_t> val;
};

namespace detail {

/// The terminal A * x (ccsr.hpp:90-113, codegen :160-262).
template <typename val_t, typename col_t, typename idx_t, typename T>
struct ccsr_product : expression_base {
    typedef typename std::common_type<val_t, T>::type value_type;
    typedef SpMatCCSR<val_t, col_t, idx_t> matrix;
    const matrix &A; const vector<T> &x;
    ccsr_product(const matrix &A, const vector<T> &x) : A(A), x(x) {
        precondition(x.nparts() == 1 && x.size() == A.n, "SpMatCCSR product needs a single-queue vector of matching size");
    }
    void preamble(gen_context &c) const {
        std::string name = c.next();
        const std::string V = type_name<val_t>(), X = type_name<T>(), R = type_name<value_type>();
        c.src.begin_function(R, name + "_ccsr_spmv");
        c.src.begin_function_parameters();
        c.src.parameter("const uint *", "idx"); c.src.parameter("const uint *", "row");
        c.src.parameter("const int *", "col"); c.src.parameter("const " + V + " *", "val");
        c.src.parameter("const " + X + " *", "vec"); c.src.parameter("ulong", "i");
        c.src.end_function_parameters();
        // Same sum, same order as ccsr.hpp:184-200, but 8 entries at a time: the 8 table
        // loads are independent, then the 8 gathers are independent -- three dependent
        // round trips per stencil row instead of two per entry (the plain loop is
        // latency-bound at 512^3: 1.74 ms).  Padding entries read vec[i] times 0.
        c.src.new_line() << R << " sum = 0;";
        c.src.new_line() << "const uint pos = idx[i], end = row[pos+1];";
        c.src.new_line() << "for(uint j0 = row[pos]; j0 < end; j0 += 8)";
        c.src.open("{");
        c.src.new_line() << "int c[8]; " << V << " v[8]; " << X << " xv[8];";
        c.src.new_line() << "#pragma unroll";
        c.src.new_line() << "for(int k = 0; k < 8; ++k) { const bool in = j0 + k < end; c[k] = in ? col[j0 + k] : 0; v[k] = in ? val[j0 + k] : (" << V << ")0; }";
        c.src.new_line() << "#pragma unroll";
        c.src.new_line() << "for(int k = 0; k < 8; ++k) xv[k] = vec[(long)i + c[k]];";
        c.src.new_line() << "#pragma unroll";
        c.src.new_line() << "for(int k = 0; k < 8; ++k) if (j0 + k < end) sum += v[k] * xv[k];";
        c.src.close("}");
        c.src.new_line() << "return sum;";
        c.src.end_function();
    }
    void params(gen_context &c) const {
        std::string n = c.next();
        c.src.parameter("const uint *", n + "_idx"); c.src.parameter("const uint *", n + "_row");
        c.src.parameter("const int *", n + "_col"); c.src.parameter("const " + type_name<val_t>() + " *", n + "_val");
        c.src.parameter("const " + type_name<T>() + " *", n + "_vec");
    }
    void local_init(gen_context &c) const { c.next(); }
    void emit(gen_context &c) const {
        std::string n = c.next();
        c.src << n << "_ccsr_spmv(" << n << "_idx, " << n << "_row, " << n << "_col, " << n << "_val, " << n << "_vec, idx)";
    }
    void set_args(arg_context &a) const {
        a.next();
        a.krn.push_arg(static_cast<const unsigned *>(A.idx.raw())); a.krn.push_arg(static_cast<const unsigned *>(A.row.raw()));
        a.krn.push_arg(static_cast<const int *>(A.col.raw())); a.krn.push_arg(static_cast<const val_t *>(A.val.raw()));
        a.krn.push_arg(static_cast<const T *>(x(0).raw()));
    }
    void get_props(prop_context &p) const {
        if (p.empty()) { p.queue = std::vector<backend::command_queue>(1, A.queue); p.part = {0, A.n}; p.size = A.n; }
    }
};

// `y (=|+=|-=) A * x` and `... s * (A * x)`: the hand-written kernel; anything larger: the fused kernel.
template <typename val_t, typename col_t, typename idx_t>
struct direct_assign<ccsr_product<val_t, col_t, idx_t, val_t>,
        typename std::enable_if<std::is_same<val_t, double>::value || std::is_same<val_t, float>::value>::type> : std::true_type {
    template <class W>
    static void apply(W &y, const ccsr_product<val_t, col_t, idx_t, val_t> &p, double scale, bool append) {
        p.A.apply(p.x, y, static_cast<val_t>(scale), append);
    }
};
template <typename S, typename val_t, typename col_t, typename idx_t>
struct direct_assign<binary_expr<tag::multiplies, scalar_terminal<S>, ccsr_product<val_t, col_t, idx_t, val_t>>,
        typename std::enable_if<std::is_same<val_t, double>::value || std::is_same<val_t, float>::value>::type> : std::true_type {
    template <class W, class E>
    static void apply(W &y, const E &e, double scale, bool append) {
        e.r.A.apply(e.r.x, y, static_cast<val_t>(scale * static_cast<double>(e.l.v)), append);
    }
};

} // namespace detail

template <typename val_t, typename col_t, typename idx_t, typename T>
detail::ccsr_product<val_t, col_t, idx_t, T>
operator*(const SpMatCCSR<val_t, col_t, idx_t> &A, const vector<T> &x) {
    return detail::ccsr_product<val_t, col_t, idx_t, T>(A, x);
}

namespace detail {
/// A * X with X a multivector: component I of the multi-expression is the product A * X(I)
/// (ccsr.hpp:240-280 of the reference; tests/spmv.cpp:345-437).
template <typename val_t, typename col_t, typename idx_t, typename T, size_t N>
struct ccsr_multi_product : expression_base {
    typedef T value_type;
    const SpMatCCSR<val_t, col_t, idx_t> &A; const multivector<T, N> &x;
    ccsr_multi_product(const SpMatCCSR<val_t, col_t, idx_t> &A, const multivector<T, N> &x) : A(A), x(x) {}
    void get_props(prop_context &p) const { ccsr_product<val_t, col_t, idx_t, T>(A, x(0)).get_props(p); }
};
template <typename val_t, typename col_t, typename idx_t, typename T, size_t N>
struct mv_dim<ccsr_multi_product<val_t, col_t, idx_t, T, N>> : std::integral_constant<size_t, N> {};
template <size_t I, typename val_t, typename col_t, typename idx_t, typename T, size_t N>
struct component_of<I, ccsr_multi_product<val_t, col_t, idx_t, T, N>, void> {
    typedef ccsr_product<val_t, col_t, idx_t, T> type;
    static type get(const ccsr_multi_product<val_t, col_t, idx_t, T, N> &p) { return type(p.A, p.x(I)); }
};
} // namespace detail

template <typename val_t, typename col_t, typename idx_t, typename T, size_t N>
detail::ccsr_multi_product<val_t, col_t, idx_t, T, N>
operator*(const SpMatCCSR<val_t, col_t, idx_t> &A, const multivector<T, N> &x) {
    return detail::ccsr_multi_product<val_t, col_t, idx_t, T, N>(A, x);
}

} // namespace vex
#endif
