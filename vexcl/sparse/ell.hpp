#ifndef VEXCL_SPARSE_ELL_HPP
#define VEXCL_SPARSE_ELL_HPP
// vex::sparse::ell<Val, Col, Ptr>: hybrid ELL + CSR-tail matrix, inlinable
// product (reference: vexcl/sparse/ell.hpp:62-510).  The CSR -> ELL conversion
// runs on the device (libvexhip: vexhip_hell_analyze_i32 / vexhip_hell_fill_*,
// the counterpart of ell.hpp:400-508 `convert_csr2ell`).
#include "product.hpp"

namespace vex {
namespace sparse {

template <typename Val, typename Col = int, typename Ptr = Col>
class ell {
    public:
        typedef Val value_type; typedef Val val_type; typedef Col col_type; typedef Ptr ptr_type;
        static_assert(std::is_same<Col, int>::value && std::is_same<Ptr, int>::value,
                "sparse::ell on MI355X stores int32 indices");
        // float / double: converted on the device; any other value type (blocks, with rhs_of and
        // spmv_ops_impl specialized): the same layout filled on the host
        static const bool device_fill = std::is_same<Val, double>::value || std::is_same<Val, float>::value;

        template <class PtrRange, class ColRange, class ValRange>
        ell(const std::vector<backend::command_queue> &q, size_t nrows, size_t ncols,
                const PtrRange &ptr, const ColRange &col, const ValRange &val, bool = true)
            : q(q[0]), n(nrows), m(ncols), nnz(val.size()), ell_width(0), ell_pitch(alignup(nrows, 16)), csr_nnz(0)
        {
            precondition(q.size() == 1, "sparse::ell is only supported for single-device contexts");
            if (!n || !nnz) return;
            backend::device_vector<int> dptr(this->q, ptr.size(), &ptr[0]);
            if constexpr (device_fill) {
                backend::device_vector<int> dcol(this->q, col.size(), &col[0]);
                backend::device_vector<Val> dval(this->q, val.size(), &val[0]);
                convert(dptr, dcol, dval);
            } else {
                convert_on_host(dptr, ptr, col, val);
            }
        }
        ell(const backend::command_queue &q) : q(q), n(0), m(0), nnz(0), ell_width(0), ell_pitch(0), csr_nnz(0) {}

        size_t rows() const { return n; }
        size_t cols() const { return m; }
        size_t nonzeros() const { return nnz; }
        size_t width() const { return ell_width; }
        size_t csr_nonzeros() const { return csr_nnz; }
        std::vector<backend::command_queue> queue_list() const { return std::vector<backend::command_queue>(1, q); }

        template <class Expr>
        friend typename std::enable_if<vex::detail::is_expr<Expr>::value,
            matrix_vector_product<ell, vex::detail::as_expr_t<Expr>>>::type
        operator*(const ell &A, const Expr &x) {
            return matrix_vector_product<ell, vex::detail::as_expr_t<Expr>>(A, vex::detail::as_expr<Expr>::get(x));
        }

        // ---- codegen (ell.hpp:195-343) ------------------------------------------
        template <class X> static void product_preamble(const X &x, vex::detail::gen_context &c, const std::string &name) {
            vex::detail::gen_context i(c, name + "_x"); x.preamble(i);
        }
        template <class X> static void product_params(const X &x, vex::detail::gen_context &c, const std::string &name) {
            c.src.template parameter<size_t>(name + "_ell_width");
            c.src.template parameter<size_t>(name + "_ell_pitch");
            c.src.template parameter<global_ptr<const Col>>(name + "_ell_col");
            c.src.template parameter<global_ptr<const Val>>(name + "_ell_val");
            c.src.template parameter<global_ptr<const Ptr>>(name + "_csr_ptr");
            c.src.template parameter<global_ptr<const Col>>(name + "_csr_col");
            c.src.template parameter<global_ptr<const Val>>(name + "_csr_val");
            vex::detail::gen_context i(c, name + "_x"); x.params(i);
        }
        template <class R, class X> static void product_local_init(const X &x, vex::detail::gen_context &c, const std::string &name) {
            spmv_ops_impl<Val, R>::decl_accum_var(c.src, name + "_sum");      // R: the value type of x
            c.src.open("{");
            c.src.new_line() << "for(size_t j = 0; j < " << name << "_ell_width; ++j)";
            c.src.open("{");
            c.src.new_line() << "size_t nnz_idx = idx + j * " << name << "_ell_pitch;";
            c.src.new_line() << type_name<Col>() << " c = " << name << "_ell_col[nnz_idx];";
            c.src.new_line() << "if (c != (" << type_name<Col>() << ")(-1))";
            c.src.open("{");
            c.src.new_line() << type_name<Col>() << " idx = c;";
            detail::append_product<Val>(x, c, name, name + "_ell_val[nnz_idx]");
            c.src.close("} else break;");
            c.src.close("}");
            c.src.new_line() << "if (" << name << "_csr_ptr)";
            c.src.open("{");
            c.src.new_line() << type_name<Ptr>() << " csr_beg = " << name << "_csr_ptr[idx];";
            c.src.new_line() << type_name<Ptr>() << " csr_end = " << name << "_csr_ptr[idx+1];";
            c.src.new_line() << "for(" << type_name<Ptr>() << " j = csr_beg; j < csr_end; ++j)";
            c.src.open("{");
            c.src.new_line() << type_name<Col>() << " idx = " << name << "_csr_col[j];";
            detail::append_product<Val>(x, c, name, name + "_csr_val[j]");
            c.src.close("}");
            c.src.close("}");
            c.src.close("}");
        }
        template <class X> void product_args(const X &x, vex::detail::arg_context &a) const {
            a.krn.push_arg(ell_width); a.krn.push_arg(ell_pitch);
            a.krn.push_arg(static_cast<const Col *>(ell_col.raw())); a.krn.push_arg(static_cast<const Val *>(ell_val.raw()));
            a.krn.push_arg(static_cast<const Ptr *>(csr_nnz ? csr_ptr.raw() : nullptr));
            a.krn.push_arg(static_cast<const Col *>(csr_col.raw())); a.krn.push_arg(static_cast<const Val *>(csr_val.raw()));
            vex::detail::arg_context i(a); x.set_args(i);
        }
    private:
        backend::command_queue q;
        size_t n, m, nnz, ell_width, ell_pitch, csr_nnz;
        backend::device_vector<Col> ell_col; backend::device_vector<Val> ell_val;
        backend::device_vector<Ptr> csr_ptr; backend::device_vector<Col> csr_col; backend::device_vector<Val> csr_val;

        static int fill(int dev, void *s, int64_t n, const int *p, const int *c, const double *v, int64_t w, int64_t pitch,
                int *ec, double *ev, int *cp, int *cc, double *cv) { return vexhip_hell_fill_f64_i32(dev, s, n, p, c, v, w, pitch, ec, ev, cp, cc, cv); }
        static int fill(int dev, void *s, int64_t n, const int *p, const int *c, const float *v, int64_t w, int64_t pitch,
                int *ec, float *ev, int *cp, int *cc, float *cv) { return vexhip_hell_fill_f32_i32(dev, s, n, p, c, v, w, pitch, ec, ev, cp, cc, cv); }

        /// Same split as the device conversion: the first `width` entries of a row go to the ELL part
        /// (column -1 marks padding), the rest to the CSR tail.
        template <class PtrRange, class ColRange, class ValRange>
        void convert_on_host(const backend::device_vector<int> &dptr, const PtrRange &ptr, const ColRange &col, const ValRange &val) {
            int64_t w = 0, tail = 0;
            backend::check(vexhip_hell_analyze_i32(q.device_ordinal(), q.raw(), (int64_t)n, dptr.raw(), &w, &tail));
            ell_width = (size_t)w; csr_nnz = (size_t)tail;
            std::vector<Col> ec(ell_pitch * ell_width, Col(-1)), cc; std::vector<Val> ev(ell_pitch * ell_width, Val()), cv;
            std::vector<Ptr> cp(n + 1, 0);
            for (size_t i = 0; i < n; ++i) {
                size_t j = 0;
                for (Ptr k = ptr[i]; k < ptr[i + 1]; ++k, ++j) {
                    if (j < ell_width) { ec[i + j * ell_pitch] = col[k]; ev[i + j * ell_pitch] = val[k]; }
                    else { cc.push_back(col[k]); cv.push_back(val[k]); }
                }
                cp[i + 1] = (Ptr)cc.size();
            }
            precondition(cc.size() == csr_nnz, "sparse::ell: host and device disagree on the CSR tail");
            if (ell_width) { ell_col = backend::device_vector<Col>(q, ec.size(), ec.data()); ell_val = backend::device_vector<Val>(q, ev.size(), ev.data()); }
            if (csr_nnz) {
                csr_ptr = backend::device_vector<Ptr>(q, cp.size(), cp.data());
                csr_col = backend::device_vector<Col>(q, cc.size(), cc.data()); csr_val = backend::device_vector<Val>(q, cv.size(), cv.data());
            }
        }

        void convert(const backend::device_vector<int> &dptr, const backend::device_vector<int> &dcol, const backend::device_vector<Val> &dval) {
            int dev = q.device_ordinal();
            int64_t w = 0, tail = 0;
            backend::check(vexhip_hell_analyze_i32(dev, q.raw(), (int64_t)n, dptr.raw(), &w, &tail));
            ell_width = (size_t)w; csr_nnz = (size_t)tail;
            if (w) { ell_col = backend::device_vector<Col>(q, ell_pitch * w); ell_val = backend::device_vector<Val>(q, ell_pitch * w); }
            if (tail) { csr_ptr = backend::device_vector<Ptr>(q, n + 1); csr_col = backend::device_vector<Col>(q, tail); csr_val = backend::device_vector<Val>(q, tail); }
            backend::check(fill(dev, q.raw(), (int64_t)n, dptr.raw(), dcol.raw(), dval.raw(), w, (int64_t)ell_pitch,
                        ell_col.raw(), ell_val.raw(), csr_ptr.raw(), csr_col.raw(), csr_val.raw()));
            q.finish();
        }
};

} // namespace sparse
} // namespace vex
#endif
