#ifndef VEXCL_SPARSE_CSR_HPP
#define VEXCL_SPARSE_CSR_HPP
// vex::sparse::csr<Val, Col, Ptr>: single-queue CSR matrix whose product is an
// inlinable terminal (reference: vexcl/sparse/csr.hpp:48-196).
#include "product.hpp"

namespace vex {
namespace sparse {

template <typename Val, typename Col = int, typename Ptr = Col>
class csr {
    public:
        typedef Val value_type; typedef Val val_type; typedef Col col_type; typedef Ptr ptr_type;

        template <class PtrRange, class ColRange, class ValRange>
        csr(const std::vector<backend::command_queue> &q, size_t nrows, size_t ncols,
                const PtrRange &ptr, const ColRange &col, const ValRange &val, bool = true)
            : q(q[0]), n(nrows), m(ncols), nnz(val.size()),
              ptr(q[0], ptr.size(), ptr.size() ? &ptr[0] : nullptr),
              col(q[0], col.size(), col.size() ? &col[0] : nullptr),
              val(q[0], val.size(), val.size() ? &val[0] : nullptr)
        { precondition(q.size() == 1, "sparse::csr is only supported for single-device contexts"); }

        /// Dummy matrix: passes empty parameters to kernels (csr.hpp:74-77).
        csr(const backend::command_queue &q) : q(q), n(0), m(0), nnz(0) {}

        size_t rows() const { return n; }
        size_t cols() const { return m; }
        size_t nonzeros() const { return nnz; }
        std::vector<backend::command_queue> queue_list() const { return std::vector<backend::command_queue>(1, q); }

        template <class Expr>
        friend typename std::enable_if<vex::detail::is_expr<Expr>::value,
            matrix_vector_product<csr, vex::detail::as_expr_t<Expr>>>::type
        operator*(const csr &A, const Expr &x) {
            return matrix_vector_product<csr, vex::detail::as_expr_t<Expr>>(A, vex::detail::as_expr<Expr>::get(x));
        }

        // ---- codegen (csr.hpp:86-195) ------------------------------------------
        template <class X> static void product_preamble(const X &x, vex::detail::gen_context &c, const std::string &name) {
            vex::detail::gen_context i(c, name + "_x"); x.preamble(i);
        }
        template <class X> static void product_params(const X &x, vex::detail::gen_context &c, const std::string &name) {
            c.src.template parameter<global_ptr<const Ptr>>(name + "_ptr");
            c.src.template parameter<global_ptr<const Col>>(name + "_col");
            c.src.template parameter<global_ptr<const Val>>(name + "_val");
            vex::detail::gen_context i(c, name + "_x"); x.params(i);
        }
        template <class R, class X> static void product_local_init(const X &x, vex::detail::gen_context &c, const std::string &name) {
            spmv_ops_impl<Val, R>::decl_accum_var(c.src, name + "_sum");      // R: the value type of x
            c.src.new_line() << "if (" << name << "_ptr)";
            c.src.open("{");
            c.src.new_line() << type_name<Ptr>() << " row_beg = " << name << "_ptr[idx];";
            c.src.new_line() << type_name<Ptr>() << " row_end = " << name << "_ptr[idx+1];";
            c.src.new_line() << "for(" << type_name<Ptr>() << " j = row_beg; j < row_end; ++j)";
            c.src.open("{");
            c.src.new_line() << type_name<Col>() << " idx = " << name << "_col[j];";
            detail::append_product<Val>(x, c, name, name + "_val[j]");
            c.src.close("}");
            c.src.close("}");
        }
        template <class X> void product_args(const X &x, vex::detail::arg_context &a) const {
            a.krn.push_arg(static_cast<const Ptr *>(nnz ? ptr.raw() : nullptr));
            a.krn.push_arg(static_cast<const Col *>(col.raw()));
            a.krn.push_arg(static_cast<const Val *>(val.raw()));
            vex::detail::arg_context i(a); x.set_args(i);
        }

        const backend::device_vector<Ptr> &ptr_data() const { return ptr; }
        const backend::device_vector<Col> &col_data() const { return col; }
        const backend::device_vector<Val> &val_data() const { return val; }
    private:
        backend::command_queue q;
        size_t n, m, nnz;
        backend::device_vector<Ptr> ptr;
        backend::device_vector<Col> col;
        backend::device_vector<Val> val;
};

} // namespace sparse
} // namespace vex
#endif
