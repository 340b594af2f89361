#ifndef VEXCL_SPARSE_PRODUCT_HPP
#define VEXCL_SPARSE_PRODUCT_HPP
// A * x as an inlinable terminal: the product is evaluated inside the fused
// kernel, with x itself an arbitrary vector expression evaluated at
// idx = col[j] (reference: vexcl/sparse/product.hpp:45-131).
#include "../operations.hpp"
#include "../vector.hpp"

namespace vex {
namespace sparse {

template <class Matrix, class X>
struct matrix_vector_product : detail::expression_base {
    typedef typename std::common_type<typename Matrix::value_type, typename X::value_type>::type value_type;
    const Matrix &A; X x;
    matrix_vector_product(const Matrix &A, const X &x) : A(A), x(x) {}

    void preamble(detail::gen_context &c) const { Matrix::product_preamble(x, c, c.next()); }
    void params(detail::gen_context &c) const { Matrix::product_params(x, c, c.next()); }
    void local_init(detail::gen_context &c) const { Matrix::template product_local_init<value_type>(x, c, c.next()); }
    void emit(detail::gen_context &c) const { c.src << c.next() << "_sum"; }
    void set_args(detail::arg_context &a) const { a.next(); A.product_args(x, a); }
    void get_props(detail::prop_context &p) const {
        if (p.empty()) { p.queue = A.queue_list(); p.part = {0, A.rows()}; p.size = A.rows(); }
    }
};

namespace detail {
    /// Emits "sum += val * ( x-expression at idx )" with x traversed under its own prefix.
    template <class X>
    void append_product(const X &x, vex::detail::gen_context &c, const std::string &name, const std::string &val) {
        { vex::detail::gen_context i(c, name + "_x"); x.local_init(i); }
        c.src.new_line() << name << "_sum += " << val << " * ( ";
        { vex::detail::gen_context i(c, name + "_x"); x.emit(i); }
        c.src << " );";
    }
}

} // namespace sparse
} // namespace vex
#endif
