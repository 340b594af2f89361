#ifndef VEXCL_SPARSE_PRODUCT_HPP
#define VEXCL_SPARSE_PRODUCT_HPP
// A * x as an inlinable terminal: the product is evaluated inside the fused
// kernel, with x itself an arbitrary vector expression evaluated at
// idx = col[j] (reference: vexcl/sparse/product.hpp:45-131).
#include "../operations.hpp"
#include "../vector.hpp"

namespace vex {
namespace sparse {

/// The vector element a matrix element multiplies; specialize for block values, e.g. a 2x2 block
/// multiplies a 2-vector (reference: vexcl/sparse/spmv_ops.hpp:40-50).
template <class MatVal, class Enable = void> struct rhs_of { typedef MatVal type; };

/// How `sum += A_ij * x_j` is written for the pair of value types; specialize together with rhs_of
/// (spmv_ops.hpp:52-80 -- same two hooks: the accumulator's declaration, one product's accumulation).
template <class MatVal, class VecVal, class Enable = void>
struct spmv_ops_impl {
    static void decl_accum_var(backend::source_generator &src, const std::string &name) {
        src.new_line() << type_name<typename std::common_type<MatVal, VecVal>::type>() << " " << name << " = 0;";
    }
    static void append_product(backend::source_generator &src, const std::string &sum, const std::string &mat_val, const std::string &vec_val) {
        src.new_line() << sum << " += " << mat_val << " * " << vec_val << ";";
    }
};

/// Value type of A * x: the usual promotion for arithmetic values, the vector's type for blocks.
template <class MatVal, class VecVal, class Enable = void> struct product_value { typedef VecVal type; };
template <class MatVal, class VecVal>
struct product_value<MatVal, VecVal, typename std::enable_if<std::is_arithmetic<MatVal>::value && std::is_arithmetic<VecVal>::value>::type> {
    typedef typename std::common_type<MatVal, VecVal>::type type;
};

template <class Matrix, class X>
struct matrix_vector_product : detail::expression_base {
    typedef typename product_value<typename Matrix::value_type, typename X::value_type>::type value_type;
    const Matrix &A; X x;
    matrix_vector_product(const Matrix &A, const X &x) : A(A), x(x) {}

    void preamble(detail::gen_context &c) const { Matrix::product_preamble(x, c, c.next()); }
    void params(detail::gen_context &c) const { Matrix::product_params(x, c, c.next()); }
    void local_init(detail::gen_context &c) const { Matrix::template product_local_init<typename X::value_type>(x, c, c.next()); }
    void emit(detail::gen_context &c) const { c.src << c.next() << "_sum"; }
    void set_args(detail::arg_context &a) const { a.next(); A.product_args(x, a); }
    void get_props(detail::prop_context &p) const {
        if (p.empty()) { p.queue = A.queue_list(); p.part = {0, A.rows()}; p.size = A.rows(); }
    }
};

namespace detail {
    /// Emits "sum += val * ( x-expression at idx )" with x traversed under its own prefix; the form of
    /// the accumulation is spmv_ops_impl's.
    template <class MatVal, class X>
    void append_product(const X &x, vex::detail::gen_context &c, const std::string &name, const std::string &val) {
        typedef typename X::value_type XV;
        { vex::detail::gen_context i(c, name + "_x"); x.local_init(i); }
        c.src.new_line() << "const " << type_name<XV>() << " vex_xv = ( ";
        { vex::detail::gen_context i(c, name + "_x"); x.emit(i); }
        c.src << " );";
        spmv_ops_impl<MatVal, XV>::append_product(c.src, name + "_sum", val, "vex_xv");
    }
}

} // namespace sparse
} // namespace vex
#endif
