#ifndef VEXCL_SPARSE_MATRIX_HPP
#define VEXCL_SPARSE_MATRIX_HPP
// vex::sparse::matrix<Val, Col, Ptr>: picks the format that suits the device
// (reference: vexcl/sparse/matrix.hpp:11-150 -- csr for CPU devices, ell for
// GPUs).  Every device here is an MI355X, so it is the ELL format.
#include "csr.hpp"
#include "ell.hpp"

namespace vex {
namespace sparse {
template <typename Val, typename Col = int, typename Ptr = Col>
class matrix : public ell<Val, Col, Ptr> {
    public:
        typedef ell<Val, Col, Ptr> Base;
        template <class PtrRange, class ColRange, class ValRange>
        matrix(const std::vector<backend::command_queue> &q, size_t nrows, size_t ncols,
                const PtrRange &ptr, const ColRange &col, const ValRange &val, bool fast_setup = true)
            : Base(q, nrows, ncols, ptr, col, val, fast_setup) {}
        matrix(const backend::command_queue &q) : Base(q) {}

        template <class Expr>
        friend typename std::enable_if<vex::detail::is_expr<Expr>::value,
            matrix_vector_product<Base, vex::detail::as_expr_t<Expr>>>::type
        operator*(const matrix &A, const Expr &x) {
            return matrix_vector_product<Base, vex::detail::as_expr_t<Expr>>(A, vex::detail::as_expr<Expr>::get(x));
        }
};
} // namespace sparse
} // namespace vex
#endif
