#ifndef VEXCL_SPARSE_SPMV_OPS_HPP
#define VEXCL_SPARSE_SPMV_OPS_HPP
// rhs_of / spmv_ops_impl, the customization points for block-valued matrices, live in product.hpp
// (reference: vexcl/sparse/spmv_ops.hpp:40-61).
#include "product.hpp"
#endif
