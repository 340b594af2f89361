#ifndef VEXCL_SPMAT_INLINE_SPMV_HPP
#define VEXCL_SPMAT_INLINE_SPMV_HPP
// vex::make_inline(A * x) lives in spmat.hpp (reference: vexcl/spmat/inline_spmv.hpp:70-198).
#include "../spmat.hpp"
#endif
