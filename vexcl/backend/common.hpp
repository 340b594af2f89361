#ifndef VEXCL_BACKEND_COMMON_HPP
#define VEXCL_BACKEND_COMMON_HPP
// Compile-option / program-header stacks and the kernel cache directory live in backend.hpp here
// (reference: vexcl/backend/common.hpp:61-285).
#include "../backend.hpp"
#endif
