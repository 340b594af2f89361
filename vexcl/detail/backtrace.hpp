#ifndef VEXCL_DETAIL_BACKTRACE_HPP
#define VEXCL_DETAIL_BACKTRACE_HPP
// vex::detail::print_backtrace(): the call stack, printed when a kernel fails to build or launch
// (reference: vexcl/detail/backtrace.hpp:40-63, glibc's backtrace facility).
#include <cstdio>
#include <cstdlib>
#if defined(__GLIBC__)
#  include <execinfo.h>
#endif
namespace vex { namespace detail {
inline void print_backtrace() {
#if defined(__GLIBC__)
    void *frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
#endif
}
} }
#endif
