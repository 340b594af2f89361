#ifndef VEXCL_CONSTANTS_HPP
#define VEXCL_CONSTANTS_HPP
// Compile-time constants baked into the kernel text (reference:
// vexcl/constants.hpp): vex::constants::pi() etc. and std::integral_constant.
#include <limits>
#include "operations.hpp"

namespace vex {
namespace detail {
template <class Impl>
struct text_constant : expression_base {
    typedef double value_type;
    void preamble(gen_context &) const {} void params(gen_context &) const {} void local_init(gen_context &) const {}
    void emit(gen_context &c) const { c.src << Impl::text(); }
    void set_args(arg_context &) const {} void get_props(prop_context &) const {}
};
}
namespace detail {
/// std::integral_constant<T, v> as an operand: the value is part of the kernel
/// text (constants.hpp:60-90 of the reference), `x = std::integral_constant<int, 42>()`.
template <class T, T v>
struct integral_text : expression_base {
    typedef T value_type;
    void preamble(gen_context &) const {} void params(gen_context &) const {} void local_init(gen_context &) const {}
    void emit(gen_context &c) const { c.src << "( " << v << " )"; }
    void set_args(arg_context &) const {} void get_props(prop_context &) const {}
};
template <class T, T v> struct is_extra_operand<std::integral_constant<T, v>> : std::true_type {};
template <class T, T v> struct as_expr<std::integral_constant<T, v>, void> {
    typedef integral_text<T, v> type;
    static type get(const std::integral_constant<T, v> &) { return type(); }
};
}
namespace constants {
#define VEXCL_CONSTANT(name, value)                                                         \
    struct name##_impl { static const char *text() { return #value; } };                   \
    inline detail::text_constant<name##_impl> name() { return detail::text_constant<name##_impl>(); }
VEXCL_CONSTANT(pi, 3.14159265358979323846)
VEXCL_CONSTANT(two_pi, 6.28318530717958647692)
VEXCL_CONSTANT(half_pi, 1.57079632679489661923)
VEXCL_CONSTANT(e, 2.71828182845904523536)
VEXCL_CONSTANT(root_two, 1.41421356237309504880)
VEXCL_CONSTANT(half, 0.5)
#undef VEXCL_CONSTANT
}
} // namespace vex
#endif
