#ifndef VEXCL_CAST_HPP
#define VEXCL_CAST_HPP
// vex::cast<T>(expr): changes the deduced type of an expression and prints an explicit
// conversion (reference: vexcl/cast.hpp:38-150; tests/cast.cpp:9-16).  The convert_<T>N /
// as_<T>N helpers of the reference work on cl_<T>N vector types, which are out of scope.
#include "operations.hpp"

namespace vex {
namespace detail {
template <class T, class E>
struct casted_expression : expression_base {
    typedef T value_type;
    E expr;
    explicit casted_expression(const E &e) : expr(e) {}
    void preamble(gen_context &c) const { expr.preamble(c); }
    void params(gen_context &c) const { expr.params(c); }
    void local_init(gen_context &c) const { expr.local_init(c); }
    void emit(gen_context &c) const { c.src << "( (" << type_name<T>() << ")( "; expr.emit(c); c.src << " ) )"; }
    void set_args(arg_context &a) const { expr.set_args(a); }
    void get_props(prop_context &p) const { expr.get_props(p); }
};
template <class T, class E> struct expr_kind<casted_expression<T, E>>
    : std::integral_constant<int, expr_kind<E>::value == 0 ? 0 : -1> {};
template <class T, class E> struct mv_dim<casted_expression<T, E>> : mv_dim<E> {};
template <size_t I, class T, class E>
struct component_of<I, casted_expression<T, E>, typename std::enable_if<(mv_dim<E>::value > 0)>::type> {
    typedef casted_expression<T, typename component_of<I, E>::type> type;
    static type get(const casted_expression<T, E> &e) { return type(component_of<I, E>::get(e.expr)); }
};
} // namespace detail

/// cast<double>(5), cast<int>(x / 2) ...
template <class T, class Expr>
typename std::enable_if<detail::is_operand<Expr>::value, const detail::casted_expression<T, detail::as_expr_t<Expr>>>::type
cast(const Expr &expr) {
    return detail::casted_expression<T, detail::as_expr_t<Expr>>(detail::as_expr<Expr>::get(expr));
}

} // namespace vex
#endif
