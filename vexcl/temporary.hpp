#ifndef VEXCL_TEMPORARY_HPP
#define VEXCL_TEMPORARY_HPP
// vex::make_temp<Tag>(expr): a named intermediate that is computed ONCE per element and
// may be used several times in the expression (reference: vexcl/temporary.hpp:54-270;
// tests/temporary.cpp).  Generated code:  `double temp_1 = <expr>;` before the statement,
// `temp_1` wherever it is used; the terminals of <expr> are declared and bound once.
// Temporaries nest, work inside reductions and inside multi-expressions (one temporary
// per component when the expression holds multivectors).
#include "operations.hpp"

namespace vex {
namespace detail {

template <size_t Tag, class T, class E>
struct temporary : expression_base {
    typedef T value_type;
    E expr;
    explicit temporary(const E &e) : expr(e) {}

    static std::string name() { return "temp_" + std::to_string(Tag); }
    static std::string prefix() { return "prm_temp_" + std::to_string(Tag); }
    template <class F> static void once(std::set<std::string> &seen, const char *pass, F &&f) {
        const std::string key = std::string(pass) + name();
        if (seen.count(key)) return;
        seen.insert(key);
        f();
    }
    void preamble(gen_context &c) const { once(c.seen, "pre:", [&] { gen_context i(c, prefix()); expr.preamble(i); }); }
    void params(gen_context &c) const { once(c.seen, "prm:", [&] { gen_context i(c, prefix()); expr.params(i); }); }
    void local_init(gen_context &c) const {
        once(c.seen, "loc:", [&] {
            { gen_context i(c, prefix()); expr.local_init(i); }       // inner temporaries are declared first
            c.src.new_line() << type_name<T>() << " " << name() << " = ";
            { gen_context i(c, prefix()); expr.emit(i); }
            c.src << ";";
        });
    }
    void emit(gen_context &c) const { c.src << name(); }
    void set_args(arg_context &a) const { once(a.seen, "arg:", [&] { arg_context i(a); expr.set_args(i); }); }
    void get_props(prop_context &p) const { expr.get_props(p); }
};
template <size_t Tag, class T, class E> struct expr_kind<temporary<Tag, T, E>>
    : std::integral_constant<int, expr_kind<E>::value == 0 ? 0 : -1> {};
template <size_t Tag, class T, class E> struct mv_dim<temporary<Tag, T, E>> : mv_dim<E> {};
// a temporary over multivectors: component I gets its own name
template <size_t I, size_t Tag, class T, class E>
struct component_of<I, temporary<Tag, T, E>, typename std::enable_if<(mv_dim<E>::value > 0)>::type> {
    typedef temporary<(size_t(1) << 20) + Tag * 64 + I, T, typename component_of<I, E>::type> type;
    static type get(const temporary<Tag, T, E> &t) { return type(component_of<I, E>::get(t.expr)); }
};

} // namespace detail

/// The value type is that of the expression ...
template <size_t Tag, class Expr>
typename std::enable_if<detail::is_operand<Expr>::value,
    const detail::temporary<Tag, typename detail::as_expr_t<Expr>::value_type, detail::as_expr_t<Expr>>>::type
make_temp(const Expr &expr) {
    return detail::temporary<Tag, typename detail::as_expr_t<Expr>::value_type, detail::as_expr_t<Expr>>(detail::as_expr<Expr>::get(expr));
}
/// ... or given explicitly: make_temp<1, double>(expr).
template <size_t Tag, class T, class Expr>
typename std::enable_if<detail::is_operand<Expr>::value, const detail::temporary<Tag, T, detail::as_expr_t<Expr>>>::type
make_temp(const Expr &expr) {
    return detail::temporary<Tag, T, detail::as_expr_t<Expr>>(detail::as_expr<Expr>::get(expr));
}

} // namespace vex
#endif
