#ifndef VEXCL_TAGGED_TERMINAL_HPP
#define VEXCL_TAGGED_TERMINAL_HPP
// vex::tag<N>(x): terminals carrying the same tag share ONE kernel parameter
// (reference: vexcl/tagged_terminal.hpp:51-262).  `auto ta = vex::tag<1>(a);
// ta = alpha * ta + b;` reads and writes a through a single pointer
// (examples/benchmark.cpp:100-107).
#include "operations.hpp"
#include "vector.hpp"

namespace vex {

template <size_t Tag, class Term>
struct tagged_terminal : detail::expression_base {
    typedef typename Term::value_type value_type;
    Term term;
    explicit tagged_terminal(const Term &t) : term(t) {}

    static std::string prefix() { return "prm_tag_" + std::to_string(Tag); }
    template <class F> static void once(std::set<std::string> &seen, const char *pass, F &&f) {
        std::string key = std::string(pass) + prefix();
        if (seen.count(key)) return;
        seen.insert(key);
        f();
    }
    void preamble(detail::gen_context &c) const {
        once(c.seen, "pre:", [&] { detail::gen_context i(c, prefix()); term.preamble(i); });
    }
    void params(detail::gen_context &c) const {
        once(c.seen, "prm:", [&] { detail::gen_context i(c, prefix()); term.params(i); });
    }
    void local_init(detail::gen_context &c) const {
        once(c.seen, "loc:", [&] { detail::gen_context i(c, prefix()); term.local_init(i); });
    }
    void emit(detail::gen_context &c) const { detail::gen_context i(c, prefix()); term.emit(i); }
    void set_args(detail::arg_context &a) const {
        once(a.seen, "arg:", [&] { detail::arg_context i(a); term.set_args(i); });
    }
    void get_props(detail::prop_context &p) const { term.get_props(p); }

    // lvalue (tagged_terminal.hpp:248-262)
#define VEXCL_TAGGED_ASSIGN(op, tag)                                                                    \
    template <class Expr>                                                                               \
    typename std::enable_if<detail::is_operand<Expr>::value, const tagged_terminal &>::type             \
    operator op(const Expr &expr) const {                                                               \
        detail::prop_context p; term.get_props(p);                                                      \
        detail::assign_expression<assign::tag>(*this, detail::as_expr<Expr>::get(expr), p.queue, p.part); \
        return *this;                                                                                   \
    }
    VEXCL_TAGGED_ASSIGN(=, SET)   VEXCL_TAGGED_ASSIGN(+=, ADD)  VEXCL_TAGGED_ASSIGN(-=, SUB)
    VEXCL_TAGGED_ASSIGN(*=, MUL)  VEXCL_TAGGED_ASSIGN(/=, DIV)  VEXCL_TAGGED_ASSIGN(%=, MOD)
    VEXCL_TAGGED_ASSIGN(&=, AND)  VEXCL_TAGGED_ASSIGN(|=, OR)   VEXCL_TAGGED_ASSIGN(^=, XOR)
    VEXCL_TAGGED_ASSIGN(<<=, LSH) VEXCL_TAGGED_ASSIGN(>>=, RSH)
#undef VEXCL_TAGGED_ASSIGN
    const tagged_terminal &operator=(const tagged_terminal &o) const {
        detail::prop_context p; term.get_props(p);
        detail::assign_expression<assign::SET>(*this, o, p.queue, p.part);
        return *this;
    }
};

/// Tags a terminal (tagged_terminal.hpp:51-80).
template <size_t Tag, class Expr>
typename std::enable_if<detail::is_operand<Expr>::value, const tagged_terminal<Tag, detail::as_expr_t<Expr>>>::type
tag(const Expr &e) { return tagged_terminal<Tag, detail::as_expr_t<Expr>>(detail::as_expr<Expr>::get(e)); }

} // namespace vex
#endif
