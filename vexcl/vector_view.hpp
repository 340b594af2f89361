#ifndef VEXCL_VECTOR_VIEW_HPP
#define VEXCL_VECTOR_VIEW_HPP
// vex::permutation(index_expr)(base): gather / scatter view of a vector
// (reference: vexcl/vector_view.hpp:602-700).  The view's
// value is base[ index_expr ]; it is an lvalue too.  Single-device only, size and
// queues come from the index expression.  The n-D slicing / reshape / reduce
// views of the reference are out of scope.
#include "vector.hpp"

namespace vex {

template <class T, class Index>
struct permutation_view : detail::expression_base {
    typedef T value_type;
    const vector<T> *base; Index index;
    permutation_view(const vector<T> &b, const Index &i) : base(&b), index(i) {
        precondition(b.nparts() == 1, "permutation is only supported for single-device vectors");
    }
    void preamble(detail::gen_context &c) const { c.next(); index.preamble(c); }
    void params(detail::gen_context &c) const { c.src.template parameter<global_ptr<T>>(c.next()); index.params(c); }
    void local_init(detail::gen_context &c) const { c.next(); index.local_init(c); }
    void emit(detail::gen_context &c) const { c.src << c.next() << "[ "; index.emit(c); c.src << " ]"; }
    void set_args(detail::arg_context &a) const { a.next(); a.krn.push_arg((*base)(a.device)); index.set_args(a); }
    void get_props(detail::prop_context &p) const {
        index.get_props(p);
        if (p.queue.empty()) { p.queue = base->queue_list(); }
        if (p.part.empty() && p.size) { p.part = {0, p.size}; }
    }

#define VEXCL_VIEW_ASSIGN(op, tag)                                                                      \
    template <class Expr>                                                                               \
    typename std::enable_if<detail::is_operand<Expr>::value, const permutation_view &>::type            \
    operator op(const Expr &expr) const {                                                               \
        detail::prop_context p; get_props(p);                                                           \
        detail::assign_expression<assign::tag>(*this, detail::as_expr<Expr>::get(expr), p.queue, p.part); \
        return *this;                                                                                   \
    }
    VEXCL_VIEW_ASSIGN(=, SET) VEXCL_VIEW_ASSIGN(+=, ADD) VEXCL_VIEW_ASSIGN(-=, SUB)
    VEXCL_VIEW_ASSIGN(*=, MUL) VEXCL_VIEW_ASSIGN(/=, DIV)
#undef VEXCL_VIEW_ASSIGN
};

template <class Index>
struct permutation_builder {
    Index index;
    template <class T>
    permutation_view<T, Index> operator()(const vector<T> &base) const { return permutation_view<T, Index>(base, index); }
};

/// permutation(index_expression)(vector)  (vector_view.hpp:602-700)
template <class Expr>
typename std::enable_if<detail::is_expr<Expr>::value, permutation_builder<detail::as_expr_t<Expr>>>::type
permutation(const Expr &index) { return permutation_builder<detail::as_expr_t<Expr>>{detail::as_expr<Expr>::get(index)}; }

} // namespace vex
#endif
