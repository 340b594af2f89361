#ifndef VEXCL_VECTOR_POINTER_HPP
#define VEXCL_VECTOR_POINTER_HPP
// vex::raw_pointer(v): passes the device pointer of a single-partition vector
// into a kernel as "T * prm_k" (reference: vexcl/vector_pointer.hpp:146-160).
#include "vector.hpp"

namespace vex {
namespace detail {
template <class T>
struct vector_pointer : expression_base {
    typedef T *value_type;
    const vector<T> *v;
    explicit vector_pointer(const vector<T> &vec) : v(&vec) {}
    void preamble(gen_context &c) const { c.next(); }
    void params(gen_context &c) const { c.src.template parameter<global_ptr<T>>(c.next()); }
    void local_init(gen_context &c) const { c.next(); }
    void emit(gen_context &c) const { c.src << c.next(); }
    void set_args(arg_context &a) const { a.next(); a.krn.push_arg((*v)(a.device)); }
    void get_props(prop_context &) const {}

    /// p[i] with i an expression: the element i positions from the start (vector_pointer.hpp:100-140).
    template <class I>
    typename std::enable_if<is_operand<I>::value, const deref_expr<binary_expr<tag::plus, vector_pointer, as_expr_t<I>>>>::type
    operator[](const I &i) const {
        typedef binary_expr<tag::plus, vector_pointer, as_expr_t<I>> sum;
        return deref_expr<sum>(sum(*this, as_expr<I>::get(i)));
    }
};
template <class T> struct expr_kind<vector_pointer<T>> : std::integral_constant<int, 0> {};

/// constant_pointer(v): the same pointer declared `const T * __restrict__` (vector_pointer.hpp:163-181 of the reference:
/// a pointer into the constant address space).
template <class T>
struct constant_vector_pointer : expression_base {
    typedef const T *value_type;
    const vector<T> *v;
    explicit constant_vector_pointer(const vector<T> &vec) : v(&vec) {}
    void preamble(gen_context &c) const { c.next(); }
    void params(gen_context &c) const { c.src.template parameter<constant_ptr<T>>(c.next()); }
    void local_init(gen_context &c) const { c.next(); }
    void emit(gen_context &c) const { c.src << c.next(); }
    void set_args(arg_context &a) const { a.next(); a.krn.push_arg((*v)(a.device)); }
    void get_props(prop_context &) const {}
    template <class I>
    typename std::enable_if<is_operand<I>::value, const deref_expr<binary_expr<tag::plus, constant_vector_pointer, as_expr_t<I>>>>::type
    operator[](const I &i) const {
        typedef binary_expr<tag::plus, constant_vector_pointer, as_expr_t<I>> sum;
        return deref_expr<sum>(sum(*this, as_expr<I>::get(i)));
    }
};
template <class T> struct expr_kind<constant_vector_pointer<T>> : std::integral_constant<int, 0> {};
} // namespace detail

template <class T>
detail::constant_vector_pointer<T> constant_pointer(const vector<T> &v) {
    precondition(v.nparts() == 1, "constant_pointer is not supported for multi-device contexts");
    return detail::constant_vector_pointer<T>(v);
}

template <class T>
detail::vector_pointer<T> raw_pointer(const vector<T> &v) {
    precondition(v.nparts() == 1, "raw_pointer is not supported for multi-device contexts");
    return detail::vector_pointer<T>(v);
}
} // namespace vex
#endif
