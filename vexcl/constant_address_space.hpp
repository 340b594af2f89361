#ifndef VEXCL_CONSTANT_ADDRESS_SPACE_HPP
#define VEXCL_CONSTANT_ADDRESS_SPACE_HPP
// vex::constant(v): reads of the wrapped vector go through the read-only path
// (reference: vexcl/constant_address_space.hpp:40-183 -- OpenCL's __constant address space).
// gfx950 has no separate constant memory; what corresponds to it is a `const T * __restrict__`
// kernel parameter: the compiler may then use scalar loads (one request per wave, through the
// scalar cache) wherever the index is uniform across the wave, and knows the data cannot alias
// the kernel's stores.  Typical use: small lookup tables indexed through vex::permutation.
#include "vector.hpp"

namespace vex {

template <class T>
struct constant_vector : detail::expression_base {
    typedef T value_type;
    const vector<T> *v;
    explicit constant_vector(const vector<T> &vec) : v(&vec) {}
    void preamble(detail::gen_context &c) const { c.next(); }
    void params(detail::gen_context &c) const { c.src.template parameter<constant_ptr<T>>(c.next()); }
    void local_init(detail::gen_context &c) const { c.next(); }
    void emit(detail::gen_context &c) const { c.src << c.next() << "[idx]"; }
    void set_args(detail::arg_context &a) const { a.next(); a.krn.push_arg((*v)(a.device)); }
    void get_props(detail::prop_context &p) const {
        if (p.empty()) { p.queue = v->queue_list(); p.part = v->partition(); p.size = v->size(); }
    }
};

/// Uses the read-only path for access to the wrapped vector.
template <class T>
constant_vector<T> constant(const vector<T> &v) { return constant_vector<T>(v); }

} // namespace vex
#endif
