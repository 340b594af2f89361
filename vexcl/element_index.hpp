#ifndef VEXCL_ELEMENT_INDEX_HPP
#define VEXCL_ELEMENT_INDEX_HPP
// vex::element_index(offset, length): the global index of the element being
// processed (reference: vexcl/element_index.hpp:38-113).  Prints
// "( prm_k + idx )"; the argument is offset + start of the device's partition.
#include "operations.hpp"

namespace vex {
namespace detail {
struct elem_index : expression_base {
    typedef size_t value_type;
    size_t offset, length;
    elem_index(size_t offset = 0, size_t length = 0) : offset(offset), length(length) {}
    void preamble(gen_context &c) const { c.next(); }
    void params(gen_context &c) const { c.src.template parameter<size_t>(c.next()); }
    void local_init(gen_context &c) const { c.next(); }
    void emit(gen_context &c) const { c.src << "( " << c.next() << " + idx )"; }
    void set_args(arg_context &a) const { a.next(); a.krn.push_arg(offset + a.offset); }
    void get_props(prop_context &p) const { if (p.size == 0 && length) p.size = length; }
};
} // namespace detail
using detail::elem_index;

/// Index of the current element, shifted by offset; length gives a size to
/// otherwise size-less expressions (element_index.hpp:52-66).
inline const detail::elem_index element_index(size_t offset = 0, size_t length = 0) {
    return detail::elem_index(offset, length);
}
} // namespace vex
#endif
