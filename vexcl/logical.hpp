#ifndef VEXCL_LOGICAL_HPP
#define VEXCL_LOGICAL_HPP
// vex::any_of / vex::all_of (reference: vexcl/logical.hpp:45-237; tests/logical.cpp):
// does any / every element of a vector expression evaluate to true?  Both are one
// fused reduction of `expr != 0` (MAX resp. MIN of the 0/1 flags) on the GPU; the
// reference uses a dedicated early-exit kernel.
#include "reductor.hpp"
#include "cast.hpp"

namespace vex {

struct any_of {
    explicit any_of(const std::vector<backend::command_queue> &queue = current_context().queue()) : rdc(queue) {}
    template <class Expr> bool operator()(const Expr &expr) const { return rdc(cast<int>(expr != 0)) != 0; }
    private: Reductor<int, MAX> rdc;
};

struct all_of {
    explicit all_of(const std::vector<backend::command_queue> &queue = current_context().queue()) : rdc(queue) {}
    template <class Expr> bool operator()(const Expr &expr) const { return rdc(cast<int>(expr != 0)) != 0; }
    private: Reductor<int, MIN> rdc;
};

} // namespace vex
#endif
