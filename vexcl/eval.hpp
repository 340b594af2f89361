#ifndef VEXCL_EVAL_HPP
#define VEXCL_EVAL_HPP
// vex::eval(expr): evaluates an expression for its side effects, no lhs
// (reference: vexcl/eval.hpp:39-108, kernel `vexcl_eval_kernel`).
#include "operations.hpp"

namespace vex {

template <class Expr>
void eval(const Expr &expr_, const std::vector<backend::command_queue> &queue, const std::vector<size_t> &part) {
    using namespace detail;
    typedef as_expr_t<Expr> E;
    const E &expr = as_expr<Expr>::get(expr_);
    static kernel_cache cache;
    for (unsigned d = 0; d < queue.size(); ++d) {
        size_t psize = part[d + 1] - part[d];
        if (!psize) continue;
        auto kernel = cache.find(queue[d]);
        if (kernel == cache.end()) {
            backend::source_generator source(queue[d]);
            { gen_context c(source, queue[d]); expr.preamble(c); }
            source.begin_kernel("vexcl_eval_kernel");
            source.begin_kernel_parameters();
            source.template parameter<size_t>("n");
            { gen_context c(source, queue[d]); expr.params(c); }
            source.end_kernel_parameters();
            source.grid_stride_loop().open("{");
            { gen_context c(source, queue[d]); expr.local_init(c); }
            source.new_line();
            { gen_context c(source, queue[d]); expr.emit(c); source << ";"; }
            source.close("}");
            source.end_kernel();
            kernel = cache.insert(queue[d], backend::kernel(queue[d], source.str(), "vexcl_eval_kernel"));
        }
        backend::kernel &krn = kernel->second;
        krn.push_arg(psize);
        arg_context a(krn, d, part[d]);
        expr.set_args(a);
        krn.config_streaming(queue[d], psize);
        krn(queue[d]);
    }
}

template <class Expr>
void eval(const Expr &expr) {
    std::vector<backend::command_queue> queue; std::vector<size_t> part; size_t size;
    get_expression_properties(expr, queue, part, size);
    precondition(!queue.empty() && !part.empty(), "Can not determine expression size and queue list");
    eval(expr, queue, part);
}

} // namespace vex
#endif
