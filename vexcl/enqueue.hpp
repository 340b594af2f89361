#ifndef VEXCL_ENQUEUE_HPP
#define VEXCL_ENQUEUE_HPP
// vex::enqueue(queues, lhs) = expr: the assignment runs on the given queues (HIP streams) instead of
// the ones the left-hand side was created with (reference: vexcl/enqueue.hpp:38-160; tests/events.cpp).
// Together with backend::enqueue_marker / enqueue_barrier this is how independent work overlaps:
//     x = f(a);                                                        // stream 1
//     enqueue_barrier(q2[0], {enqueue_marker(q1[0])});
//     enqueue(q2, y) = g(x);                                           // stream 2, after the marker
#include "operations.hpp"
#include "vector.hpp"
#include "multivector.hpp"

namespace vex {

template <class T>
class enqueued_vector {
    public:
        enqueued_vector(vector<T> &lhs, const std::vector<backend::command_queue> &q) : lhs(lhs), q(q) {
            precondition(q.size() == lhs.queue_list().size(), "enqueue: as many queues as the vector has partitions are expected");
        }
#define VEXCL_ENQUEUE_ASSIGN(op, tag)                                                                     \
        template <class Expr>                                                                             \
        typename std::enable_if<detail::is_operand<Expr>::value, const vector<T> &>::type                 \
        operator op(const Expr &expr) const {                                                             \
            detail::assign_any<assign::tag>(detail::vector_ref<T>(lhs), lhs, detail::as_expr<Expr>::get(expr), q, lhs.partition()); \
            return lhs;                                                                                   \
        }
        VEXCL_ENQUEUE_ASSIGN(=, SET)   VEXCL_ENQUEUE_ASSIGN(+=, ADD)  VEXCL_ENQUEUE_ASSIGN(-=, SUB)  VEXCL_ENQUEUE_ASSIGN(*=, MUL)
        VEXCL_ENQUEUE_ASSIGN(/=, DIV)  VEXCL_ENQUEUE_ASSIGN(%=, MOD)  VEXCL_ENQUEUE_ASSIGN(&=, AND)  VEXCL_ENQUEUE_ASSIGN(|=, OR)
        VEXCL_ENQUEUE_ASSIGN(^=, XOR)  VEXCL_ENQUEUE_ASSIGN(<<=, LSH) VEXCL_ENQUEUE_ASSIGN(>>=, RSH)
#undef VEXCL_ENQUEUE_ASSIGN
    private:
        vector<T> &lhs;
        std::vector<backend::command_queue> q;
};

template <class T, size_t N>
class enqueued_multivector {
    public:
        enqueued_multivector(multivector<T, N> &lhs, const std::vector<backend::command_queue> &q) : lhs(lhs), q(q) {}
#define VEXCL_ENQUEUE_ASSIGN(op, tag)                                                                     \
        template <class Expr>                                                                             \
        typename std::enable_if<detail::is_operand<Expr>::value, const multivector<T, N> &>::type         \
        operator op(const Expr &expr) const {                                                             \
            detail::assign_multi<assign::tag>(lhs.components(), detail::as_expr<Expr>::get(expr), std::make_index_sequence<N>(), &q); \
            return lhs;                                                                                   \
        }
        VEXCL_ENQUEUE_ASSIGN(=, SET)   VEXCL_ENQUEUE_ASSIGN(+=, ADD)  VEXCL_ENQUEUE_ASSIGN(-=, SUB)  VEXCL_ENQUEUE_ASSIGN(*=, MUL)
        VEXCL_ENQUEUE_ASSIGN(/=, DIV)  VEXCL_ENQUEUE_ASSIGN(%=, MOD)  VEXCL_ENQUEUE_ASSIGN(&=, AND)  VEXCL_ENQUEUE_ASSIGN(|=, OR)
        VEXCL_ENQUEUE_ASSIGN(^=, XOR)  VEXCL_ENQUEUE_ASSIGN(<<=, LSH) VEXCL_ENQUEUE_ASSIGN(>>=, RSH)
#undef VEXCL_ENQUEUE_ASSIGN
    private:
        multivector<T, N> &lhs;
        std::vector<backend::command_queue> q;
};

template <class T>
enqueued_vector<T> enqueue(const std::vector<backend::command_queue> &q, vector<T> &lhs) { return enqueued_vector<T>(lhs, q); }
template <class T, size_t N>
enqueued_multivector<T, N> enqueue(const std::vector<backend::command_queue> &q, multivector<T, N> &lhs) { return enqueued_multivector<T, N>(lhs, q); }

} // namespace vex
#endif
