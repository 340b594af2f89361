#ifndef VEXCL_SVM_VECTOR_HPP
#define VEXCL_SVM_VECTOR_HPP
// vex::svm_vector<T>: a vector in memory the host and the device address alike
// (reference: vexcl/svm_vector.hpp:40-221 -- OpenCL 2 shared virtual memory; CUDA backend:
// cuMemAllocManaged, backend/cuda/svm_vector.hpp:50-98).  Here: hipMallocManaged through
// vexhip_malloc_managed.  One queue, one partition.
//
//   vex::svm_vector<int> x(queue, n);
//   { auto p = x.map(vex::backend::MAP_WRITE); for (...) p[i] = i; }   // host writes in place
//   y = x * 2;  x = y / 2;                                              // terminal and lvalue of fused kernels
//   y = f(vex::element_index(), vex::raw_pointer(x));
#include "vector.hpp"
#include "vector_pointer.hpp"

namespace vex {

namespace backend {
typedef int map_flags;
static const map_flags MAP_READ = 1, MAP_WRITE = 2;
}

template <class T> class svm_vector;

namespace detail {
template <class T>
struct svm_ref : expression_base {
    typedef T value_type;
    const svm_vector<T> *v;
    svm_ref(const svm_vector<T> &vec) : v(&vec) {}
    void preamble(gen_context &c) const { c.next(); }
    void params(gen_context &c) const { c.src.template parameter<global_ptr<T>>(c.next()); }
    void local_init(gen_context &c) const { c.next(); }
    void emit(gen_context &c) const { c.src << c.next() << "[idx]"; }
    void set_args(arg_context &a) const { a.next(); a.krn.push_arg(v->get()); }
    void get_props(prop_context &p) const {
        if (p.empty()) { p.queue.assign(1, v->queue()); p.part = {0, v->size()}; p.size = v->size(); }
    }
};

/// raw_pointer(svm_vector): the pointer itself as a kernel parameter.
template <class T>
struct svm_pointer : expression_base {
    typedef T *value_type;
    const svm_vector<T> *v;
    explicit svm_pointer(const svm_vector<T> &vec) : v(&vec) {}
    void preamble(gen_context &c) const { c.next(); }
    void params(gen_context &c) const { c.src.template parameter<global_ptr<T>>(c.next()); }
    void local_init(gen_context &c) const { c.next(); }
    void emit(gen_context &c) const { c.src << c.next(); }
    void set_args(arg_context &a) const { a.next(); a.krn.push_arg(v->get()); }
    void get_props(prop_context &) const {}
};
} // namespace detail

template <class T>
class svm_vector : public detail::expression_base {
    public:
        typedef T value_type;
        typedef T *mapped_pointer;
        typedef detail::svm_ref<T> expr_ref_type;

        svm_vector(const backend::command_queue &q, size_t n) : n(n), q(q), p(nullptr) {
            void *ptr = nullptr;
            backend::check(vexhip_malloc_managed(q.device_ordinal(), n * sizeof(T), &ptr));
            p = static_cast<T *>(ptr);
        }
        ~svm_vector() { if (p) { q.finish(); vexhip_free(q.device_ordinal(), p); } }

        size_t size() const { return n; }
        T *get() const { return p; }
        const backend::command_queue &queue() const { return q; }

        /// Host access in place: waits for the kernels in flight on the queue, then hands out the pointer.
        mapped_pointer map(backend::map_flags) { q.finish(); return p; }

        const svm_vector &operator=(const svm_vector &other) {
            assign<assign::SET>(detail::svm_ref<T>(other));
            return *this;
        }

#define VEXCL_SVM_ASSIGNMENT(op, tag)                                                                    \
        template <class Expr>                                                                            \
        typename std::enable_if<detail::is_operand<Expr>::value, const svm_vector &>::type               \
        operator op(const Expr &expr) { assign<assign::tag>(detail::as_expr<Expr>::get(expr)); return *this; }
        VEXCL_SVM_ASSIGNMENT(=, SET)  VEXCL_SVM_ASSIGNMENT(+=, ADD) VEXCL_SVM_ASSIGNMENT(-=, SUB)
        VEXCL_SVM_ASSIGNMENT(*=, MUL) VEXCL_SVM_ASSIGNMENT(/=, DIV) VEXCL_SVM_ASSIGNMENT(%=, MOD)
        VEXCL_SVM_ASSIGNMENT(&=, AND) VEXCL_SVM_ASSIGNMENT(|=, OR)  VEXCL_SVM_ASSIGNMENT(^=, XOR)
        VEXCL_SVM_ASSIGNMENT(<<=, LSH) VEXCL_SVM_ASSIGNMENT(>>=, RSH)
#undef VEXCL_SVM_ASSIGNMENT

    private:
        size_t n;
        backend::command_queue q;
        T *p;
        svm_vector(const svm_vector &);

        template <class OP, class E>
        void assign(const E &e) {
            std::vector<backend::command_queue> ql(1, q);
            std::vector<size_t> part = {0, n};
            detail::assign_expression<OP>(detail::svm_ref<T>(*this), e, ql, part);
        }
};

template <class T>
detail::svm_pointer<T> raw_pointer(const svm_vector<T> &v) { return detail::svm_pointer<T>(v); }

} // namespace vex
#endif
