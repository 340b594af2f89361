#ifndef VEXCL_VECTOR_HPP
#define VEXCL_VECTOR_HPP
// vex::vector<T>: a device vector partitioned contiguously across the queues
// of a context, one segment per GPU (reference: vexcl/vector.hpp:79-190
// partitioning, :220-935 the class, :948-992 terminal traits, :998-1228 copy).
#include <algorithm>
#include <functional>
#include <iomanip>
#include <iostream>
#include <iterator>
#include <map>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <vector>

#include "backend.hpp"
#include "devlist.hpp"
#include "profiler.hpp"
#include "operations.hpp"

namespace vex {

// ---- partitioning (vector.hpp:79-190) -------------------------------------------
/// Equal weights: equal partitioning (vector.hpp:79-81).
inline double equal_weights(const backend::command_queue &) { return 1; }

template <bool dummy = true>
struct partitioning_scheme {
    typedef std::function<double(const backend::command_queue &)> weight_function;

    static void set(weight_function f) {
        std::lock_guard<std::mutex> lock(mx());
        if (!is_set()) { weight() = f; is_set() = true; }
        else std::cerr << "Warning: device weighting function is already set and will be left as is." << std::endl;
    }

    static std::vector<size_t> get(size_t n, const std::vector<backend::command_queue> &queue) {
        {
            std::lock_guard<std::mutex> lock(mx());
            // Identical GPUs in one node: the reference's timing probe
            // (device_vector_perf, vector.hpp:1237-1255) would return equal
            // weights up to noise; equal weights keep row splits deterministic.
            if (!is_set()) { weight() = equal_weights; is_set() = true; }
        }
        std::vector<size_t> part;
        part.reserve(queue.size() + 1);
        part.push_back(0);
        if (queue.size() > 1) {
            std::vector<double> cumsum(1, 0.0);
            for (const auto &q : queue) cumsum.push_back(cumsum.back() + weight()(q));
            for (unsigned d = 1; d < queue.size(); ++d)
                part.push_back(std::min(n, alignup(static_cast<size_t>(n * cumsum[d] / cumsum.back()))));
        }
        part.push_back(n);
        return part;
    }
    private:
        static bool &is_set() { static bool v = false; return v; }
        static weight_function &weight() { static weight_function w; return w; }
        static std::mutex &mx() { static std::mutex m; return m; }
};

/// Sets the partitioning weight function once (vector.hpp:178-183).
inline void set_partitioning(std::function<double(const backend::command_queue &)> f) { partitioning_scheme<>::set(f); }

/// Partition boundaries of an n-element vector over the queues (vector.hpp:186-190).
inline std::vector<size_t> partition(size_t n, const std::vector<backend::command_queue> &queue) {
    return partitioning_scheme<>::get(n, queue);
}

template <class T> class vector;

namespace detail {

/// How a vex::vector appears inside an expression: parameter "T * prm_k",
/// text "prm_k[idx]", argument = the device's buffer (vector.hpp:948-992).
template <class T>
struct vector_ref : expression_base {
    typedef T value_type;
    const vector<T> *v;
    vector_ref(const vector<T> &vec) : v(&vec) {}
    void preamble(gen_context &c) const { c.next(); }
    void params(gen_context &c) const { c.src.template parameter<global_ptr<T>>(c.next()); }
    void local_init(gen_context &c) const { c.next(); }
    void emit(gen_context &c) const { c.src << c.next() << "[idx]"; }
    void set_args(arg_context &a) const { a.next(); a.krn.push_arg((*v)(a.device)); }
    void get_props(prop_context &p) const {
        if (p.empty()) { p.queue = v->queue_list(); p.part = v->partition(); p.size = v->size(); }
        else p.also(v->queue_list().size(), v->size());
    }
};

} // namespace detail

template <class T>
class vector : public detail::expression_base {
    public:
        typedef T value_type;
        typedef size_t size_type;
        typedef detail::vector_ref<T> expr_ref_type;

        /// Proxy of one element in device memory (vector.hpp:232-270).
        class element {
            public:
                operator T() const { T v; buf.read(queue, index, 1, &v, true); return v; }
                T operator=(T v) { buf.write(queue, index, 1, &v, true); return v; }
                T operator=(const element &o) { return *this = static_cast<T>(o); }
                friend void swap(element &&a, element &&b) { T t = a; a = static_cast<T>(b); b = t; }
            private:
                element(const backend::command_queue &q, const backend::device_vector<T> &b, size_t i)
                    : queue(q), buf(b), index(i) {}
                const backend::command_queue &queue;
                const backend::device_vector<T> &buf;
                size_t index;
                friend class vector;
        };

        /// Random access iterator over device elements (vector.hpp:274-345).
        template <class V, class E>
        class iterator_type {
            public:
                typedef std::random_access_iterator_tag iterator_category;
                typedef T value_type; typedef ptrdiff_t difference_type; typedef E *pointer; typedef E reference;
                static const bool device_iterator = true;

                iterator_type() : vec(0), pos(0) {}
                iterator_type(V &v, size_t p) : vec(&v), pos(p) {}
                E operator*() const { return (*vec)[pos]; }
                iterator_type &operator++() { ++pos; return *this; }
                iterator_type operator++(int) { iterator_type t(*this); ++pos; return t; }
                iterator_type &operator--() { --pos; return *this; }
                iterator_type &operator+=(ptrdiff_t d) { pos += d; return *this; }
                iterator_type &operator-=(ptrdiff_t d) { pos -= d; return *this; }
                iterator_type operator+(ptrdiff_t d) const { return iterator_type(*vec, pos + d); }
                iterator_type operator-(ptrdiff_t d) const { return iterator_type(*vec, pos - d); }
                ptrdiff_t operator-(const iterator_type &o) const { return (ptrdiff_t)pos - (ptrdiff_t)o.pos; }
                bool operator==(const iterator_type &o) const { return pos == o.pos; }
                bool operator!=(const iterator_type &o) const { return pos != o.pos; }
                bool operator<(const iterator_type &o) const { return pos < o.pos; }
                E operator[](ptrdiff_t d) const { return (*vec)[pos + d]; }
                V *vec; size_t pos;
        };
        typedef iterator_type<vector, element> iterator;
        typedef iterator_type<const vector, const element> const_iterator;

        // ---- construction (vector.hpp:352-470) --------------------------------
        vector() {}

        /// Wraps a native buffer (vector.hpp:375-386).
        vector(const backend::command_queue &q, const backend::device_vector<T> &buffer, size_t size = 0)
            : queue(1, q), part(2), buf(1, buffer) { part[0] = 0; part[1] = size ? size : buffer.size(); }

        vector(const std::vector<backend::command_queue> &queue, size_t size, const T *host = 0,
               backend::mem_flags flags = backend::MEM_READ_WRITE)
            : queue(queue), part(vex::partition(size, queue)), buf(queue.size())
        { if (size) allocate_buffers(flags, host); }

        explicit vector(size_t size)
            : queue(current_context().queue()), part(vex::partition(size, queue)), buf(queue.size())
        { if (size) allocate_buffers(backend::MEM_READ_WRITE, 0); }

        vector(const std::vector<backend::command_queue> &queue, const std::vector<T> &host,
               backend::mem_flags flags = backend::MEM_READ_WRITE)
            : queue(queue), part(vex::partition(host.size(), queue)), buf(queue.size())
        { if (!host.empty()) allocate_buffers(flags, host.data()); }

        vector(const std::vector<T> &host, backend::mem_flags flags = backend::MEM_READ_WRITE)
            : queue(current_context().queue()), part(vex::partition(host.size(), queue)), buf(queue.size())
        { if (!host.empty()) allocate_buffers(flags, host.data()); }

        /// Deep copy (vector.hpp:365-373).
        vector(const vector &v) : detail::expression_base(), queue(v.queue), part(v.part), buf(v.queue.size()) {
#ifdef VEXCL_SHOW_COPIES
            std::cout << "Copying vex::vector<" << type_name<T>() << "> of size " << v.size() << std::endl;
#endif
            if (size()) { allocate_buffers(backend::MEM_READ_WRITE, 0); *this = v; }
        }
        vector(vector &&v) noexcept { swap(v); }

        /// From an expression (vector.hpp:438-470): takes size and queues from it.
        template <class Expr, class = typename std::enable_if<
            detail::is_expr<Expr>::value && !std::is_same<typename std::decay<Expr>::type, vector>::value>::type>
        vector(const Expr &expr) {
            size_t n;
            get_expression_properties(expr, queue, part, n);
            precondition(!queue.empty() && !part.empty(), "Can not determine expression size and queue list");
            buf.resize(queue.size());
            allocate_buffers(backend::MEM_READ_WRITE, 0);
            *this = expr;
        }

        // ---- resize / swap ------------------------------------------------------
        void resize(const vector &v, backend::mem_flags flags = backend::MEM_READ_WRITE) { vector(v.queue, v.size(), 0, flags).swap(*this); *this = v; }
        void resize(const std::vector<backend::command_queue> &q, size_t size, const T *host = 0,
                    backend::mem_flags flags = backend::MEM_READ_WRITE) { vector(q, size, host, flags).swap(*this); }
        void resize(const std::vector<backend::command_queue> &q, const std::vector<T> &host,
                    backend::mem_flags flags = backend::MEM_READ_WRITE) { vector(q, host, flags).swap(*this); }
        void resize(size_t size) { vector(size).swap(*this); }
        void resize(const std::vector<T> &host) { vector(host).swap(*this); }
        void clear() { vector().swap(*this); }
        void swap(vector &v) { std::swap(queue, v.queue); std::swap(part, v.part); std::swap(buf, v.buf); }

        // ---- access -------------------------------------------------------------
        const backend::device_vector<T> &operator()(unsigned d = 0) const { return buf[d]; }
        backend::device_vector<T> &operator()(unsigned d = 0) { return buf[d]; }

        const_iterator begin() const { return const_iterator(*this, 0); }
        const_iterator end() const { return const_iterator(*this, size()); }
        iterator begin() { return iterator(*this, 0); }
        iterator end() { return iterator(*this, size()); }

        const element operator[](size_t index) const {
            size_t d = owner(index);
            return element(queue[d], buf[d], index - part[d]);
        }
        element operator[](size_t index) {
            size_t d = owner(index);
            return element(queue[d], buf[d], index - part[d]);
        }
        const element at(size_t index) const { if (index >= size()) throw std::out_of_range("vex::vector"); return (*this)[index]; }
        element at(size_t index) { if (index >= size()) throw std::out_of_range("vex::vector"); return (*this)[index]; }

        size_t size() const { return part.empty() ? 0 : part.back(); }
        size_t nparts() const { return queue.size(); }
        size_t part_size(unsigned d) const { return part[d + 1] - part[d]; }
        size_t part_start(unsigned d) const { return part[d]; }
        const std::vector<size_t> &partition() const { return part; }
        const std::vector<backend::command_queue> &queue_list() const { return queue; }

        /// The same device memory seen as elements of another type (vector.hpp:472-494;
        /// tests/reinterpret.cpp): sizes scale with sizeof(T) / sizeof(U), partitions stay in place.
        template <class U>
        vector<U> reinterpret() const {
            vector<U> r;
            r.queue = queue;
            r.part.resize(part.size());
            for (size_t d = 0; d < part.size(); ++d) {
                precondition(part[d] * sizeof(T) % sizeof(U) == 0, "reinterpret: partition size is not a multiple of the new element size");
                r.part[d] = part[d] * sizeof(T) / sizeof(U);
            }
            for (const auto &b : buf) r.buf.push_back(b.template reinterpret<U>());
            return r;
        }

        typename backend::device_vector<T>::mapped_array map(unsigned d = 0) { return buf[d].map(queue[d]); }
        typename backend::device_vector<T>::mapped_array map(unsigned d = 0) const { return buf[d].map(queue[d]); }

        void write_data(size_t offset, size_t size, const T *hostptr, bool blocking) {
            if (!size) return;
            for (unsigned d = 0; d < queue.size(); ++d) {
                size_t start = std::max(offset, part[d]), stop = std::min(offset + size, part[d + 1]);
                if (stop > start) buf[d].write(queue[d], start - part[d], stop - start, hostptr + start - offset);
            }
            if (blocking) for (unsigned d = 0; d < queue.size(); ++d) queue[d].finish();
        }
        void read_data(size_t offset, size_t size, T *hostptr, bool blocking) const {
            if (!size) return;
            for (unsigned d = 0; d < queue.size(); ++d) {
                size_t start = std::max(offset, part[d]), stop = std::min(offset + size, part[d + 1]);
                if (stop > start) buf[d].read(queue[d], start - part[d], stop - start, hostptr + start - offset);
            }
            if (blocking) for (unsigned d = 0; d < queue.size(); ++d) queue[d].finish();
        }

        // ---- assignment (vector.hpp:667-801) -------------------------------------
        const vector &operator=(const vector &x) {
            if (&x != this) detail::assign_expression<assign::SET>(expr_ref_type(*this), expr_ref_type(x), queue, part);
            return *this;
        }
        const vector &operator=(vector &&v) { swap(v); return *this; }

#define VEXCL_VECTOR_ASSIGN(op, tag)                                                                      \
        template <class Expr>                                                                             \
        typename std::enable_if<detail::is_operand<Expr>::value &&                                        \
            !std::is_same<typename std::decay<Expr>::type, vector>::value, const vector &>::type          \
        operator op(const Expr &expr) {                                                                   \
            detail::assign_any<assign::tag>(expr_ref_type(*this), *this, detail::as_expr<Expr>::get(expr), queue, part); \
            return *this;                                                                                 \
        }
        VEXCL_VECTOR_ASSIGN(=, SET)   VEXCL_VECTOR_ASSIGN(+=, ADD)  VEXCL_VECTOR_ASSIGN(-=, SUB)
        VEXCL_VECTOR_ASSIGN(*=, MUL)  VEXCL_VECTOR_ASSIGN(/=, DIV)  VEXCL_VECTOR_ASSIGN(%=, MOD)
        VEXCL_VECTOR_ASSIGN(&=, AND)  VEXCL_VECTOR_ASSIGN(|=, OR)   VEXCL_VECTOR_ASSIGN(^=, XOR)
        VEXCL_VECTOR_ASSIGN(<<=, LSH) VEXCL_VECTOR_ASSIGN(>>=, RSH)
#undef VEXCL_VECTOR_ASSIGN
        const vector &operator+=(const vector &x) { detail::assign_expression<assign::ADD>(expr_ref_type(*this), expr_ref_type(x), queue, part); return *this; }
        const vector &operator-=(const vector &x) { detail::assign_expression<assign::SUB>(expr_ref_type(*this), expr_ref_type(x), queue, part); return *this; }
        const vector &operator*=(const vector &x) { detail::assign_expression<assign::MUL>(expr_ref_type(*this), expr_ref_type(x), queue, part); return *this; }
        const vector &operator/=(const vector &x) { detail::assign_expression<assign::DIV>(expr_ref_type(*this), expr_ref_type(x), queue, part); return *this; }

        // node interface, so that a vector can be used where a terminal is expected
        void get_props(detail::prop_context &p) const { expr_ref_type(*this).get_props(p); }

    private:
        template <class U> friend class vector;
        std::vector<backend::command_queue> queue;
        std::vector<size_t> part;
        std::vector<backend::device_vector<T>> buf;

        size_t owner(size_t index) const {
            size_t d = std::upper_bound(part.begin(), part.end(), index) - part.begin() - 1;
            return std::min(d, queue.size() - 1);
        }

        void allocate_buffers(backend::mem_flags flags, const T *hostptr) {
            for (unsigned d = 0; d < queue.size(); ++d)
                if (size_t psize = part[d + 1] - part[d])
                    buf[d] = backend::device_vector<T>(queue[d], psize, hostptr ? hostptr + part[d] : 0, flags);
        }
};

template <class T> void swap(vector<T> &x, vector<T> &y) { x.swap(y); }

// ---- copy (vector.hpp:998-1228) -----------------------------------------------------
template <class T> void copy(const vector<T> &dv, T *hv, bool blocking = true) { dv.read_data(0, dv.size(), hv, blocking); }
template <class T> void copy(const T *hv, vector<T> &dv, bool blocking = true) { dv.write_data(0, dv.size(), hv, blocking); }
template <class T> void copy(const vector<T> &dv, std::vector<T> &hv, bool blocking = true) {
    precondition(hv.size() >= dv.size(), "Host vector is too small");
    dv.read_data(0, dv.size(), hv.data(), blocking);
}
template <class T> void copy(const std::vector<T> &hv, vector<T> &dv, bool blocking = true) {
    precondition(hv.size() >= dv.size(), "Host vector is too small");
    dv.write_data(0, dv.size(), hv.data(), blocking);
}
template <class T1, class T2> void copy(const vector<T1> &src, vector<T2> &dst) { dst = src; }

/// copy(q, device, host) on explicitly given queues (tests/events.cpp:90-104).
template <class T> void copy(const std::vector<backend::command_queue> &q, const vector<T> &dv, std::vector<T> &hv, bool blocking = true) {
    for (unsigned d = 0; d < q.size(); ++d)
        if (size_t n = dv.part_size(d)) dv(d).read(q[d], 0, n, hv.data() + dv.part_start(d), false);
    if (blocking) for (const auto &queue : q) queue.finish();
}
template <class T> void copy(const std::vector<backend::command_queue> &q, const std::vector<T> &hv, vector<T> &dv, bool blocking = true) {
    for (unsigned d = 0; d < q.size(); ++d)
        if (size_t n = dv.part_size(d)) dv(d).write(q[d], 0, n, hv.data() + dv.part_start(d), false);
    if (blocking) for (const auto &queue : q) queue.finish();
}

namespace detail {
    template <class It, class = void> struct is_device_iterator : std::false_type {};
    template <class It> struct is_device_iterator<It, typename std::enable_if<It::device_iterator>::type> : std::true_type {};
}
/// Device range -> host iterator.
template <class InIt, class OutIt>
typename std::enable_if<detail::is_device_iterator<InIt>::value && !detail::is_device_iterator<OutIt>::value, OutIt>::type
copy(InIt first, InIt last, OutIt result, bool blocking = true) {
    first.vec->read_data(first.pos, last - first, &result[0], blocking);
    return result + (last - first);
}
/// Host range -> device iterator.
template <class InIt, class OutIt>
typename std::enable_if<!detail::is_device_iterator<InIt>::value && detail::is_device_iterator<OutIt>::value, OutIt>::type
copy(InIt first, InIt last, OutIt result, bool blocking = true) {
    result.vec->write_data(result.pos, last - first, &first[0], blocking);
    return result + (last - first);
}

/// Relative device performance probe (vector.hpp:1237-1255).  All GPUs of a
/// node are identical MI355X parts: the weight is 1.
inline double device_vector_perf(const backend::command_queue &) { return 1.0; }

/// Prints the vector (vector.hpp:1259-1282).
template <class T>
std::ostream &operator<<(std::ostream &o, const vector<T> &t) {
    std::vector<T> data(t.size());
    copy(t, data);
    // ten elements per line behind the index of the first; integers in 6 columns, reals as %14.6e
    o << "{" << std::setprecision(6);
    for (size_t i = 0; i < data.size(); ++i) {
        if (i % 10 == 0) o << "\n" << std::setw(6) << i << ":";
        if (std::is_integral<T>::value) o << " " << std::setw(6) << data[i];
        else o << std::scientific << std::setw(14) << data[i];
    }
    return o << "\n}\n";
}

namespace detail {
/// `&x` as an operand: the address of the current element of x (vector_arithmetics.cpp:263 of the
/// reference's tests, `*if_else(c, &y, &z)`).
template <class T>
struct vector_address : expression_base {
    typedef T *value_type;
    const vector<T> *v;
    explicit vector_address(const vector<T> *v) : v(v) {}
    void preamble(gen_context &c) const { c.next(); }
    void params(gen_context &c) const { c.src.template parameter<global_ptr<T>>(c.next()); }
    void local_init(gen_context &c) const { c.next(); }
    void emit(gen_context &c) const { c.src << "( " << c.next() << " + idx )"; }
    void set_args(arg_context &a) const { a.next(); a.krn.push_arg((*v)(a.device)); }
    void get_props(prop_context &p) const {
        if (p.empty()) { p.queue = v->queue_list(); p.part = v->partition(); p.size = v->size(); }
    }
};
template <class T> struct expr_kind<vector_address<T>> : std::integral_constant<int, 0> {};
// (operations.hpp, y = z +- A * x in one pass: a vector terminal is the z of such an expression; the target must be a plain vector)
template <class T> struct axpby_leaf<vector_ref<T>, void> : std::integral_constant<int, 1> { static const void *get(const vector_ref<T> &e) { return e.v; } };
template <class T> class plain_vector_of<vector<T>> { public: typedef T type; };
template <class T> struct is_extra_operand<vector<T> *> : std::true_type {};
template <class T> struct as_expr<vector<T> *, void> {
    typedef vector_address<T> type;
    static type get(vector<T> *const &v) { return type(v); }
};
} // namespace detail

} // namespace vex
#endif
