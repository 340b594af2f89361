#ifndef VEXCL_MULTIVECTOR_HPP
#define VEXCL_MULTIVECTOR_HPP
// vex::multivector<T, N>: N equally sized device vectors that are assigned
// together, and vex::tie(v1, v2, ...) which ties existing vectors the same way
// (reference: vexcl/multivector.hpp:56-118 grammar, :138-720 the class,
// :722-790 vex::tie; tests/multivector_create.cpp, multivector_arithmetics.cpp).
//
// Design.  The reference walks a Proto tree with a "component extractor"
// transform.  Here a multi-expression is an ordinary node tree whose leaves may
// be mv_ref<T, N> (a multivector) or tuple_node<E...> (a std::tuple of
// per-component operands: std::make_tuple(1, 2, 3), std::tie(a + b, a - b)).
// component_of<I, E> (operations.hpp) rebuilds, at compile time, the ordinary
// expression of component I.  All N right-hand sides are then generated into ONE
// kernel (`vexcl_multivector_kernel`, multivector.hpp:486-600 of the reference):
// every component is evaluated into a register before the first one is stored,
// so `vex::tie(x, y) = std::tie(y, x)` swaps and each source element is read
// once per use.  Components holding A*x terms fall back to per-component
// assignment, as the reference does (multivector.hpp:602-650).
#include <array>
#include <memory>
#include <tuple>
#include "operations.hpp"
#include "vector.hpp"

namespace vex {

template <class T, size_t N> class multivector;

namespace detail {

/// A multivector inside an expression: never generates code itself, only through
/// component_of<I, ...>, which turns it into vector_ref<T> of component I.
template <class T, size_t N>
struct mv_ref : expression_base {
    typedef T value_type;
    const multivector<T, N> *mv;
    mv_ref(const multivector<T, N> &m) : mv(&m) {}
    void get_props(prop_context &p) const { (*mv)(0).get_props(p); }
};
template <class T, size_t N> struct mv_dim<mv_ref<T, N>> : std::integral_constant<size_t, N> {};
template <size_t I, class T, size_t N> struct component_of<I, mv_ref<T, N>, void> {
    static_assert(I < N, "component index out of range");
    typedef vector_ref<T> type;
    static type get(const mv_ref<T, N> &m) { return type((*m.mv)(I)); }
};

/// std::tuple operand: one operand (scalar or expression) per component.
template <class... E>
struct tuple_node : expression_base {
    typedef typename std::common_type<typename E::value_type...>::type value_type;
    std::tuple<E...> e;
    explicit tuple_node(const E &...e) : e(e...) {}
    void get_props(prop_context &p) const { tuple_for_each(e, [&p](const auto &a, size_t) { a.get_props(p); }); }
};
template <class... E> struct mv_dim<tuple_node<E...>> : std::integral_constant<size_t, sizeof...(E)> {};
template <size_t I, class... E> struct component_of<I, tuple_node<E...>, void> {
    static_assert(I < sizeof...(E), "component index out of range");
    typedef typename std::tuple_element<I, std::tuple<E...>>::type type;
    static_assert(mv_dim<type>::value == 0, "tuple elements must be ordinary expressions");
    static const type &get(const tuple_node<E...> &t) { return std::get<I>(t.e); }
};

template <class... E> struct all_operands_tuple : std::true_type {};
template <class H, class... T> struct all_operands_tuple<H, T...>
    : std::integral_constant<bool, is_operand<H>::value && all_operands_tuple<T...>::value> {};
template <class T, size_t> struct repeat { typedef T type; };
template <class... E> struct is_extra_operand<std::tuple<E...>> : all_operands_tuple<E...> {};
template <class... E> struct as_expr<std::tuple<E...>, void> {
    typedef tuple_node<as_expr_t<typename std::decay<E>::type>...> type;
    template <size_t... K>
    static type make(const std::tuple<E...> &t, std::index_sequence<K...>) {
        return type(as_expr<typename std::decay<E>::type>::get(std::get<K>(t))...);
    }
    static type get(const std::tuple<E...> &t) { return make(t, std::index_sequence_for<E...>()); }
};
template <class S, size_t N> struct is_extra_operand<std::array<S, N>> : std::is_arithmetic<S> {};
template <class S, size_t N, class Seq = std::make_index_sequence<N>> struct array_as_tuple;
template <class S, size_t N, size_t... K> struct array_as_tuple<S, N, std::index_sequence<K...>> {
    template <size_t> using scalar = scalar_terminal<S>;
    typedef tuple_node<scalar<K>...> type;
    static type get(const std::array<S, N> &a) { return type(scalar<K>(a[K])...); }
};
template <class S, size_t N> struct as_expr<std::array<S, N>, void> {
    typedef typename array_as_tuple<S, N>::type type;
    static type get(const std::array<S, N> &a) { return array_as_tuple<S, N>::get(a); }
};

/// A * X with X a multivector: component I is A * X(I).
template <class M, class T, size_t N> struct mv_dim<additive_operator<M, multivector<T, N>>> : std::integral_constant<size_t, N> {};
template <size_t I, class M, class T, size_t N> struct component_of<I, additive_operator<M, multivector<T, N>>, void> {
    typedef additive_operator<M, vector<T>> type;
    static type get(const additive_operator<M, multivector<T, N>> &op) { return type(op.A, op.x(I)); }
};

// ---- the fused multi-assignment kernel ------------------------------------------
template <class OP, class... L, class... R>
std::string multi_assignment_source(const std::tuple<L...> &lhs, const std::tuple<R...> &rhs, const backend::command_queue &q) {
    static_assert(sizeof...(L) == sizeof...(R), "component count mismatch");
    backend::source_generator source(q);
    auto both = [&](auto &&f) { tuple_for_each(lhs, f); tuple_for_each(rhs, f); };
    { gen_context c(source, q); both([&c](const auto &a, size_t) { a.preamble(c); }); }
    source.begin_kernel("vexcl_multivector_kernel");
    source.begin_kernel_parameters();
    source.template parameter<size_t>("n");
    { gen_context c(source, q); both([&c](const auto &a, size_t) { a.params(c); }); }
    source.end_kernel_parameters();
    source.grid_stride_loop().open("{");
    int lhs_terminals = 0;
    {
        gen_context c(source, q);
        tuple_for_each(lhs, [&c](const auto &a, size_t) { a.local_init(c); });
        lhs_terminals = c.pos;
        tuple_for_each(rhs, [&c](const auto &a, size_t) { a.local_init(c); });
    }
    {
        gen_context c(source, q);
        c.pos = lhs_terminals;
        tuple_for_each(rhs, [&](const auto &a, size_t i) {
            typedef typename std::decay<decltype(a)>::type node;
            source.new_line() << type_name<typename node::value_type>() << " buf_" << i + 1 << " = ";
            a.emit(c);
            source << ";";
        });
    }
    {
        gen_context c(source, q);
        tuple_for_each(lhs, [&](const auto &a, size_t i) {
            source.new_line();
            a.emit(c);
            source << " " << OP::string() << " buf_" << i + 1 << ";";
        });
    }
    source.close("}");
    source.end_kernel();
    return source.str();
}

template <class OP, class... L, class... R>
void assign_multiexpression(const std::tuple<L...> &lhs, const std::tuple<R...> &rhs,
        const std::vector<backend::command_queue> &queue, const std::vector<size_t> &part)
{
    static kernel_cache cache;
    for (unsigned d = 0; d < queue.size(); ++d) {
        size_t psize = part[d + 1] - part[d];
        if (!psize) continue;
        auto kernel = cache.find(queue[d]);
        if (kernel == cache.end())
            kernel = cache.insert(queue[d], backend::kernel(queue[d],
                        multi_assignment_source<OP>(lhs, rhs, queue[d]), "vexcl_multivector_kernel"));
        backend::kernel &krn = kernel->second;
        krn.push_arg(psize);
        arg_context a(krn, d, part[d]);
        tuple_for_each(lhs, [&a](const auto &n, size_t) { n.set_args(a); });
        tuple_for_each(rhs, [&a](const auto &n, size_t) { n.set_args(a); });
        krn.config_streaming(queue[d], psize, 1);
        krn(queue[d]);
    }
}

/// The left-hand sides of a multi-assignment, as the target of A*X terms
/// (SpMat::apply(multivector, multi_target) writes all components in one pass).
template <class... Ts> struct multi_target { std::tuple<vector<Ts> &...> v; };

// a tuple operand holding A*x terms can only be assigned component by component
template <class... E> struct expr_kind<tuple_node<E...>>
    : std::integral_constant<int, all_vector_kind<E...>::value ? 0 : -1> {};

template <class... C> struct all_fusable : std::true_type {};
template <class H, class... T> struct all_fusable<H, T...>
    : std::integral_constant<bool, expr_kind<H>::value == 0 && !direct_assign<H>::value && all_fusable<T...>::value> {};

/// lhs(I) OP component I of expr, for all I (multivector.hpp:486-650).
template <class OP, class... Ts, class Expr, size_t... I>
void assign_multi(const std::tuple<vector<Ts> &...> &lhs, const Expr &expr, std::index_sequence<I...>,
        const std::vector<backend::command_queue> *on = nullptr)     // enqueue.hpp: queues to launch on
{
    constexpr size_t N = sizeof...(Ts);
    static_assert(mv_dim<Expr>::value == 0 || mv_dim<Expr>::value == N,
            "the expression and the multivector it is assigned to have different numbers of components");
    auto &first = std::get<0>(lhs);
    const auto &queue = on ? *on : first.queue_list();
    const auto &part = first.partition();
    precondition(queue.size() == first.queue_list().size(), "as many queues as the left-hand side has partitions are expected");
    {
        prop_context p;
        expr.get_props(p);
        precondition(p.size == 0 || p.empty() || p.size == first.size(), "Incompatible expression sizes");
    }
    auto rhs = std::make_tuple(component_of<I, Expr>::get(expr)...);
    constexpr int kind = expr_kind<Expr>::value;
    if constexpr (all_fusable<typename component_of<I, Expr>::type...>::value) {
        auto l = std::make_tuple(vector_ref<Ts>(std::get<I>(lhs))...);
        assign_multiexpression<OP>(l, rhs, queue, part);
    } else if constexpr (kind == 1 || kind == 2) {
        // A*X terms joined by + / - / scalars with an (optional) vector part: one fused kernel
        // for the vector parts of all components, then every product term once -- it writes
        // all components in one pass over the matrix (operations.hpp assign_any, N-fold)
        constexpr bool set = std::is_same<OP, assign::SET>::value;
        constexpr bool sub = std::is_same<OP, assign::SUB>::value;
        static_assert(set || sub || std::is_same<OP, assign::ADD>::value, "A*x terms support only =, += and -=");
        bool append = !set;
        if constexpr (kind == 2) {
            auto vp = vector_part(expr);
            typedef decltype(vp) VP;
            auto l = std::make_tuple(vector_ref<Ts>(std::get<I>(lhs))...);
            auto r = std::make_tuple(component_of<I, VP>::get(vp)...);
            assign_multiexpression<OP>(l, r, queue, part);
            append = true;
        }
        multi_target<Ts...> target = {lhs};
        apply_transforms(target, expr, sub ? -1.0 : 1.0, append);
    } else {
        int dummy[] = {0, (assign_any<OP>(vector_ref<Ts>(std::get<I>(lhs)), std::get<I>(lhs), std::get<I>(rhs), queue, part), 0)...};
        (void)dummy;
    }
}

template <class... T> struct all_exprs : std::true_type {};
template <class H, class... T> struct all_exprs<H, T...> : std::integral_constant<bool, is_expr<H>::value && all_exprs<T...>::value> {};
} // namespace detail

#define VEXCL_MULTI_ASSIGN_ONE(Self, op, tag)                                                             \
        template <class Expr>                                                                             \
        typename std::enable_if<detail::is_operand<Expr>::value, const Self &>::type                      \
        operator op(const Expr &expr) {                                                                   \
            detail::assign_multi<assign::tag>(refs(), detail::as_expr<Expr>::get(expr), std::make_index_sequence<dim>()); \
            return *this;                                                                                 \
        }

/// N device vectors of equal size and partitioning, assigned by one kernel
/// (multivector.hpp:138-720).
template <class T, size_t N>
class multivector : public detail::expression_base {
    static_assert(N > 0, "What's the point?");
    public:
        typedef vex::vector<T> subtype;
        typedef std::array<T, N> value_type;
        typedef T sub_value_type;
        typedef detail::mv_ref<T, N> expr_ref_type;
        static const size_t dim = N;
        static const size_t NDIM = N;

        /// Proxy of one element: N values, one from each component (multivector.hpp:161-199).
        class element {
            public:
                operator value_type() const {
                    value_type v;
                    for (size_t i = 0; i < N; ++i) v[i] = (*vec)(i)[index];
                    return v;
                }
                value_type operator=(const value_type &v) {
                    for (size_t i = 0; i < N; ++i) (*vec)(i)[index] = v[i];
                    return v;
                }
                T operator()(size_t i) const { return (*vec)(i)[index]; }
            private:
                element(multivector &v, size_t i) : vec(&v), index(i) {}
                multivector *vec; size_t index;
                friend class multivector;
        };
        class const_element {
            public:
                operator value_type() const {
                    value_type v;
                    for (size_t i = 0; i < N; ++i) v[i] = (*vec)(i)[index];
                    return v;
                }
                T operator()(size_t i) const { return (*vec)(i)[index]; }
            private:
                const_element(const multivector &v, size_t i) : vec(&v), index(i) {}
                const multivector *vec; size_t index;
                friend class multivector;
        };

        template <class V, class E>
        class iterator_type {
            public:
                typedef std::random_access_iterator_tag iterator_category;
                typedef std::array<T, N> value_type; typedef ptrdiff_t difference_type; typedef E *pointer; typedef E reference;
                iterator_type() : vec(0), pos(0) {}
                iterator_type(V &v, size_t p) : vec(&v), pos(p) {}
                E operator*() const { return (*vec)[pos]; }
                iterator_type &operator++() { ++pos; return *this; }
                iterator_type operator++(int) { iterator_type t(*this); ++pos; return t; }
                iterator_type &operator+=(ptrdiff_t d) { pos += d; return *this; }
                iterator_type operator+(ptrdiff_t d) const { return iterator_type(*vec, pos + d); }
                ptrdiff_t operator-(const iterator_type &o) const { return (ptrdiff_t)pos - (ptrdiff_t)o.pos; }
                bool operator==(const iterator_type &o) const { return pos == o.pos; }
                bool operator!=(const iterator_type &o) const { return pos != o.pos; }
                V *vec; size_t pos;
        };
        typedef iterator_type<multivector, element> iterator;
        typedef iterator_type<const multivector, const_element> const_iterator;

        // ---- construction (multivector.hpp:260-330) ---------------------------------
        multivector() { for (auto &v : vec) v.reset(new subtype()); }

        /// Host data holds the components one after another: N * size values.
        multivector(const std::vector<backend::command_queue> &queue, const std::vector<T> &host,
                    backend::mem_flags flags = backend::MEM_READ_WRITE)
        {
            precondition(host.size() % N == 0, "host data size is not a multiple of the number of components");
            size_t n = host.size() / N;
            for (size_t i = 0; i < N; ++i) vec[i].reset(new subtype(queue, n, host.data() + i * n, flags));
        }
        multivector(const std::vector<backend::command_queue> &queue, size_t size, const T *host = 0,
                    backend::mem_flags flags = backend::MEM_READ_WRITE)
        {
            for (size_t i = 0; i < N; ++i) vec[i].reset(new subtype(queue, size, host ? host + i * size : 0, flags));
        }
        explicit multivector(size_t size) { for (auto &v : vec) v.reset(new subtype(size)); }
        multivector(const std::vector<T> &host) {
            precondition(host.size() % N == 0, "host data size is not a multiple of the number of components");
            size_t n = host.size() / N;
            for (size_t i = 0; i < N; ++i) vec[i].reset(new subtype(current_context().queue(), n, host.data() + i * n));
        }
        multivector(const multivector &mv) : detail::expression_base() {
#ifdef VEXCL_SHOW_COPIES
            std::cout << "Copying vex::multivector<" << type_name<T>() << ", " << N << "> of size " << mv.size() << std::endl;
#endif
            for (size_t i = 0; i < N; ++i) vec[i].reset(new subtype(mv(i)));
        }
        multivector(multivector &&mv) noexcept { for (size_t i = 0; i < N; ++i) vec[i] = std::move(mv.vec[i]); }

        /// From a multi-expression: size and queues are taken from it (multivector.hpp:332-360).
        template <class Expr, class = typename std::enable_if<
            detail::is_operand<Expr>::value && !detail::is_scalar<Expr>::value &&
            !std::is_same<typename std::decay<Expr>::type, multivector>::value>::type>
        multivector(const Expr &expr) {
            std::vector<backend::command_queue> queue; std::vector<size_t> part; size_t n;
            get_expression_properties(expr, queue, part, n);
            precondition(!queue.empty() && !part.empty(), "Can not determine expression size and queue list");
            for (auto &v : vec) v.reset(new subtype(queue, n));
            *this = expr;
        }

        void resize(const std::vector<backend::command_queue> &queue, size_t size) { for (auto &v : vec) v->resize(queue, size); }
        void resize(size_t size) { for (auto &v : vec) v->resize(size); }
        void resize(const multivector &mv) { for (size_t i = 0; i < N; ++i) vec[i]->resize(mv(i)); }
        void clear() { for (auto &v : vec) v->clear(); }
        void swap(multivector &o) { std::swap(vec, o.vec); }

        // ---- access -------------------------------------------------------------------
        size_t size() const { return vec[0]->size(); }
        const subtype &operator()(size_t i) const { return *vec[i]; }
        subtype &operator()(size_t i) { return *vec[i]; }
        const_iterator begin() const { return const_iterator(*this, 0); }
        const_iterator end() const { return const_iterator(*this, size()); }
        iterator begin() { return iterator(*this, 0); }
        iterator end() { return iterator(*this, size()); }
        const_element operator[](size_t i) const { return const_element(*this, i); }
        element operator[](size_t i) { return element(*this, i); }
        const std::vector<backend::command_queue> &queue_list() const { return vec[0]->queue_list(); }

        // ---- assignment (multivector.hpp:400-484) ----------------------------------------
        const multivector &operator=(const multivector &mv) {
            if (&mv != this)
                detail::assign_multi<assign::SET>(refs(), expr_ref_type(mv), std::make_index_sequence<N>());
            return *this;
        }
        const multivector &operator=(multivector &&mv) { swap(mv); return *this; }

        VEXCL_MULTI_ASSIGN_ONE(multivector, =, SET)   VEXCL_MULTI_ASSIGN_ONE(multivector, +=, ADD)
        VEXCL_MULTI_ASSIGN_ONE(multivector, -=, SUB)  VEXCL_MULTI_ASSIGN_ONE(multivector, *=, MUL)
        VEXCL_MULTI_ASSIGN_ONE(multivector, /=, DIV)  VEXCL_MULTI_ASSIGN_ONE(multivector, %=, MOD)
        VEXCL_MULTI_ASSIGN_ONE(multivector, &=, AND)  VEXCL_MULTI_ASSIGN_ONE(multivector, |=, OR)
        VEXCL_MULTI_ASSIGN_ONE(multivector, ^=, XOR)  VEXCL_MULTI_ASSIGN_ONE(multivector, <<=, LSH)
        VEXCL_MULTI_ASSIGN_ONE(multivector, >>=, RSH)

        void get_props(detail::prop_context &p) const { vec[0]->get_props(p); }

        /// The components as a tuple of references (what the assignment machinery works on).
        auto components() const { return refs(); }

    private:
        std::array<std::unique_ptr<subtype>, N> vec;

        template <size_t... I>
        auto refs_impl(std::index_sequence<I...>) const { return std::tuple<typename detail::repeat<subtype &, I>::type...>(*vec[I]...); }
        auto refs() const { return refs_impl(std::make_index_sequence<N>()); }
};

template <class T, size_t N> void swap(multivector<T, N> &a, multivector<T, N> &b) { a.swap(b); }

/// Copies: the host vector holds the components one after another (multivector.hpp:800-830).
template <class T, size_t N>
void copy(const multivector<T, N> &mv, std::vector<T> &hv) {
    precondition(hv.size() >= N * mv.size(), "Host vector is too small");
    for (size_t i = 0; i < N; ++i) copy(mv(i), hv.data() + i * mv.size());
}
template <class T, size_t N>
void copy(const std::vector<T> &hv, multivector<T, N> &mv) {
    precondition(hv.size() >= N * mv.size(), "Host vector is too small");
    for (size_t i = 0; i < N; ++i) copy(hv.data() + i * mv.size(), mv(i));
}

// ---- vex::tie (multivector.hpp:722-790; tests/multivector_arithmetics.cpp:96-120) -----------
/// Existing vectors tied together so that one kernel assigns all of them:
/// vex::tie(a, b) = std::tie(x + y, x - y);
template <class... Ts>
class tied_vectors {
    public:
        static const size_t dim = sizeof...(Ts);
        explicit tied_vectors(vector<Ts> &...v) : vec(v...) {
            precondition(same_size(std::make_index_sequence<dim>()), "tied vectors must have equal sizes");
        }
        template <size_t I> auto &get() const { return std::get<I>(vec); }
        size_t size() const { return std::get<0>(vec).size(); }

        VEXCL_MULTI_ASSIGN_ONE(tied_vectors, =, SET)   VEXCL_MULTI_ASSIGN_ONE(tied_vectors, +=, ADD)
        VEXCL_MULTI_ASSIGN_ONE(tied_vectors, -=, SUB)  VEXCL_MULTI_ASSIGN_ONE(tied_vectors, *=, MUL)
        VEXCL_MULTI_ASSIGN_ONE(tied_vectors, /=, DIV)  VEXCL_MULTI_ASSIGN_ONE(tied_vectors, %=, MOD)
        VEXCL_MULTI_ASSIGN_ONE(tied_vectors, &=, AND)  VEXCL_MULTI_ASSIGN_ONE(tied_vectors, |=, OR)
        VEXCL_MULTI_ASSIGN_ONE(tied_vectors, ^=, XOR)  VEXCL_MULTI_ASSIGN_ONE(tied_vectors, <<=, LSH)
        VEXCL_MULTI_ASSIGN_ONE(tied_vectors, >>=, RSH)
    private:
        std::tuple<vector<Ts> &...> vec;
        const std::tuple<vector<Ts> &...> &refs() const { return vec; }
        template <size_t... I> bool same_size(std::index_sequence<I...>) const {
            bool ok = true;
            int dummy[] = {0, (ok = ok && std::get<I>(vec).size() == std::get<0>(vec).size(), 0)...};
            (void)dummy;
            return ok;
        }
};

template <class... Ts>
tied_vectors<Ts...> tie(vector<Ts> &...v) { return tied_vectors<Ts...>(v...); }

/// Writable expressions tied together -- slices of vectors, dereferenced pointer expressions:
///     vex::tie(rows[1](x), rows[2](x)) = std::make_tuple(1, 2);     vex::tie(*if_else(c, &y, &z)) = 42;
/// One kernel evaluates every right-hand side, then stores through every left-hand side.
template <class... L>
class tied_expressions {
    public:
        static const size_t dim = sizeof...(L);
        explicit tied_expressions(const L &...l) : lhs(detail::as_expr<L>::get(l)...) {}

#define VEXCL_TIED_ASSIGN(op, tag)                                                                        \
        template <class Expr>                                                                             \
        typename std::enable_if<detail::is_operand<Expr>::value, const tied_expressions &>::type          \
        operator op(const Expr &expr) const {                                                             \
            assign<assign::tag>(detail::as_expr<Expr>::get(expr), std::make_index_sequence<dim>());       \
            return *this;                                                                                 \
        }
        VEXCL_TIED_ASSIGN(=, SET)   VEXCL_TIED_ASSIGN(+=, ADD)  VEXCL_TIED_ASSIGN(-=, SUB)  VEXCL_TIED_ASSIGN(*=, MUL)
        VEXCL_TIED_ASSIGN(/=, DIV)  VEXCL_TIED_ASSIGN(%=, MOD)  VEXCL_TIED_ASSIGN(&=, AND)  VEXCL_TIED_ASSIGN(|=, OR)
        VEXCL_TIED_ASSIGN(^=, XOR)  VEXCL_TIED_ASSIGN(<<=, LSH) VEXCL_TIED_ASSIGN(>>=, RSH)
#undef VEXCL_TIED_ASSIGN
    private:
        std::tuple<detail::as_expr_t<L>...> lhs;

        template <class OP, class Expr, size_t... I>
        void assign(const Expr &expr, std::index_sequence<I...>) const {
            static_assert(detail::mv_dim<Expr>::value == 0 || detail::mv_dim<Expr>::value == dim,
                    "the expression and the tied left-hand sides have different numbers of components");
            static_assert(detail::expr_kind<Expr>::value == 0, "tied expressions take vector expressions only");
            detail::prop_context p;
            detail::tuple_for_each(lhs, [&p](const auto &a, size_t) { a.get_props(p); });
            expr.get_props(p);
            precondition(!p.empty(), "vex::tie: can not determine the size and the queues of the assignment");
            auto rhs = std::make_tuple(detail::component_of<I, Expr>::get(expr)...);
            detail::assign_multiexpression<OP>(lhs, rhs, p.queue, p.part);
        }
};

template <class... L>
typename std::enable_if<(sizeof...(L) > 0) && detail::all_exprs<L...>::value, tied_expressions<L...>>::type
tie(const L &...l) { return tied_expressions<L...>(l...); }

#undef VEXCL_MULTI_ASSIGN_ONE

} // namespace vex
#endif
