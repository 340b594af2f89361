#ifndef VEXCL_OPERATIONS_HPP
#define VEXCL_OPERATIONS_HPP
// The expression engine (reference: vexcl/operations.hpp, 2 326 lines on
// Boost.Proto).  Re-designed as plain C++17 expression templates: every node is
// a small value type that knows how to
//     preamble()    emit the device functions it needs          (operations.hpp:1010-1124)
//     params()      declare its kernel parameters                (:146-165, kernel_param_declaration)
//     local_init()  emit per-element statements before the expression (:166-190, local_terminal_init)
//     emit()        print its part of the expression text        (:1209-1354, vector_expr_context)
//     set_args()    push its kernel arguments for one device     (:1356-1409, kernel_arg_setter)
//     get_props()   report queue list / partition / size         (:1411-1460, expression_properties)
// Terminals are numbered in traversal order, lhs first, exactly as the
// reference does (prm_1, prm_2, ...), so generated kernels have the shape of
// SURVEY appendix A.1 and are cached per expression TYPE per context.
#include <functional>
#include <limits>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "backend.hpp"
#include "cache.hpp"

namespace vex {

/// Assignment operators (operations.hpp:63-96).
namespace assign {
#define VEXCL_ASSIGN_OP(name, op) struct name { static std::string string() { return #op; } }
    VEXCL_ASSIGN_OP(SET, =);  VEXCL_ASSIGN_OP(ADD, +=); VEXCL_ASSIGN_OP(SUB, -=);
    VEXCL_ASSIGN_OP(MUL, *=); VEXCL_ASSIGN_OP(DIV, /=); VEXCL_ASSIGN_OP(MOD, %=);
    VEXCL_ASSIGN_OP(AND, &=); VEXCL_ASSIGN_OP(OR, |=);  VEXCL_ASSIGN_OP(XOR, ^=);
    VEXCL_ASSIGN_OP(LSH, <<=); VEXCL_ASSIGN_OP(RSH, >>=);
#undef VEXCL_ASSIGN_OP
}

namespace detail {

// ---- traversal contexts -----------------------------------------------------
struct gen_context {
    backend::source_generator &src;
    const backend::command_queue &queue;
    std::string prefix;
    int pos;
    std::set<std::string> own_seen;
    std::set<std::string> &seen;    // functions / tagged terminals already handled in this pass
    bool symbolic;                  // generator.hpp: scalars are printed as literals, there are no parameters
    gen_context(backend::source_generator &s, const backend::command_queue &q, const std::string &p = "prm")
        : src(s), queue(q), prefix(p), pos(0), seen(own_seen), symbolic(false) {}
    /// Child context with its own numbering (tagged terminals, the x operand of a sparse product).
    gen_context(gen_context &parent, const std::string &p)
        : src(parent.src), queue(parent.queue), prefix(p), pos(0), seen(parent.seen), symbolic(parent.symbolic) {}
    /// Recording context of the symbolic generator: the set of emitted functions outlives the context.
    gen_context(backend::source_generator &s, const backend::command_queue &q, std::set<std::string> &seen_functions)
        : src(s), queue(q), prefix("prm"), pos(0), seen(seen_functions), symbolic(true) {}
    gen_context(const gen_context &) = delete;
    std::string next() { std::ostringstream n; n << prefix << "_" << ++pos; return n.str(); }
};

struct arg_context {
    backend::kernel &krn;
    unsigned device;
    size_t offset;                  // start of this device's partition
    std::set<std::string> own_seen;
    std::set<std::string> &seen;
    int pos;
    arg_context(backend::kernel &k, unsigned d, size_t off) : krn(k), device(d), offset(off), seen(own_seen), pos(0) {}
    arg_context(arg_context &parent) : krn(parent.krn), device(parent.device), offset(parent.offset), seen(parent.seen), pos(0) {}
    std::string next(const std::string &prefix = "prm") { std::ostringstream n; n << prefix << "_" << ++pos; return n.str(); }
};

struct prop_context {
    std::vector<backend::command_queue> queue;
    std::vector<size_t> part;
    size_t size;
    prop_context() : size(0) {}
    bool empty() const { return queue.empty(); }
    /// A further sized terminal joins the expression.  With VEXCL_CHECK_SIZES > 0 (a compile-time option of the
    /// reference, operations.hpp:1442-1457) it must agree with what the expression already has.
    void also(size_t nqueues, size_t n) const {
#if defined(VEXCL_CHECK_SIZES) && (VEXCL_CHECK_SIZES > 0)
        precondition(nqueues == 0 || queue.empty() || nqueues == queue.size(), "Incompatible queue lists");
        precondition(n == 0 || size == 0 || n == size, "Incompatible expression sizes");
#else
        (void)nqueues; (void)n;
#endif
    }
};

// ---- node classification ------------------------------------------------------
struct expression_base {};
template <class T> struct is_expr : std::is_base_of<expression_base, typename std::decay<T>::type> {};
template <class T> struct is_scalar : std::is_arithmetic<typename std::decay<T>::type> {};

// kind: 0 = vector expression, 1 = additive transform only (A*x terms),
//       2 = mix of both joined by + / -, -1 = not assignable
template <class E, class Enable = void> struct expr_kind : std::integral_constant<int, 0> {};

/// Scalar literal: a value parameter, never baked into the source
/// (operations.hpp:167-175,228-236).
template <class T>
struct scalar_terminal : expression_base {
    typedef T value_type;
    T v;
    explicit scalar_terminal(T v) : v(v) {}
    void preamble(gen_context &c) const { c.next(); }
    void params(gen_context &c) const { c.src.template parameter<T>(c.next()); }
    void local_init(gen_context &c) const { c.next(); }
    void emit(gen_context &c) const {
        const std::string n = c.next();
        if (!c.symbolic) { c.src << n; return; }
        std::ostringstream lit;                 // recorded code: the value itself (generator.hpp:374-388 of the reference)
        lit.precision(std::numeric_limits<T>::max_digits10);
        if (std::is_floating_point<T>::value) lit << std::scientific;
        lit << +v;
        c.src << lit.str();
    }
    void set_args(arg_context &a) const { a.next(); a.krn.push_arg(v); }
    void get_props(prop_context &) const {}
};

// how an operand is stored inside a node: scalars by value in a
// scalar_terminal, light nodes by value, heavy objects (vex::vector) through
// the reference type they nominate
template <class T, class Enable = void> struct as_expr;
template <class T> struct as_expr<T, typename std::enable_if<is_scalar<T>::value>::type> {
    typedef scalar_terminal<typename std::decay<T>::type> type;
    static type get(const T &t) { return type(t); }
};
template <class T, class = void> struct has_ref_type : std::false_type {};
template <class T> struct has_ref_type<T, typename std::enable_if<!std::is_void<typename T::expr_ref_type>::value>::type> : std::true_type {};
template <class T> struct as_expr<T, typename std::enable_if<is_expr<T>::value && has_ref_type<typename std::decay<T>::type>::value>::type> {
    typedef typename std::decay<T>::type::expr_ref_type type;
    static type get(const T &t) { return type(t); }
};
template <class T> struct as_expr<T, typename std::enable_if<is_expr<T>::value && !has_ref_type<typename std::decay<T>::type>::value>::type> {
    typedef typename std::decay<T>::type type;
    static const type &get(const T &t) { return t; }
};
template <class T> using as_expr_t = typename as_expr<T>::type;

// operands that are neither nodes nor arithmetic values (std::tuple of per-component
// operands -- multivector.hpp; std::integral_constant -- constants.hpp) opt in here
template <class T> struct is_extra_operand : std::false_type {};
template <class T> struct is_operand : std::integral_constant<bool,
    is_expr<T>::value || is_scalar<T>::value || is_extra_operand<typename std::decay<T>::type>::value> {};

// ---- operator tags -----------------------------------------------------------
namespace tag {
#define VEXCL_TAG(name, op) struct name { static const char *str() { return #op; } }
    VEXCL_TAG(plus, +); VEXCL_TAG(minus, -); VEXCL_TAG(multiplies, *); VEXCL_TAG(divides, /); VEXCL_TAG(modulus, %);
    VEXCL_TAG(shift_left, <<); VEXCL_TAG(shift_right, >>);
    VEXCL_TAG(less, <); VEXCL_TAG(greater, >); VEXCL_TAG(less_equal, <=); VEXCL_TAG(greater_equal, >=);
    VEXCL_TAG(equal_to, ==); VEXCL_TAG(not_equal_to, !=);
    VEXCL_TAG(logical_and, &&); VEXCL_TAG(logical_or, ||);
    VEXCL_TAG(bitwise_and, &); VEXCL_TAG(bitwise_or, |); VEXCL_TAG(bitwise_xor, ^);
    VEXCL_TAG(negate, -); VEXCL_TAG(unary_plus, +); VEXCL_TAG(logical_not, !); VEXCL_TAG(complement, ~);
#undef VEXCL_TAG
    template <class T> struct is_comparison : std::false_type {};
    template <> struct is_comparison<less> : std::true_type {};
    template <> struct is_comparison<greater> : std::true_type {};
    template <> struct is_comparison<less_equal> : std::true_type {};
    template <> struct is_comparison<greater_equal> : std::true_type {};
    template <> struct is_comparison<equal_to> : std::true_type {};
    template <> struct is_comparison<not_equal_to> : std::true_type {};
    template <> struct is_comparison<logical_and> : std::true_type {};
    template <> struct is_comparison<logical_or> : std::true_type {};
}

// result types (operations.hpp:1723-1812): common type for arithmetic,
// cl_long for comparisons and logical operators, left type for shifts
template <class Tag, class L, class R, class Enable = void>
struct binary_result { typedef typename std::common_type<L, R>::type type; };
template <class Tag, class L, class R>
struct binary_result<Tag, L, R, typename std::enable_if<tag::is_comparison<Tag>::value>::type> { typedef cl_long type; };
/// A short vector combined with a scalar (or with itself) stays that short vector.
template <class Tag, class L, class R>
struct binary_result<Tag, L, R, typename std::enable_if<
    !tag::is_comparison<Tag>::value && !std::is_same<Tag, tag::shift_left>::value && !std::is_same<Tag, tag::shift_right>::value &&
    (is_cl_vector<L>::value || is_cl_vector<R>::value)>::type>
{   // the vector length of the vector operand over the common type of the scalars (double * cl_int2 is a cl_double2)
    typedef typename cl_vector_of<
        typename std::common_type<typename cl_scalar_of<L>::type, typename cl_scalar_of<R>::type>::type,
        (cl_vector_length<L>::value > cl_vector_length<R>::value ? cl_vector_length<L>::value : cl_vector_length<R>::value)>::type type;
};
/// pointer + offset, pointer - offset (vector_pointer.hpp: *(p + i)).
template <class L, class R> struct binary_result<tag::plus, L *, R, typename std::enable_if<std::is_integral<R>::value>::type> { typedef L *type; };
template <class L, class R> struct binary_result<tag::minus, L *, R, typename std::enable_if<std::is_integral<R>::value>::type> { typedef L *type; };
template <class L, class R> struct binary_result<tag::plus, L, R *, typename std::enable_if<std::is_integral<L>::value>::type> { typedef R *type; };
template <class L, class R> struct binary_result<tag::shift_left, L, R, void> { typedef L type; };
template <class L, class R> struct binary_result<tag::shift_right, L, R, void> { typedef L type; };

template <class Tag, class L, class R>
struct binary_expr : expression_base {
    typedef typename binary_result<Tag, typename L::value_type, typename R::value_type>::type value_type;
    typedef Tag tag_type; typedef L left_type; typedef R right_type;
    L l; R r;
    binary_expr(const L &l, const R &r) : l(l), r(r) {}
    void preamble(gen_context &c) const { l.preamble(c); r.preamble(c); }
    void params(gen_context &c) const { l.params(c); r.params(c); }
    void local_init(gen_context &c) const { l.local_init(c); r.local_init(c); }
    void emit(gen_context &c) const { c.src << "( "; l.emit(c); c.src << " " << Tag::str() << " "; r.emit(c); c.src << " )"; }
    void set_args(arg_context &a) const { l.set_args(a); r.set_args(a); }
    void get_props(prop_context &p) const { l.get_props(p); r.get_props(p); }
};

template <class Tag, class A>
struct unary_expr : expression_base {
    typedef typename std::conditional<std::is_same<Tag, tag::logical_not>::value, cl_long, typename A::value_type>::type value_type;
    typedef Tag tag_type; typedef A arg_type;
    A a;
    explicit unary_expr(const A &a) : a(a) {}
    void preamble(gen_context &c) const { a.preamble(c); }
    void params(gen_context &c) const { a.params(c); }
    void local_init(gen_context &c) const { a.local_init(c); }
    void emit(gen_context &c) const { c.src << "( " << Tag::str() << "( "; a.emit(c); c.src << " ) )"; }
    void set_args(arg_context &s) const { a.set_args(s); }
    void get_props(prop_context &p) const { a.get_props(p); }
};

// ---- tuple helpers for n-ary nodes -------------------------------------------
template <class Tuple, class F, size_t... I>
void tuple_for_each_impl(const Tuple &t, F &&f, std::index_sequence<I...>) {
    int dummy[] = {0, (f(std::get<I>(t), I), 0)...};
    (void)dummy;
}
template <class... T, class F>
void tuple_for_each(const std::tuple<T...> &t, F &&f) {
    tuple_for_each_impl(t, std::forward<F>(f), std::index_sequence_for<T...>());
}

/// Call of a device function: user functions (VEX_FUNCTION) and builtins.
/// F provides name(), and for user functions define(source_generator&).
template <class F, class R, class... Args>
struct function_call : expression_base {
    typedef R value_type;
    std::tuple<Args...> args;
    explicit function_call(const Args &...a) : args(a...) {}
    void preamble(gen_context &c) const {
        F::preamble(c);
        tuple_for_each(args, [&c](const auto &a, size_t) { a.preamble(c); });
    }
    void params(gen_context &c) const { tuple_for_each(args, [&c](const auto &a, size_t) { a.params(c); }); }
    void local_init(gen_context &c) const { tuple_for_each(args, [&c](const auto &a, size_t) { a.local_init(c); }); }
    void emit(gen_context &c) const {
        c.src << F::name() << "( ";
        tuple_for_each(args, [&c](const auto &a, size_t i) { if (i) c.src << ", "; a.emit(c); });
        c.src << " )";
    }
    void set_args(arg_context &s) const { tuple_for_each(args, [&s](const auto &a, size_t) { a.set_args(s); }); }
    void get_props(prop_context &p) const { tuple_for_each(args, [&p](const auto &a, size_t) { a.get_props(p); }); }
};

/// ( c ? a : b ) -- vex::if_else (operations.hpp ternary, tests/vector_arithmetics.cpp:238-250).
/// Type of ( c ? a : b ): the common type; for pointers to different types, a pointer to the
/// common type of what they point to (type deduction only: `*if_else(c, &x, &y)`, tests/deduce.cpp:126).
template <class A, class B, class Enable = void> struct ternary_result { typedef typename std::common_type<A, B>::type type; };
template <class A, class B>
struct ternary_result<A *, B *, typename std::enable_if<!std::is_same<A, B>::value>::type> { typedef typename std::common_type<A, B>::type *type; };

template <class C, class A, class B>
struct ternary_expr : expression_base {
    typedef typename ternary_result<typename A::value_type, typename B::value_type>::type value_type;
    C c_; A a; B b;
    ternary_expr(const C &c, const A &a, const B &b) : c_(c), a(a), b(b) {}
    void preamble(gen_context &c) const { c_.preamble(c); a.preamble(c); b.preamble(c); }
    void params(gen_context &c) const { c_.params(c); a.params(c); b.params(c); }
    void local_init(gen_context &c) const { c_.local_init(c); a.local_init(c); b.local_init(c); }
    void emit(gen_context &c) const { c.src << "( "; c_.emit(c); c.src << " ? "; a.emit(c); c.src << " : "; b.emit(c); c.src << " )"; }
    void set_args(arg_context &s) const { c_.set_args(s); a.set_args(s); b.set_args(s); }
    void get_props(prop_context &p) const { c_.get_props(p); a.get_props(p); b.get_props(p); }
};

/// *p and p[i] for pointer-valued expressions; usable on the left of an assignment through vex::tie.
template <class P>
struct deref_expr : expression_base {
    typedef typename std::remove_cv<typename std::remove_pointer<typename P::value_type>::type>::type value_type;
    P p;
    explicit deref_expr(const P &p) : p(p) {}
    void preamble(gen_context &c) const { p.preamble(c); }
    void params(gen_context &c) const { p.params(c); }
    void local_init(gen_context &c) const { p.local_init(c); }
    void emit(gen_context &c) const { c.src << "( *"; p.emit(c); c.src << " )"; }
    void set_args(arg_context &a) const { p.set_args(a); }
    void get_props(prop_context &q) const { p.get_props(q); }
};

/// &view: the address of the element a writable view designates (eval.cpp of the reference's tests:
/// `atomic_add(&permutation(i)(y), 1)`).
template <class V>
struct address_expr : expression_base {
    typedef typename V::value_type *value_type;
    V v;
    explicit address_expr(const V &v) : v(v) {}
    void preamble(gen_context &c) const { v.preamble(c); }
    void params(gen_context &c) const { v.params(c); }
    void local_init(gen_context &c) const { v.local_init(c); }
    void emit(gen_context &c) const { c.src << "( &"; v.emit(c); c.src << " )"; }
    void set_args(arg_context &a) const { v.set_args(a); }
    void get_props(prop_context &p) const { v.get_props(p); }
};

// ---- additive transforms: A*x terms (operations.hpp:759-776) ------------------
struct additive_transform_base : expression_base {};

template <class V, class Enable = void> struct element_type_of { typedef typename V::value_type type; };
template <class V> struct element_type_of<V, typename std::enable_if<!std::is_void<typename V::sub_value_type>::value>::type> {
    typedef typename V::sub_value_type type;    // multivector: the type of one component's elements
};

template <class M, class V>
struct additive_operator : additive_transform_base {
    typedef typename element_type_of<V>::type value_type;
    const M &A; const V &x;
    additive_operator(const M &A, const V &x) : A(A), x(x) {}
    template <class W> void apply(W &y, double scale, bool append) const {
        A.apply(x, y, static_cast<typename M::scalar_type>(scale), append);
    }
    // never part of generated source
    void preamble(gen_context &) const {} void params(gen_context &) const {} void local_init(gen_context &) const {}
    void emit(gen_context &) const {} void set_args(arg_context &) const {} void get_props(prop_context &) const {}
};

template <class E> struct is_transform : std::is_base_of<additive_transform_base, E> {};
template <class E> struct is_scalar_terminal : std::false_type {};
template <class T> struct is_scalar_terminal<scalar_terminal<T>> : std::true_type {};

template <class M, class V> struct expr_kind<additive_operator<M, V>> : std::integral_constant<int, 1> {};

constexpr int join_additive(int l, int r) { return (l < 0 || r < 0) ? -1 : (l == r && l != 2) ? l : 2; }
constexpr int join_product(int l, int r, bool lscalar, bool rscalar) {
    return (l == 0 && r == 0) ? 0 : ((l == 1 && rscalar) || (r == 1 && lscalar)) ? 1 : -1;
}
constexpr int join_other(int l, int r) { return (l == 0 && r == 0) ? 0 : -1; }

template <class L, class R> struct expr_kind<binary_expr<tag::plus, L, R>>
    : std::integral_constant<int, join_additive(expr_kind<L>::value, expr_kind<R>::value)> {};
template <class L, class R> struct expr_kind<binary_expr<tag::minus, L, R>>
    : std::integral_constant<int, join_additive(expr_kind<L>::value, expr_kind<R>::value)> {};
template <class L, class R> struct expr_kind<binary_expr<tag::multiplies, L, R>>
    : std::integral_constant<int, join_product(expr_kind<L>::value, expr_kind<R>::value,
            is_scalar_terminal<L>::value, is_scalar_terminal<R>::value)> {};
template <class Tag, class L, class R> struct expr_kind<binary_expr<Tag, L, R>>
    : std::integral_constant<int, join_other(expr_kind<L>::value, expr_kind<R>::value)> {};
template <class A> struct expr_kind<unary_expr<tag::negate, A>> : std::integral_constant<int, expr_kind<A>::value> {};
template <class Tag, class A> struct expr_kind<unary_expr<Tag, A>>
    : std::integral_constant<int, expr_kind<A>::value == 0 ? 0 : -1> {};

template <class P> struct expr_kind<deref_expr<P>> : std::integral_constant<int, expr_kind<P>::value == 0 ? 0 : -1> {};

template <class... A> struct all_vector_kind : std::true_type {};
template <class H, class... T> struct all_vector_kind<H, T...>
    : std::integral_constant<bool, expr_kind<H>::value == 0 && all_vector_kind<T...>::value> {};
template <class F, class R, class... A> struct expr_kind<function_call<F, R, A...>>
    : std::integral_constant<int, all_vector_kind<A...>::value ? 0 : -1> {};
template <class C, class A, class B> struct expr_kind<ternary_expr<C, A, B>>
    : std::integral_constant<int, all_vector_kind<C, A, B>::value ? 0 : -1> {};

// ---- multi-expressions (multivector.hpp) --------------------------------------
// Number of components an expression carries: 0 for ordinary vector expressions,
// N when it contains a multivector<T, N> or an N-tuple operand (multivector.hpp:70-150
// of the reference: multivector grammar).
constexpr size_t join_dim(size_t a, size_t b) { return a == 0 ? b : a; }
template <class E, class Enable = void> struct mv_dim : std::integral_constant<size_t, 0> {};
template <class Tag, class L, class R> struct mv_dim<binary_expr<Tag, L, R>>
    : std::integral_constant<size_t, join_dim(mv_dim<L>::value, mv_dim<R>::value)> {
    static_assert(mv_dim<L>::value == 0 || mv_dim<R>::value == 0 || mv_dim<L>::value == mv_dim<R>::value,
            "operands of a multi-expression have different numbers of components");
};
template <class Tag, class A> struct mv_dim<unary_expr<Tag, A>> : mv_dim<A> {};
template <class... A> struct mv_dim_all : std::integral_constant<size_t, 0> {};
template <class H, class... T> struct mv_dim_all<H, T...>
    : std::integral_constant<size_t, join_dim(mv_dim<H>::value, mv_dim_all<T...>::value)> {};
template <class F, class R, class... A> struct mv_dim<function_call<F, R, A...>> : mv_dim_all<A...> {};
template <class C, class A, class B> struct mv_dim<ternary_expr<C, A, B>> : mv_dim_all<C, A, B> {};

/// Component I of a multi-expression, as an ordinary expression; ordinary
/// sub-expressions are shared by all components (multivector.hpp subexpression extraction).
template <size_t I, class E, class Enable = void> struct component_of {
    typedef E type;
    static const E &get(const E &e) { return e; }
};
template <size_t I, class Tag, class L, class R>
struct component_of<I, binary_expr<Tag, L, R>, typename std::enable_if<(mv_dim<binary_expr<Tag, L, R>>::value > 0)>::type> {
    typedef binary_expr<Tag, typename component_of<I, L>::type, typename component_of<I, R>::type> type;
    static type get(const binary_expr<Tag, L, R> &e) { return type(component_of<I, L>::get(e.l), component_of<I, R>::get(e.r)); }
};
template <size_t I, class Tag, class A>
struct component_of<I, unary_expr<Tag, A>, typename std::enable_if<(mv_dim<A>::value > 0)>::type> {
    typedef unary_expr<Tag, typename component_of<I, A>::type> type;
    static type get(const unary_expr<Tag, A> &e) { return type(component_of<I, A>::get(e.a)); }
};
template <size_t I, class F, class R, class... A>
struct component_of<I, function_call<F, R, A...>, typename std::enable_if<(mv_dim_all<A...>::value > 0)>::type> {
    typedef function_call<F, R, typename component_of<I, A>::type...> type;
    template <size_t... K>
    static type make(const function_call<F, R, A...> &e, std::index_sequence<K...>) {
        return type(component_of<I, A>::get(std::get<K>(e.args))...);
    }
    static type get(const function_call<F, R, A...> &e) { return make(e, std::index_sequence_for<A...>()); }
};
template <size_t I, class C, class A, class B>
struct component_of<I, ternary_expr<C, A, B>, typename std::enable_if<(mv_dim_all<C, A, B>::value > 0)>::type> {
    typedef ternary_expr<typename component_of<I, C>::type, typename component_of<I, A>::type, typename component_of<I, B>::type> type;
    static type get(const ternary_expr<C, A, B> &e) {
        return type(component_of<I, C>::get(e.c_), component_of<I, A>::get(e.a), component_of<I, B>::get(e.b));
    }
};

/// Hook: an expression that is, as a whole, one product with a hand-written kernel
/// (e.g. `y += A * x` with A a SpMatCCSR) nominates it here; it remains an ordinary
/// terminal inside larger expressions.
template <class E, class Enable = void> struct direct_assign : std::false_type {};

/// Applies every A*x term of an additive expression to y
/// (operations.hpp:1475-1576: negations pushed to the leaves, first term SET or ADD, rest ADD).
template <class W, class E>
void apply_transforms(W &y, const E &e, double scale, bool &append) {
    if constexpr (expr_kind<E>::value == 0) {
        (void)y; (void)e; (void)scale; (void)append;             // vector part: handled by the caller
    } else if constexpr (is_transform<E>::value) {
        e.apply(y, scale, append);
        append = true;
    } else if constexpr (std::is_same<typename E::tag_type, tag::negate>::value) {
        apply_transforms(y, e.a, -scale, append);
    } else if constexpr (std::is_same<typename E::tag_type, tag::plus>::value) {
        apply_transforms(y, e.l, scale, append);
        apply_transforms(y, e.r, scale, append);
    } else if constexpr (std::is_same<typename E::tag_type, tag::minus>::value) {
        apply_transforms(y, e.l, scale, append);
        apply_transforms(y, e.r, -scale, append);
    } else {                                                       // scalar * transform (is_scalable)
        if constexpr (is_scalar_terminal<typename E::left_type>::value)
            apply_transforms(y, e.r, scale * static_cast<double>(e.l.v), append);
        else
            apply_transforms(y, e.l, scale * static_cast<double>(e.r.v), append);
    }
}

/// The expression with its A*x terms removed (operations.hpp:514-564 extractors).
template <class E>
auto vector_part(const E &e) {
    static_assert(expr_kind<E>::value == 0 || expr_kind<E>::value == 2, "no vector part");
    if constexpr (expr_kind<E>::value == 0) {
        return e;
    } else if constexpr (std::is_same<typename E::tag_type, tag::negate>::value) {
        auto v = vector_part(e.a);
        return unary_expr<tag::negate, decltype(v)>(v);
    } else {
        typedef typename E::left_type L; typedef typename E::right_type R;
        constexpr bool is_minus = std::is_same<typename E::tag_type, tag::minus>::value;
        if constexpr (expr_kind<L>::value == 1) {
            auto v = vector_part(e.r);
            if constexpr (is_minus) return unary_expr<tag::negate, decltype(v)>(v); else return v;
        } else if constexpr (expr_kind<R>::value == 1) {
            return vector_part(e.l);
        } else {
            auto a = vector_part(e.l); auto b = vector_part(e.r);
            return binary_expr<typename E::tag_type, decltype(a), decltype(b)>(a, b);
        }
    }
}

/// Number of terminal positions a node consumes (its traversal passes all advance `pos` alike).
template <class Node>
int count_terminals(const Node &node, const backend::command_queue &q) {
    backend::source_generator scratch;
    gen_context c(scratch, q);
    node.local_init(c);
    return c.pos;
}

// ---- the fused elementwise kernel (operations.hpp:1818-1897) -------------------
/// Source of the kernel `lhs OP rhs`.  Shape of the reference's kernel (SURVEY A.1:
/// one parameter per terminal in traversal order, lhs first; the statement text is
/// the reference's), with ONE change for MI355X: a lane handles TWO elements per
/// trip of the grid-stride loop (idx and idx + grid_size, both fully coalesced).  Both
/// right-hand sides are evaluated -- all their loads issued -- before either result is
/// stored, so a lane keeps twice the memory requests in flight; gather-like terminals
/// (permutation, sparse / CCSR products) are latency-bound without it.  The pair loop runs
/// while both elements exist; a lane's leftover element is handled once after it.
template <class OP, class LHS, class RHS>
std::string assignment_source(const LHS &lhs, const RHS &rhs, const backend::command_queue &q) {
    backend::source_generator source(q);
    { gen_context c(source, q); lhs.preamble(c); rhs.preamble(c); }
    source.begin_kernel("vexcl_vector_kernel");
    source.begin_kernel_parameters();
    source.template parameter<size_t>("n");
    { gen_context c(source, q); lhs.params(c); rhs.params(c); }
    source.end_kernel_parameters();
    const std::string R = type_name<typename RHS::value_type>();
    source.new_line() << "const ulong grid_size = blockDim.x * (ulong)gridDim.x;";
    source.new_line() << "ulong vex_i = blockDim.x * (ulong)blockIdx.x + threadIdx.x;";
    // pairs while BOTH elements exist ...
    source.new_line() << "for(; vex_i + grid_size < n; vex_i += 2 * grid_size)";
    source.open("{");
    source.new_line() << R << " vex_r0, vex_r1;";
    for (int e = 0; e < 2; ++e) {
        source.open("{");
        source.new_line() << "const ulong idx = " << (e ? "vex_i + grid_size" : "vex_i") << ";";
        { gen_context c(source, q); lhs.local_init(c); rhs.local_init(c); }
        source.new_line() << "vex_r" << e << " = ";
        { gen_context c(source, q); c.pos = count_terminals(lhs, q); rhs.emit(c); }
        source << ";";
        source.close("}");
    }
    for (int e = 0; e < 2; ++e) {
        source.open("{");
        source.new_line() << "const ulong idx = " << (e ? "vex_i + grid_size" : "vex_i") << ";";
        source.new_line();
        { gen_context c(source, q); lhs.emit(c); }
        source << " " << OP::string() << " vex_r" << e << ";";
        source.close("}");
    }
    source.close("}");
    // ... then at most ONE leftover element per lane: every element is evaluated exactly once, as in the reference's
    // one-element loop (an expression may have side effects: atomics in a user function, writes through raw pointers)
    source.new_line() << "if (vex_i < n)";
    source.open("{");
    source.new_line() << "const ulong idx = vex_i;";
    { gen_context c(source, q); lhs.local_init(c); rhs.local_init(c); }
    source.new_line() << "const " << R << " vex_r0 = ";
    { gen_context c(source, q); c.pos = count_terminals(lhs, q); rhs.emit(c); }
    source << ";";
    source.new_line();
    { gen_context c(source, q); lhs.emit(c); }
    source << " " << OP::string() << " vex_r0;";
    source.close("}");
    source.end_kernel();
    return source.str();
}

template <class OP, class LHS, class RHS>
void assign_expression(const LHS &lhs, const RHS &rhs,
        const std::vector<backend::command_queue> &queue, const std::vector<size_t> &part)
{
    static_assert(expr_kind<RHS>::value == 0, "expression contains terms that cannot be fused into a kernel");
    static kernel_cache cache;
#if defined(VEXCL_CHECK_SIZES) && (VEXCL_CHECK_SIZES > 0)
    {   // operations.hpp:1824-1840 of the reference
        prop_context p;
        lhs.get_props(p); rhs.get_props(p);
        precondition(p.queue.empty() || p.queue.size() == queue.size(), "Incompatible queue lists");
        precondition(p.size == 0 || p.size == part.back(), "Incompatible expression sizes");
    }
#endif

    for (unsigned d = 0; d < queue.size(); ++d) {
        size_t psize = part[d + 1] - part[d];
        if (!psize) continue;

        auto kernel = cache.find(queue[d]);
        if (kernel == cache.end())
            kernel = cache.insert(queue[d], backend::kernel(queue[d], assignment_source<OP>(lhs, rhs, queue[d]), "vexcl_vector_kernel"));

        backend::kernel &krn = kernel->second;
        krn.push_arg(psize);
        arg_context a(krn, d, part[d]);
        lhs.set_args(a);
        rhs.set_args(a);
        krn.config_streaming(queue[d], psize);
        krn(queue[d]);
    }
}

// ---- y = [c1 *] z +- [c2 *] (A * x): one vector and one product term (round 6) ---------------------------------------
// The reference evaluates `y = z - A * x` as "y = z", then "y -= A * x" (vector.hpp:698-801): two passes over y.  Products that can add
// a vector in the same pass (SpMat::apply_axpby -> include/vexhip.h vexhip_spmat_apply_axpby_f64) take such an expression whole; the
// same shape with make_inline(A * x) in place of A * x lands there too.  SHAPE is decided at compile time (axpby_shape), the scales and
// operands are read off at run time (axpby_collect); anything else, or a product that declines, takes the general route below.
/// A leaf of such an expression: kind 1 = a vector of the target's type (get), kind 2 = a product term (apply), 0 = neither.
template <class E, class Enable = void> struct axpby_leaf : std::integral_constant<int, 0> {};
template <class M, class = void> struct has_apply_axpby : std::false_type {};
template <class M> struct has_apply_axpby<M, typename std::enable_if<M::has_axpby_product>::type> : std::true_type {};
template <class T> class plain_vector_of { public: typedef void type; };        // vector.hpp: plain_vector_of<vector<T>>::type = T

template <class E, class Enable = void> struct axpby_shape { static constexpr int vectors = 0, terms = 0; static constexpr bool ok = false; };
template <class E> struct axpby_shape<E, typename std::enable_if<axpby_leaf<E>::value == 1>::type> { static constexpr int vectors = 1, terms = 0; static constexpr bool ok = true; };
template <class E> struct axpby_shape<E, typename std::enable_if<axpby_leaf<E>::value == 2>::type> { static constexpr int vectors = 0, terms = 1; static constexpr bool ok = true; };
template <class L, class R> struct axpby_shape<binary_expr<tag::plus, L, R>, void> {
    static constexpr int vectors = axpby_shape<L>::vectors + axpby_shape<R>::vectors, terms = axpby_shape<L>::terms + axpby_shape<R>::terms;
    static constexpr bool ok = axpby_shape<L>::ok && axpby_shape<R>::ok; };
template <class L, class R> struct axpby_shape<binary_expr<tag::minus, L, R>, void> : axpby_shape<binary_expr<tag::plus, L, R>, void> {};
template <class S, class R> struct axpby_shape<binary_expr<tag::multiplies, scalar_terminal<S>, R>, void> : axpby_shape<R> {};
template <class L, class S> struct axpby_shape<binary_expr<tag::multiplies, L, scalar_terminal<S>>, typename std::enable_if<!is_scalar_terminal<L>::value>::type> : axpby_shape<L> {};
template <class A> struct axpby_shape<unary_expr<tag::negate, A>, void> : axpby_shape<A> {};

template <class T> struct axpby_form {
    const void *z = nullptr; double beta = 0, alpha = 0;
    std::function<bool(double, double)> term;          // (alpha, beta) -> the product took the expression
};
template <class T, class Y, class E> void axpby_collect(axpby_form<T> &f, Y &y, const E &e, double scale);
template <class T, class Y, class E> struct axpby_visit {
    static void go(axpby_form<T> &f, Y &y, const E &e, double scale) {
        if constexpr (axpby_leaf<E>::value == 1) {
            (void)y;
            if constexpr (std::is_same<typename E::value_type, T>::value) { f.z = axpby_leaf<E>::get(e); f.beta = scale; } else { (void)e; (void)scale; }
        }
        else { f.alpha = scale; f.term = [&f, &y, &e](double a, double b) { return axpby_leaf<E>::apply(e, y, a, f.z, b); }; }
    }
};
template <class T, class Y, class L, class R> struct axpby_visit<T, Y, binary_expr<tag::plus, L, R>> {
    static void go(axpby_form<T> &f, Y &y, const binary_expr<tag::plus, L, R> &e, double s) { axpby_collect(f, y, e.l, s); axpby_collect(f, y, e.r, s); } };
template <class T, class Y, class L, class R> struct axpby_visit<T, Y, binary_expr<tag::minus, L, R>> {
    static void go(axpby_form<T> &f, Y &y, const binary_expr<tag::minus, L, R> &e, double s) { axpby_collect(f, y, e.l, s); axpby_collect(f, y, e.r, -s); } };
template <class T, class Y, class S, class R> struct axpby_visit<T, Y, binary_expr<tag::multiplies, scalar_terminal<S>, R>> {
    static void go(axpby_form<T> &f, Y &y, const binary_expr<tag::multiplies, scalar_terminal<S>, R> &e, double s) { axpby_collect(f, y, e.r, s * static_cast<double>(e.l.v)); } };
template <class T, class Y, class L, class S> struct axpby_visit<T, Y, binary_expr<tag::multiplies, L, scalar_terminal<S>>> {
    static void go(axpby_form<T> &f, Y &y, const binary_expr<tag::multiplies, L, scalar_terminal<S>> &e, double s) { axpby_collect(f, y, e.l, s * static_cast<double>(e.r.v)); } };
template <class T, class Y, class A> struct axpby_visit<T, Y, unary_expr<tag::negate, A>> {
    static void go(axpby_form<T> &f, Y &y, const unary_expr<tag::negate, A> &e, double s) { axpby_collect(f, y, e.a, -s); } };
template <class T, class Y, class E> void axpby_collect(axpby_form<T> &f, Y &y, const E &e, double scale) { axpby_visit<T, Y, E>::go(f, y, e, scale); }

template <class M, class V> struct axpby_leaf<additive_operator<M, V>, typename std::enable_if<has_apply_axpby<M>::value && !std::is_void<typename plain_vector_of<V>::type>::value>::type>
    : std::integral_constant<int, 2> {
    template <class Y> static bool apply(const additive_operator<M, V> &e, Y &y, double alpha, const void *z, double beta) {
        if constexpr (std::is_same<Y, V>::value) return e.A.apply_axpby(e.x, y, alpha, *static_cast<const Y *>(z), beta);
        else { (void)e; (void)y; (void)alpha; (void)z; (void)beta; return false; }
    }
};

/// Assignment of any assignable expression to an lvalue terminal:
/// fused kernel for the vector part, SpMat::apply for every A*x term
/// (vector.hpp:698-801).
template <class OP, class LHS, class W, class Expr>
void assign_any(const LHS &lhs, W &target, const Expr &expr,
        const std::vector<backend::command_queue> &queue, const std::vector<size_t> &part)
{
    constexpr int kind = expr_kind<Expr>::value;
    static_assert(kind >= 0, "this expression cannot be assigned: A*x terms may only be added, subtracted or scaled");
    typedef typename plain_vector_of<W>::type target_value;
    if constexpr (std::is_same<OP, assign::SET>::value && !std::is_void<target_value>::value && axpby_shape<Expr>::ok
                  && axpby_shape<Expr>::vectors == 1 && axpby_shape<Expr>::terms == 1) {
        // one vector and one product term: the product adds the vector in its own pass, if it can
        axpby_form<target_value> f;
        axpby_collect(f, target, expr, 1.0);
        if (f.z && f.term && f.term(f.alpha, f.beta)) return;
    }
    if constexpr (kind == 0) {
        constexpr bool lin = std::is_same<OP, assign::SET>::value || std::is_same<OP, assign::ADD>::value || std::is_same<OP, assign::SUB>::value;
        if constexpr (direct_assign<Expr>::value && lin) {
            // the whole right-hand side is one product with a hand-written kernel
            (void)lhs; (void)queue; (void)part;
            direct_assign<Expr>::apply(target, expr, std::is_same<OP, assign::SUB>::value ? -1.0 : 1.0,
                                       !std::is_same<OP, assign::SET>::value);
        } else {
            (void)target;
            assign_expression<OP>(lhs, expr, queue, part);
        }
    } else {
        constexpr bool set = std::is_same<OP, assign::SET>::value;
        constexpr bool add = std::is_same<OP, assign::ADD>::value;
        constexpr bool sub = std::is_same<OP, assign::SUB>::value;
        static_assert(set || add || sub, "A*x terms support only =, += and -=");
        bool append = !set;
        if constexpr (kind == 2) {
            assign_expression<OP>(lhs, vector_part(expr), queue, part);
            append = true;
        }
        apply_transforms(target, expr, sub ? -1.0 : 1.0, append);
    }
}

} // namespace detail

// ---- operators ----------------------------------------------------------------
// Results are CONST prvalues, as Boost.Proto's are in the reference: that is what lets
// `std::tie(x + y, x - y)` bind them (std::tie takes lvalue references; a const prvalue
// binds to `const E &`), tests/multivector_arithmetics.cpp:68.
// Defined in vex::detail, where the node types live (ADL), and re-exported to vex::
// for operands that are vex::vector / tagged terminals.
namespace detail {
#define VEXCL_BINARY_OPERATOR(tagname, op)                                                                     \
    template <class L, class R>                                                                                \
    typename std::enable_if<                                                                                   \
        (is_expr<L>::value || is_expr<R>::value) &&                                            \
        is_operand<L>::value && is_operand<R>::value,                                          \
        const binary_expr<tag::tagname, as_expr_t<L>, as_expr_t<R>>>::type     \
    operator op(const L &l, const R &r) {                                                                      \
        return binary_expr<tag::tagname, as_expr_t<L>, as_expr_t<R>>(          \
                as_expr<L>::get(l), as_expr<R>::get(r));                                       \
    }

VEXCL_BINARY_OPERATOR(plus, +)
VEXCL_BINARY_OPERATOR(minus, -)
VEXCL_BINARY_OPERATOR(multiplies, *)
VEXCL_BINARY_OPERATOR(divides, /)
VEXCL_BINARY_OPERATOR(modulus, %)
VEXCL_BINARY_OPERATOR(shift_left, <<)
VEXCL_BINARY_OPERATOR(shift_right, >>)
VEXCL_BINARY_OPERATOR(less, <)
VEXCL_BINARY_OPERATOR(greater, >)
VEXCL_BINARY_OPERATOR(less_equal, <=)
VEXCL_BINARY_OPERATOR(greater_equal, >=)
VEXCL_BINARY_OPERATOR(equal_to, ==)
VEXCL_BINARY_OPERATOR(not_equal_to, !=)
VEXCL_BINARY_OPERATOR(logical_and, &&)
VEXCL_BINARY_OPERATOR(logical_or, ||)
VEXCL_BINARY_OPERATOR(bitwise_and, &)
VEXCL_BINARY_OPERATOR(bitwise_or, |)
VEXCL_BINARY_OPERATOR(bitwise_xor, ^)
#undef VEXCL_BINARY_OPERATOR

#define VEXCL_UNARY_OPERATOR(tagname, op)                                                                      \
    template <class A>                                                                                         \
    typename std::enable_if<is_expr<A>::value,                                                         \
        const unary_expr<tag::tagname, as_expr_t<A>>>::type                            \
    operator op(const A &a) {                                                                                  \
        return unary_expr<tag::tagname, as_expr_t<A>>(as_expr<A>::get(a));     \
    }
VEXCL_UNARY_OPERATOR(negate, -)
VEXCL_UNARY_OPERATOR(unary_plus, +)
VEXCL_UNARY_OPERATOR(logical_not, !)
VEXCL_UNARY_OPERATOR(complement, ~)
#undef VEXCL_UNARY_OPERATOR

/// Dereference of a pointer-valued expression.
template <class A>
typename std::enable_if<is_expr<A>::value && std::is_pointer<typename as_expr_t<A>::value_type>::value,
    const deref_expr<as_expr_t<A>>>::type
operator*(const A &a) { return deref_expr<as_expr_t<A>>(as_expr<A>::get(a)); }

/// Elementwise ternary: if_else(cond, a, b).
template <class C, class A, class B>
typename std::enable_if<is_operand<C>::value && is_operand<A>::value && is_operand<B>::value &&
    (is_expr<C>::value || is_expr<A>::value || is_expr<B>::value),
    const ternary_expr<as_expr_t<C>, as_expr_t<A>, as_expr_t<B>>>::type
if_else(const C &c, const A &a, const B &b) {
    return ternary_expr<as_expr_t<C>, as_expr_t<A>, as_expr_t<B>>(
            as_expr<C>::get(c), as_expr<A>::get(a), as_expr<B>::get(b));
}

} // namespace detail

using detail::operator+;
using detail::operator-;
using detail::operator*;
using detail::operator/;
using detail::operator%;
using detail::operator<<;
using detail::operator>>;
using detail::operator<;
using detail::operator>;
using detail::operator<=;
using detail::operator>=;
using detail::operator==;
using detail::operator!=;
using detail::operator&&;
using detail::operator||;
using detail::operator&;
using detail::operator|;
using detail::operator^;
using detail::operator!;
using detail::operator~;
using detail::if_else;

/// Queue list, partitioning and size of an expression (operations.hpp:1411-1460).
template <class Expr>
void get_expression_properties(const Expr &expr, std::vector<backend::command_queue> &queue,
        std::vector<size_t> &part, size_t &size)
{
    detail::prop_context p;
    detail::as_expr<Expr>::get(expr).get_props(p);
    queue = p.queue; part = p.part; size = p.size;
}

namespace detail {
/// The value type an expression (or a plain value used as one) evaluates to (operations.hpp:1680-1812).
template <class Expr> struct return_type { typedef typename as_expr_t<Expr>::value_type type; };
}

/// Tag of the reference's vector terminal (operations.hpp:560-580 there); kept for trait queries.
struct vector_terminal {};
namespace traits {
/// What may stand as a leaf of a vector expression: values of device-native types and expression nodes.
template <class T, class Enable = void>
struct is_vector_expr_terminal : std::integral_constant<bool, is_cl_native<T>::value || detail::is_expr<T>::value> {};
template <> struct is_vector_expr_terminal<vector_terminal, void> : std::true_type {};
}

/// (queues, size) of an expression or a tuple of expressions (operations.hpp:2370-2385 in the reference).
template <class Expr>
std::tuple<std::vector<backend::command_queue>, size_t> expression_properties(const Expr &expr) {
    std::vector<backend::command_queue> queue; std::vector<size_t> part; size_t size = 0;
    get_expression_properties(expr, queue, part, size);
    return std::make_tuple(queue, size);
}

namespace detail {
template <class Tuple, size_t... I>
void tuple_props(const Tuple &t, prop_context &p, std::index_sequence<I...>) {
    int dummy[] = {0, (as_expr<typename std::tuple_element<I, Tuple>::type>::get(std::get<I>(t)).get_props(p), 0)...};
    (void)dummy;
}
} // namespace detail

/// ... of a tuple of expressions: the first component that carries a queue list decides.
template <class... Expr>
std::tuple<std::vector<backend::command_queue>, size_t> expression_properties(const std::tuple<Expr...> &expr) {
    detail::prop_context p;
    detail::tuple_props(expr, p, std::index_sequence_for<Expr...>());
    return std::make_tuple(p.queue, p.size);
}

} // namespace vex
#endif
