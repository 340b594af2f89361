#ifndef VEXCL_FUNCTION_HPP
#define VEXCL_FUNCTION_HPP
// User-defined device functions and builtin math (reference: vexcl/function.hpp
// :146-226 VEX_FUNCTION* macros, :255-448 builtins, :465-505 abs -> fabs;
// UserFunction base operations.hpp:575-628).  No Boost.Preprocessor: the
// (type, name)(type, name) parameter sequence is walked with the classic
// two-macro ping-pong.
#include <string>
#include <utility>
#include <vector>
#include "operations.hpp"

namespace vex {

/// Base of user functions.  Impl provides name(), body() and
/// params(vector<pair<type,name>>&); optionally dependencies(gen_context&).
template <class Impl, class R>
struct UserFunction {
    typedef R value_type;

    // emitted once per kernel, keyed by NAME (operations.hpp:1068-1094): two
    // functions with the same signature but different names stay distinct
    static void preamble(detail::gen_context &c) {
        const std::string key = "fun:" + Impl::name();
        if (c.seen.count(key)) return;
        c.seen.insert(key);
        Impl::dependencies(c);
        std::vector<std::pair<std::string, std::string>> prm;
        Impl::params(prm);
        c.src.begin_function(type_name<R>(), Impl::name());
        c.src.begin_function_parameters();
        for (const auto &p : prm) c.src.parameter(p.first, p.second);
        c.src.end_function_parameters();
        c.src.new_line() << Impl::body();
        c.src.end_function();
    }
    static void dependencies(detail::gen_context &) {}
    static std::string name() { return Impl::name(); }

    template <class... Args>
    typename std::enable_if<(sizeof...(Args) > 0),
        const detail::function_call<UserFunction<Impl, R>, R, detail::as_expr_t<Args>...>>::type
    operator()(const Args &...args) const {
        return detail::function_call<UserFunction<Impl, R>, R, detail::as_expr_t<Args>...>(
                detail::as_expr<Args>::get(args)...);
    }
};

namespace detail {
    // VEX_FUNCTION_V1: parameters of a signature R(A1, A2, ...) are called prm1, prm2, ...
    template <class Sig> struct signature_params;
    template <class R, class... A> struct signature_params<R(A...)> {
        typedef R result_type;
        static void get(std::vector<std::pair<std::string, std::string>> &p) {
            std::string types[] = {type_name<A>()..., std::string()};
            for (size_t i = 0; i < sizeof...(A); ++i) p.push_back(std::make_pair(types[i], "prm" + std::to_string(i + 1)));
        }
    };
}

} // namespace vex

// ---- sequence walking: (a, b)(c, d)... ---------------------------------------
#define VEXCL_CAT_(a, b) a##b
#define VEXCL_CAT(a, b) VEXCL_CAT_(a, b)

#define VEXCL_PRM_A(t, n) p.push_back(std::make_pair(vex::type_name<t>(), std::string(#n))); VEXCL_PRM_B
#define VEXCL_PRM_B(t, n) p.push_back(std::make_pair(vex::type_name<t>(), std::string(#n))); VEXCL_PRM_A
#define VEXCL_PRM_A_END
#define VEXCL_PRM_B_END
#define VEXCL_PRM_SEQ(seq) VEXCL_CAT(VEXCL_PRM_A seq, _END)

#define VEXCL_DEP_A(f) std::decay<decltype(f)>::type::preamble(c); VEXCL_DEP_B
#define VEXCL_DEP_B(f) std::decay<decltype(f)>::type::preamble(c); VEXCL_DEP_A
#define VEXCL_DEP_A_END
#define VEXCL_DEP_B_END
#define VEXCL_DEP_SEQ(seq) VEXCL_CAT(VEXCL_DEP_A seq, _END)

#define VEX_FUNCTION_SINK(rettype, fname, args, deps, body_str)                                     \
    struct vex_function_##fname : vex::UserFunction<vex_function_##fname, rettype> {                \
        vex_function_##fname() {}                                                                   \
        static std::string name() { return #fname; }                                                \
        static void params(std::vector<std::pair<std::string, std::string>> &p) { (void)p; VEXCL_PRM_SEQ(args) } \
        static void dependencies(vex::detail::gen_context &c) { (void)c; deps }                     \
        static std::string body() { return body_str; }                                              \
    } const fname

/// VEX_FUNCTION(double, sqr, (double, x), return x * x;);
#define VEX_FUNCTION(rettype, fname, args, ...) VEX_FUNCTION_SINK(rettype, fname, args, , #__VA_ARGS__)
/// Body given as a string.
#define VEX_FUNCTION_S(rettype, fname, args, body) VEX_FUNCTION_SINK(rettype, fname, args, , body)
/// With dependencies: VEX_FUNCTION_D(double, f, (double, x), (g)(h), return g(x) + h(x););
#define VEX_FUNCTION_D(rettype, fname, args, deps, ...) VEX_FUNCTION_SINK(rettype, fname, args, VEXCL_DEP_SEQ(deps), #__VA_ARGS__)
#define VEX_FUNCTION_DS(rettype, fname, args, deps, body) VEX_FUNCTION_SINK(rettype, fname, args, VEXCL_DEP_SEQ(deps), body)
#define VEX_FUNCTION_SD VEX_FUNCTION_DS                       /* the reference's spelling (function.hpp:194-203) */
/// Unquoted source text as a string (function.hpp:46).
#define VEX_STRINGIZE_SOURCE(...) #__VA_ARGS__

/// Old style: VEX_FUNCTION_V1(name, double(double, double), "return prm1 + prm2;");
#define VEX_FUNCTION_V1(fname, signature, body_str)                                                 \
    struct vex_function_##fname : vex::UserFunction<vex_function_##fname,                           \
            vex::detail::signature_params<signature>::result_type> {                               \
        vex_function_##fname() {}                                                                   \
        static std::string name() { return #fname; }                                                \
        static void params(std::vector<std::pair<std::string, std::string>> &p) {                   \
            vex::detail::signature_params<signature>::get(p); }                                     \
        static std::string body() { return body_str; }                                              \
    } const fname

// host parameter list of (t1, a)(t2, b)...: `t1 a, t2 b`
// (the separating comma is produced one scan later than everything else, so that it never splits the
// arguments of VEXCL_CAT)
#define VEXCL_EMPTY()
#define VEXCL_COMMA() ,
#define VEXCL_ARG_FIRST(t, n) t n VEXCL_ARG_A
#define VEXCL_ARG_A(t, n) VEXCL_COMMA VEXCL_EMPTY() () t n VEXCL_ARG_B
#define VEXCL_ARG_B(t, n) VEXCL_COMMA VEXCL_EMPTY() () t n VEXCL_ARG_A
#define VEXCL_ARG_A_END
#define VEXCL_ARG_B_END
#define VEXCL_ARG_SEQ(seq) VEXCL_CAT(VEXCL_ARG_FIRST seq, _END)

/// Inside a struct: the same body as the device function `device` and as the host operator()
/// (sort comparators, scan / reduce operators; function.hpp:228-247 of the reference):
///     struct less { VEX_DUAL_FUNCTOR(bool, (int, a)(int, b), return a < b;) };
#define VEX_DUAL_FUNCTOR(rettype, args, ...)                                                        \
    VEX_FUNCTION(rettype, device, args, __VA_ARGS__);                                               \
    rettype operator()(VEXCL_ARG_SEQ(args)) const { __VA_ARGS__ }

namespace vex {

// ---- builtin functions (function.hpp:255-448) -----------------------------------
namespace detail {
    template <class... A> struct any_expr : std::false_type {};
    template <class H, class... T> struct any_expr<H, T...> : std::integral_constant<bool, is_expr<H>::value || any_expr<T...>::value> {};
    template <class... A> struct all_operands : std::true_type {};
    template <class H, class... T> struct all_operands<H, T...> : std::integral_constant<bool, is_operand<H>::value && all_operands<T...>::value> {};

    template <class Tag> struct builtin_function {
        static void preamble(gen_context &) {}
        static std::string name() { return Tag::name(); }
    };
}

// defined in vex::detail (ADL for node operands) and re-exported to vex::
#define VEXCL_BUILTIN_FUNCTION(fname)                                                               \
    namespace detail {                                                                              \
    struct builtin_##fname { static const char *name() { return #fname; } };                        \
    template <class... Args>                                                                        \
    typename std::enable_if<any_expr<Args...>::value && all_operands<Args...>::value,               \
        const function_call<builtin_function<builtin_##fname>,                                      \
            typename std::common_type<typename as_expr_t<Args>::value_type...>::type,               \
            as_expr_t<Args>...>>::type                                                              \
    fname(const Args &...args) {                                                                    \
        return function_call<builtin_function<builtin_##fname>,                                     \
            typename std::common_type<typename as_expr_t<Args>::value_type...>::type,               \
            as_expr_t<Args>...>(as_expr<Args>::get(args)...);                                       \
    }                                                                                               \
    }                                                                                               \
    using detail::fname;

VEXCL_BUILTIN_FUNCTION(acos)   VEXCL_BUILTIN_FUNCTION(acosh)  VEXCL_BUILTIN_FUNCTION(asin)
VEXCL_BUILTIN_FUNCTION(asinh)  VEXCL_BUILTIN_FUNCTION(atan)   VEXCL_BUILTIN_FUNCTION(atan2)
VEXCL_BUILTIN_FUNCTION(atanh)  VEXCL_BUILTIN_FUNCTION(cbrt)   VEXCL_BUILTIN_FUNCTION(ceil)
VEXCL_BUILTIN_FUNCTION(copysign) VEXCL_BUILTIN_FUNCTION(cos)  VEXCL_BUILTIN_FUNCTION(cosh)
VEXCL_BUILTIN_FUNCTION(erf)    VEXCL_BUILTIN_FUNCTION(erfc)   VEXCL_BUILTIN_FUNCTION(exp)
VEXCL_BUILTIN_FUNCTION(exp2)   VEXCL_BUILTIN_FUNCTION(exp10)  VEXCL_BUILTIN_FUNCTION(expm1)
VEXCL_BUILTIN_FUNCTION(fabs)   VEXCL_BUILTIN_FUNCTION(fdim)   VEXCL_BUILTIN_FUNCTION(floor)
VEXCL_BUILTIN_FUNCTION(fma)    VEXCL_BUILTIN_FUNCTION(fmax)   VEXCL_BUILTIN_FUNCTION(fmin)
VEXCL_BUILTIN_FUNCTION(fmod)   VEXCL_BUILTIN_FUNCTION(hypot)  VEXCL_BUILTIN_FUNCTION(ldexp)
VEXCL_BUILTIN_FUNCTION(lgamma) VEXCL_BUILTIN_FUNCTION(log)    VEXCL_BUILTIN_FUNCTION(log2)
VEXCL_BUILTIN_FUNCTION(log10)  VEXCL_BUILTIN_FUNCTION(log1p)  VEXCL_BUILTIN_FUNCTION(max)
VEXCL_BUILTIN_FUNCTION(min)    VEXCL_BUILTIN_FUNCTION(pow)    VEXCL_BUILTIN_FUNCTION(remainder)
VEXCL_BUILTIN_FUNCTION(rint)   VEXCL_BUILTIN_FUNCTION(round)  VEXCL_BUILTIN_FUNCTION(rsqrt)
VEXCL_BUILTIN_FUNCTION(sin)    VEXCL_BUILTIN_FUNCTION(sinh)   VEXCL_BUILTIN_FUNCTION(sqrt)
VEXCL_BUILTIN_FUNCTION(tan)    VEXCL_BUILTIN_FUNCTION(tanh)   VEXCL_BUILTIN_FUNCTION(tgamma)
VEXCL_BUILTIN_FUNCTION(trunc)  VEXCL_BUILTIN_FUNCTION(isnan)  VEXCL_BUILTIN_FUNCTION(isinf)
#undef VEXCL_BUILTIN_FUNCTION

// ---- atomics (function.hpp:413-434): the first operand is a pointer-valued expression, `&view` or
//      `p + i`; the value of the call is the value the location held before.  Both spellings.
namespace detail {
template <class P, class... V> struct atomic_result { typedef typename std::remove_pointer<typename as_expr_t<P>::value_type>::type type; };
}
#define VEXCL_ATOMIC_FUNCTION(fname, device_name)                                                   \
    namespace detail {                                                                              \
    struct builtin_##fname { static const char *name() { return #device_name; } };                  \
    template <class P, class... V>                                                                  \
    typename std::enable_if<is_expr<P>::value && std::is_pointer<typename as_expr_t<P>::value_type>::value && \
        all_operands<V...>::value,                                                                  \
        const function_call<builtin_function<builtin_##fname>, typename atomic_result<P>::type,     \
            as_expr_t<P>, as_expr_t<V>...>>::type                                                   \
    fname(const P &p, const V &...v) {                                                              \
        return function_call<builtin_function<builtin_##fname>, typename atomic_result<P>::type,    \
            as_expr_t<P>, as_expr_t<V>...>(as_expr<P>::get(p), as_expr<V>::get(v)...);              \
    }                                                                                               \
    }                                                                                               \
    using detail::fname;
VEXCL_ATOMIC_FUNCTION(atomicAdd, atomicAdd)   VEXCL_ATOMIC_FUNCTION(atomic_add, atomicAdd)
VEXCL_ATOMIC_FUNCTION(atomicSub, atomicSub)   VEXCL_ATOMIC_FUNCTION(atomic_sub, atomicSub)
VEXCL_ATOMIC_FUNCTION(atomicExch, atomicExch) VEXCL_ATOMIC_FUNCTION(atomic_xchg, atomicExch)
VEXCL_ATOMIC_FUNCTION(atomicMin, atomicMin)   VEXCL_ATOMIC_FUNCTION(atomic_min, atomicMin)
VEXCL_ATOMIC_FUNCTION(atomicMax, atomicMax)   VEXCL_ATOMIC_FUNCTION(atomic_max, atomicMax)
VEXCL_ATOMIC_FUNCTION(atomicCAS, atomicCAS)   VEXCL_ATOMIC_FUNCTION(atomic_cmpxchg, atomicCAS)
VEXCL_ATOMIC_FUNCTION(atomicAnd, atomicAnd)   VEXCL_ATOMIC_FUNCTION(atomic_and, atomicAnd)
VEXCL_ATOMIC_FUNCTION(atomicOr, atomicOr)     VEXCL_ATOMIC_FUNCTION(atomic_or, atomicOr)
VEXCL_ATOMIC_FUNCTION(atomicXor, atomicXor)   VEXCL_ATOMIC_FUNCTION(atomic_xor, atomicXor)
VEXCL_ATOMIC_FUNCTION(atomicInc, atomicInc)   VEXCL_ATOMIC_FUNCTION(atomicDec, atomicDec)
#undef VEXCL_ATOMIC_FUNCTION

// abs(): fabs for floating point expressions, abs for integers (function.hpp:465-505)
namespace detail {
struct builtin_abs_int { static const char *name() { return "abs"; } };
template <class Arg>
typename std::enable_if<is_expr<Arg>::value && std::is_floating_point<typename as_expr_t<Arg>::value_type>::value,
    const function_call<builtin_function<builtin_fabs>, typename as_expr_t<Arg>::value_type, as_expr_t<Arg>>>::type
abs(const Arg &a) {
    return function_call<builtin_function<builtin_fabs>, typename as_expr_t<Arg>::value_type,
           as_expr_t<Arg>>(as_expr<Arg>::get(a));
}
template <class Arg>
typename std::enable_if<is_expr<Arg>::value && !std::is_floating_point<typename as_expr_t<Arg>::value_type>::value,
    const function_call<builtin_function<builtin_abs_int>, typename as_expr_t<Arg>::value_type, as_expr_t<Arg>>>::type
abs(const Arg &a) {
    return function_call<builtin_function<builtin_abs_int>, typename as_expr_t<Arg>::value_type,
           as_expr_t<Arg>>(as_expr<Arg>::get(a));
}
} // namespace detail
using detail::abs;

} // namespace vex
#endif
