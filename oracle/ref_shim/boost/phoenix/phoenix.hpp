// Stand-in for <boost/phoenix/phoenix.hpp>, as far as tests/generator.cpp of the reference uses it:
// the placeholders arg1..arg3 and lazy + - * / over them.  Test infrastructure only.
#ifndef VEX_REF_SHIM_BOOST_PHOENIX_HPP
#define VEX_REF_SHIM_BOOST_PHOENIX_HPP
#include <tuple>
#include <utility>
namespace boost { namespace phoenix {
struct actor_base {};
template <int N> struct argument : actor_base {
    template <class... A> auto operator()(const A &...a) const -> decltype(std::get<N>(std::tie(a...))) { return std::get<N>(std::tie(a...)); }
};
#define VEX_SHIM_PHOENIX_OP(name, op)                                                                         \
    template <class L, class R> struct name : actor_base {                                                    \
        L l; R r; name(const L &l, const R &r) : l(l), r(r) {}                                                \
        template <class... A> auto operator()(const A &...a) const -> decltype(l(a...) op r(a...)) { return l(a...) op r(a...); } \
    };                                                                                                        \
    template <class L, class R>                                                                               \
    typename std::enable_if<std::is_base_of<actor_base, L>::value && std::is_base_of<actor_base, R>::value, name<L, R>>::type \
    operator op(const L &l, const R &r) { return name<L, R>(l, r); }
VEX_SHIM_PHOENIX_OP(lazy_plus, +)
VEX_SHIM_PHOENIX_OP(lazy_minus, -)
VEX_SHIM_PHOENIX_OP(lazy_mul, *)
VEX_SHIM_PHOENIX_OP(lazy_div, /)
#undef VEX_SHIM_PHOENIX_OP
namespace arg_names { static const argument<0> arg1; static const argument<1> arg2; static const argument<2> arg3; }
} }
#endif
