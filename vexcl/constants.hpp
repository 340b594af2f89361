#ifndef VEXCL_CONSTANTS_HPP
#define VEXCL_CONSTANTS_HPP
// Compile-time constants baked into the kernel text (reference:
// vexcl/constants.hpp): vex::constants::pi() etc. and std::integral_constant.
#include <iomanip>
#include <limits>
#include <sstream>
#include "operations.hpp"

namespace vex {
namespace detail {
/// A user-defined constant: Impl::host() is evaluated on the host when the kernel text is written and
/// printed with enough digits to round-trip (constants.hpp:93-162 of the reference).  The constant is
/// identified by its TYPE, so the kernel cache stays keyed on the expression type alone.
template <class Impl>
struct user_constant : expression_base {
    typedef typename Impl::value_type value_type;
    void preamble(gen_context &) const {} void params(gen_context &) const {} void local_init(gen_context &) const {}
    void emit(gen_context &c) const {
        std::ostringstream s;
        if (std::is_floating_point<value_type>::value) s << std::scientific << std::setprecision(16);
        s << "( " << Impl::host() << " )";
        c.src << s.str();
    }
    void set_args(arg_context &) const {} void get_props(prop_context &) const {}
};
}
namespace detail {
/// std::integral_constant<T, v> as an operand: the value is part of the kernel
/// text (constants.hpp:60-90 of the reference), `x = std::integral_constant<int, 42>()`.
template <class T, T v>
struct integral_text : expression_base {
    typedef T value_type;
    void preamble(gen_context &) const {} void params(gen_context &) const {} void local_init(gen_context &) const {}
    void emit(gen_context &c) const { c.src << "( " << v << " )"; }
    void set_args(arg_context &) const {} void get_props(prop_context &) const {}
};
template <class T, T v> struct is_extra_operand<std::integral_constant<T, v>> : std::true_type {};
template <class T, T v> struct as_expr<std::integral_constant<T, v>, void> {
    typedef integral_text<T, v> type;
    static type get(const std::integral_constant<T, v> &) { return type(); }
};
}

/// VEX_CONSTANT(name, value): `name()` is an expression terminal whose value is written into the kernel
/// text; `name` itself converts to the host value.  Usable at namespace and at function scope.
#define VEX_CONSTANT(name, value)                                                               \
    struct constant_##name {                                                                    \
        typedef decltype(value) value_type;                                                     \
        static value_type host() { return value; }                                              \
        vex::detail::user_constant<constant_##name> operator()() const {                        \
            return vex::detail::user_constant<constant_##name>();                               \
        }                                                                                       \
        operator value_type() const { return host(); }                                          \
    };                                                                                          \
    const constant_##name name = {}

/// Mathematical constants (the set boost::math::constants provides to the reference, constants.hpp:164-235).
namespace constants {
VEX_CONSTANT(pi, 3.141592653589793238462643383279502884);
VEX_CONSTANT(two_pi, 6.283185307179586476925286766559005768);
VEX_CONSTANT(half_pi, 1.570796326794896619231321691639751442);
VEX_CONSTANT(root_pi, 1.772453850905516027298167483341145183);
VEX_CONSTANT(root_half_pi, 1.253314137315500251207882642405522627);
VEX_CONSTANT(root_two_pi, 2.506628274631000502415765284811045253);
VEX_CONSTANT(root_ln_four, 1.177410022515474691011569326459699637);
VEX_CONSTANT(e, 2.718281828459045235360287471352662498);
VEX_CONSTANT(half, 0.5);
VEX_CONSTANT(third, 1.0 / 3.0);
VEX_CONSTANT(twothirds, 2.0 / 3.0);
VEX_CONSTANT(euler, 0.577215664901532860606512090082402431);
VEX_CONSTANT(root_two, 1.414213562373095048801688724209698079);
VEX_CONSTANT(root_three, 1.732050807568877293527446341505872367);
VEX_CONSTANT(half_root_two, 0.707106781186547524400844362104849039);
VEX_CONSTANT(one_div_root_two, 0.707106781186547524400844362104849039);
VEX_CONSTANT(ln_two, 0.693147180559945309417232121458176568);
VEX_CONSTANT(ln_ten, 2.302585092994045684017991454684364208);
VEX_CONSTANT(ln_ln_two, -0.366512920581664327012439158232669470);
VEX_CONSTANT(log10_e, 0.434294481903251827651128918916605082);
VEX_CONSTANT(one_div_log10_e, 2.302585092994045684017991454684364208);
VEX_CONSTANT(pi_minus_three, 0.141592653589793238462643383279502884);
VEX_CONSTANT(four_minus_pi, 0.858407346410206761537356616720497116);
VEX_CONSTANT(exp_minus_half, 0.606530659712633423603799534991180453);
VEX_CONSTANT(one_div_two_pi, 0.159154943091895335768883763372514362);
VEX_CONSTANT(one_div_root_pi, 0.564189583547756286948079451560772586);
VEX_CONSTANT(one_div_root_two_pi, 0.398942280401432677939946059934381868);
VEX_CONSTANT(one_div_euler, 1.732454714600633473583025315860829681);
VEX_CONSTANT(degree, 0.017453292519943295769236907684886127);
VEX_CONSTANT(radian, 57.29577951308232087679815481410517033);
VEX_CONSTANT(pi_sqr, 9.869604401089358618834490999876151135);
VEX_CONSTANT(pi_cubed, 31.00627668029982017547631506710139520);
VEX_CONSTANT(pi_sqr_div_six, 1.644934066848226436472415166646025189);
VEX_CONSTANT(four_thirds_pi, 4.188790204786390984616857844372670512);
VEX_CONSTANT(cbrt_pi, 1.464591887561523263020142527263790391);
VEX_CONSTANT(one_div_cbrt_pi, 0.682784063255295681467020833158164171);
VEX_CONSTANT(e_pow_pi, 23.14069263277926900572908636794854738);
VEX_CONSTANT(pi_pow_e, 22.45915771836104547342715220454373502);
VEX_CONSTANT(root_e, 1.648721270700128146848650787814163571);
VEX_CONSTANT(euler_sqr, 0.333177923807718674318376136355244226);
VEX_CONSTANT(phi, 1.618033988749894848204586834365638117);
VEX_CONSTANT(ln_phi, 0.481211825059603447497758913424368423);
VEX_CONSTANT(one_div_ln_phi, 2.078086921235027537601322606117795767);
VEX_CONSTANT(catalan, 0.915965594177219015054603514932384110);
VEX_CONSTANT(glaisher, 1.282427129100622636875342568869791727);
VEX_CONSTANT(khinchin, 2.685452001065306445309714835481795693);
VEX_CONSTANT(cos_one, 0.540302305868139717400936607442976603);
VEX_CONSTANT(sin_one, 0.841470984807896506652502321630298999);
VEX_CONSTANT(cosh_one, 1.543080634815243778477905620757061682);
VEX_CONSTANT(sinh_one, 1.175201193643801456882381850595600815);
}
} // namespace vex
#endif
