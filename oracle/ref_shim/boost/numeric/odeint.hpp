// Stand-in for <boost/numeric/odeint.hpp>, as far as examples/symbolic.cpp of the reference uses it: the classical
// fourth-order Runge-Kutta stepper over a range state (std::array of values supporting + and scalar *).  The example
// instantiates it with vex::symbolic<double> to RECORD one step as kernel source.  Test infrastructure only.
#ifndef VEX_REF_SHIM_BOOST_ODEINT_HPP
#define VEX_REF_SHIM_BOOST_ODEINT_HPP
#include <cstddef>
namespace boost { namespace numeric { namespace odeint {
struct range_algebra {};
struct vector_space_algebra {};
struct default_operations {};
template <class State, class Value = double, class Deriv = State, class Time = Value,
          class Algebra = range_algebra, class Operations = default_operations>
class runge_kutta4 {
    public:
        template <class System>
        void do_step(System system, State &x, Time t, Time dt) {
            Deriv k1, k2, k3, k4;
            State xt;
            const std::size_t n = x.size();
            const Value h = dt, h2 = dt / 2, h6 = dt / 6;
            system(x, k1, t);
            for (std::size_t i = 0; i < n; ++i) xt[i] = x[i] + h2 * k1[i];
            system(xt, k2, t + h2);
            for (std::size_t i = 0; i < n; ++i) xt[i] = x[i] + h2 * k2[i];
            system(xt, k3, t + h2);
            for (std::size_t i = 0; i < n; ++i) xt[i] = x[i] + h * k3[i];
            system(xt, k4, t + h);
            for (std::size_t i = 0; i < n; ++i) x[i] += h6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
        }
};
} } }
#endif
