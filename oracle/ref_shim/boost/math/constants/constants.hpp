// shim: the few boost::math::constants the reference's tests compare against
#ifndef VEX_REF_SHIM_MATH_CONSTANTS_HPP
#define VEX_REF_SHIM_MATH_CONSTANTS_HPP
namespace boost { namespace math { namespace constants {
template <class T> constexpr T pi() { return static_cast<T>(3.141592653589793238462643383279502884L); }
template <class T> constexpr T two_pi() { return static_cast<T>(6.283185307179586476925286766559005768L); }
template <class T> constexpr T half_pi() { return static_cast<T>(1.570796326794896619231321691639751442L); }
template <class T> constexpr T e() { return static_cast<T>(2.718281828459045235360287471352662498L); }
template <class T> constexpr T root_two() { return static_cast<T>(1.414213562373095048801688724209698079L); }
} } }
#endif
