#ifndef VEXCL_TYPES_HPP
#define VEXCL_TYPES_HPP
// Scalar type vocabulary of the vex:: API (reference: vexcl/types.hpp:72-320).
// The OpenCL cl_* scalar names are kept so user code recompiles unchanged; the
// cl_<T>N vector types are out of scope.
#include <cstddef>
#include <cstdint>
#include <string>
#include <type_traits>

typedef int8_t   cl_char;
typedef uint8_t  cl_uchar;
typedef int16_t  cl_short;
typedef uint16_t cl_ushort;
typedef int32_t  cl_int;
typedef uint32_t cl_uint;
typedef long     cl_long;      // 64-bit on this platform, prints as "long" in kernels
typedef unsigned long cl_ulong;
typedef float    cl_float;
typedef double   cl_double;

namespace vex {

/// Device-side spelling of a host type (types.hpp:203-268).
template <class T, class Enable = void> struct type_name_impl;

#define VEXCL_TYPE_NAME(T, S) \
    template <> struct type_name_impl<T> { static std::string get() { return S; } }
VEXCL_TYPE_NAME(float, "float");
VEXCL_TYPE_NAME(double, "double");
VEXCL_TYPE_NAME(char, "char");
VEXCL_TYPE_NAME(signed char, "char");
VEXCL_TYPE_NAME(unsigned char, "uchar");
VEXCL_TYPE_NAME(short, "short");
VEXCL_TYPE_NAME(unsigned short, "ushort");
VEXCL_TYPE_NAME(int, "int");
VEXCL_TYPE_NAME(unsigned int, "uint");
VEXCL_TYPE_NAME(long, "long");
VEXCL_TYPE_NAME(unsigned long, "ulong");
VEXCL_TYPE_NAME(long long, "long");
VEXCL_TYPE_NAME(unsigned long long, "ulong");
VEXCL_TYPE_NAME(bool, "bool");
#undef VEXCL_TYPE_NAME

template <class T> struct type_name_impl<T*> {
    static std::string get() { return type_name_impl<typename std::decay<T>::type>::get() + " *"; }
};
template <class T> struct type_name_impl<const T*> {
    static std::string get() { return "const " + type_name_impl<typename std::decay<T>::type>::get() + " *"; }
};

template <class T> inline std::string type_name() {
    return type_name_impl<typename std::remove_cv<T>::type>::get();
}

template <class T> struct is_cl_scalar : std::is_arithmetic<T> {};
template <class T> struct is_cl_vector : std::false_type {};
template <class T> struct is_cl_native : std::is_arithmetic<T> {};
template <class T> struct cl_scalar_of { typedef T type; };
template <class T> struct cl_vector_length : std::integral_constant<unsigned, 1> {};

} // namespace vex
#endif
