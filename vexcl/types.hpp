#ifndef VEXCL_TYPES_HPP
#define VEXCL_TYPES_HPP
// Scalar type vocabulary of the vex:: API (reference: vexcl/types.hpp:72-320).
// The OpenCL cl_* scalar names are kept so user code recompiles unchanged.  The short
// vector types cl_<T>{2,4,8,16} are plain aggregates `{ T s[N]; }` with the alignment of
// the device type; on the device lengths 2 and 4 are the HIP built-in vector types
// (int2, float4, ...), lengths 8 and 16 are clang extended vectors declared in the
// standard kernel header (backend.hpp) -- both support component-wise arithmetic.
#include <cstddef>
#include <cstdint>
#include <iostream>
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

// the device-side shorthands, usable in host signatures of user functions as well
typedef unsigned char  uchar;
typedef unsigned short ushort;
typedef unsigned int   uint;
typedef unsigned long  ulong;

namespace vex {
/// Host image of a device short vector (types.hpp:72-150 of the reference: cl_float2, ... cl_ulong16).
template <class T, unsigned N>
struct alignas(sizeof(T) * N) cl_vector_type {
    T s[N];
    T &operator[](size_t i) { return s[i]; }
    const T &operator[](size_t i) const { return s[i]; }
};
}

#define VEXCL_SHORT_VECTORS(T)                       \
    typedef vex::cl_vector_type<cl_##T, 2> cl_##T##2;   \
    typedef vex::cl_vector_type<cl_##T, 4> cl_##T##4;   \
    typedef vex::cl_vector_type<cl_##T, 8> cl_##T##8;   \
    typedef vex::cl_vector_type<cl_##T, 16> cl_##T##16
VEXCL_SHORT_VECTORS(char); VEXCL_SHORT_VECTORS(uchar); VEXCL_SHORT_VECTORS(short); VEXCL_SHORT_VECTORS(ushort);
VEXCL_SHORT_VECTORS(int); VEXCL_SHORT_VECTORS(uint); VEXCL_SHORT_VECTORS(long); VEXCL_SHORT_VECTORS(ulong);
VEXCL_SHORT_VECTORS(float); VEXCL_SHORT_VECTORS(double);
#undef VEXCL_SHORT_VECTORS

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

template <class T, unsigned N> struct type_name_impl<cl_vector_type<T, N>> {
    static std::string get() { return type_name_impl<T>::get() + std::to_string(N); }
};

template <class T> struct is_cl_scalar : std::is_arithmetic<T> {};
template <class T> struct is_cl_vector : std::false_type {};
template <class T, unsigned N> struct is_cl_vector<cl_vector_type<T, N>> : std::true_type {};
template <class T> struct is_cl_native : std::integral_constant<bool, is_cl_scalar<T>::value || is_cl_vector<T>::value> {};
template <class T> struct cl_scalar_of { typedef T type; };
template <class T, unsigned N> struct cl_scalar_of<cl_vector_type<T, N>> { typedef T type; };
template <class T> struct cl_vector_length : std::integral_constant<unsigned, 1> {};
template <class T, unsigned N> struct cl_vector_length<cl_vector_type<T, N>> : std::integral_constant<unsigned, N> {};
/// cl_vector_of<float, 4>::type is cl_float4; length 1 is the scalar itself (types.hpp:152-200).
template <class T, unsigned N> struct cl_vector_of { typedef cl_vector_type<T, N> type; };
template <class T> struct cl_vector_of<T, 1> { typedef T type; };

// component-wise host arithmetic, vector with vector and vector with scalar
#define VEXCL_SHORT_VECTOR_OP(op)                                                                             \
    template <class T, unsigned N>                                                                            \
    cl_vector_type<T, N> &operator op##=(cl_vector_type<T, N> &a, const cl_vector_type<T, N> &b) {           \
        for (unsigned i = 0; i < N; ++i) a.s[i] op##= b.s[i];                                                 \
        return a;                                                                                             \
    }                                                                                                         \
    template <class T, unsigned N>                                                                            \
    cl_vector_type<T, N> &operator op##=(cl_vector_type<T, N> &a, T b) {                                      \
        for (unsigned i = 0; i < N; ++i) a.s[i] op##= b;                                                      \
        return a;                                                                                             \
    }                                                                                                         \
    template <class T, unsigned N>                                                                            \
    cl_vector_type<T, N> operator op(cl_vector_type<T, N> a, const cl_vector_type<T, N> &b) { return a op##= b; } \
    template <class T, unsigned N>                                                                            \
    cl_vector_type<T, N> operator op(cl_vector_type<T, N> a, T b) { return a op##= b; }                       \
    template <class T, unsigned N>                                                                            \
    cl_vector_type<T, N> operator op(T a, const cl_vector_type<T, N> &b) {                                    \
        cl_vector_type<T, N> r;                                                                               \
        for (unsigned i = 0; i < N; ++i) r.s[i] = a op b.s[i];                                                \
        return r;                                                                                             \
    }
VEXCL_SHORT_VECTOR_OP(+) VEXCL_SHORT_VECTOR_OP(-) VEXCL_SHORT_VECTOR_OP(*) VEXCL_SHORT_VECTOR_OP(/)
#undef VEXCL_SHORT_VECTOR_OP

template <class T, unsigned N>
bool operator==(const cl_vector_type<T, N> &a, const cl_vector_type<T, N> &b) {
    for (unsigned i = 0; i < N; ++i) if (!(a.s[i] == b.s[i])) return false;
    return true;
}
template <class T, unsigned N>
bool operator!=(const cl_vector_type<T, N> &a, const cl_vector_type<T, N> &b) { return !(a == b); }

template <class T, unsigned N>
std::ostream &operator<<(std::ostream &os, const cl_vector_type<T, N> &v) {
    os << "(";
    for (unsigned i = 0; i < N; ++i) os << (i ? "," : "") << +v.s[i];
    return os << ")";
}

} // namespace vex
#endif
