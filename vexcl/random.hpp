#ifndef VEXCL_RANDOM_HPP
#define VEXCL_RANDOM_HPP
// vex::Random<T, Generator> / vex::RandomNormal<T, Generator>: counter-based random numbers as
// expression functions -- `x = rnd(vex::element_index(), seed)` -- so a Monte-Carlo estimate is
// ONE fused kernel (reference: vexcl/random.hpp:60-275; generators vexcl/random/philox.hpp,
// vexcl/random/threefry.hpp, both from the Random123 suite of Salmon et al., SC'11;
// tests/random.cpp).  T is a scalar or a short vector (cl_float4, cl_double4, ...) of up to 32 bytes.
//
// The streams are the reference's.  Outputs of less than 32 bytes use 32-bit words: counter =
// (uint)idx, (uint)seed -- repeated when four words are needed (outputs above 8 bytes) -- and every
// key word 0x12345678; a 32-byte output uses four 64-bit words the same way.  The output words are
// read as the result's bytes; floats are then the 32-bit word divided by 2^32 - 1, doubles the
// 64-bit word divided by 2^64 - 1.  RandomNormal is Box-Muller on two uniforms (doubles: the 4x32
// generators).  The device functions are written as loops over the rounds (the reference unrolls
// them into the source text); the test suite holds the same generators in numpy, pinned by the
// published Random123 known-answer vectors.
#include "operations.hpp"

namespace vex {
namespace random {

/// Philox-NxW-10 (Salmon et al.), W = 32 or 64 bits: rounds of one multiplication per word pair.
struct philox {
    static std::string name() { return "philox"; }
    static std::string function_name(size_t n, size_t bits = 32) {
        return std::string("philox_") + (bits == 32 ? "uint_" : "ulong_") + std::to_string(n) + "_10";
    }
    static size_t key_words(size_t n) { return n / 2; }
    static void define(detail::gen_context &c, size_t n, size_t bits = 32) {
        const std::string key = "gen:" + function_name(n, bits);
        if (c.seen.count(key)) return;
        c.seen.insert(key);
        const bool w32 = bits == 32;
        const std::string W = w32 ? "uint" : "ulong", hi = w32 ? "__umulhi" : "__umul64hi";
        const std::string weyl0 = w32 ? "0x9E3779B9u" : "0x9E3779B97F4A7C15ul", weyl1 = w32 ? "0xBB67AE85u" : "0xBB67AE8584CAA73Bul";
        const std::string m2 = w32 ? "0xD256D193u" : "0xD2B74407B1CE6E93ul";
        const std::string m40 = w32 ? "0xD2511F53u" : "0xD2E7470EE14C6C93ul", m41 = w32 ? "0xCD9E8D57u" : "0xCA5A826395121157ul";
        c.src.begin_function("void", function_name(n, bits));
        c.src.begin_function_parameters();
        c.src.parameter(W + " *", "ctr"); c.src.parameter(W + " *", "key");
        c.src.end_function_parameters();
        c.src.new_line() << "for(int round = 0; round < 10; ++round)";
        c.src.open("{");
        if (n == 2) {
            c.src.new_line() << "if (round) key[0] += " << weyl0 << ";";
            c.src.new_line() << "const " << W << " hi = " << hi << "(" << m2 << ", ctr[0]), lo = " << m2 << " * ctr[0];";
            c.src.new_line() << "ctr[0] = hi ^ key[0] ^ ctr[1];";
            c.src.new_line() << "ctr[1] = lo;";
        } else {
            c.src.new_line() << "if (round) { key[0] += " << weyl0 << "; key[1] += " << weyl1 << "; }";
            c.src.new_line() << "const " << W << " hi0 = " << hi << "(" << m40 << ", ctr[0]), lo0 = " << m40 << " * ctr[0];";
            c.src.new_line() << "const " << W << " hi1 = " << hi << "(" << m41 << ", ctr[2]), lo1 = " << m41 << " * ctr[2];";
            c.src.new_line() << "ctr[0] = hi1 ^ ctr[1] ^ key[0];";
            c.src.new_line() << "ctr[1] = lo1;";
            c.src.new_line() << "ctr[2] = hi0 ^ ctr[3] ^ key[1];";
            c.src.new_line() << "ctr[3] = lo0;";
        }
        c.src.close("}");
        c.src.end_function();
    }
};

/// Threefry-NxW-20 (Salmon et al.): add / rotate / xor rounds with a key injection every four.
/// N = 4 follows the reference's variant, which mixes the word pairs (0,1) and (2,3) without the
/// word permutation of Threefish (threefry.hpp:178-215).
struct threefry {
    static std::string name() { return "threefry"; }
    static std::string function_name(size_t n, size_t bits = 32) {
        return std::string("threefry_") + (bits == 32 ? "uint_" : "ulong_") + std::to_string(n) + "_20";
    }
    static size_t key_words(size_t n) { return n; }
    static void define(detail::gen_context &c, size_t n, size_t bits = 32) {
        const std::string key = "gen:" + function_name(n, bits);
        if (c.seen.count(key)) return;
        c.seen.insert(key);
        const bool w32 = bits == 32;
        const std::string W = w32 ? "uint" : "ulong", parity = w32 ? "0x1BD11BDAu" : "0x1BD11BDAA9FC1A22ul";
        c.src.begin_function("void", function_name(n, bits));
        c.src.begin_function_parameters();
        c.src.parameter(W + " *", "ctr"); c.src.parameter(W + " *", "key");
        c.src.end_function_parameters();
        if (n == 2) {
            c.src.new_line() << "const uint rot[8] = {" << (w32 ? "13, 15, 26, 6, 17, 29, 16, 24" : "16, 42, 12, 31, 16, 32, 24, 21") << "};";
            c.src.new_line() << W << " ks[3] = {key[0], key[1], " << parity << " ^ key[0] ^ key[1]};";
        } else {
            c.src.new_line() << "const uint rot[16] = {" << (w32 ? "10, 26, 11, 21, 13, 27, 23, 5, 6, 20, 17, 11, 25, 10, 18, 20"
                                                                 : "14, 16, 52, 57, 23, 40, 5, 37, 25, 33, 46, 12, 58, 22, 32, 32") << "};";
            c.src.new_line() << W << " ks[5] = {key[0], key[1], key[2], key[3], " << parity << " ^ key[0] ^ key[1] ^ key[2] ^ key[3]};";
        }
        c.src.new_line() << "for(int i = 0; i < " << n << "; ++i) ctr[i] += key[i];";
        c.src.new_line() << "for(int round = 0; round < 20; ++round)";
        c.src.open("{");
        if (n == 2) {
            c.src.new_line() << "const uint r = rot[round % 8];";
            c.src.new_line() << "ctr[0] += ctr[1]; ctr[1] = (ctr[1] << r) | (ctr[1] >> (" << bits << " - r)); ctr[1] ^= ctr[0];";
        } else {
            c.src.new_line() << "const uint r0 = rot[2 * (round % 8) + (round % 2)], r1 = rot[2 * (round % 8) + ((round + 1) % 2)];";
            c.src.new_line() << "ctr[0] += ctr[1]; ctr[1] = (ctr[1] << r0) | (ctr[1] >> (" << bits << " - r0)); ctr[1] ^= ctr[0];";
            c.src.new_line() << "ctr[2] += ctr[3]; ctr[3] = (ctr[3] << r1) | (ctr[3] >> (" << bits << " - r1)); ctr[3] ^= ctr[2];";
        }
        c.src.new_line() << "if ((round + 1) % 4 == 0)";
        c.src.open("{");
        c.src.new_line() << "const int j = round / 4 + 1;";
        c.src.new_line() << "for(int i = 0; i < " << n << "; ++i) ctr[i] += ks[(j + i) % " << n + 1 << "];";
        c.src.new_line() << "ctr[" << n - 1 << "] += j;";
        c.src.close("}");
        c.src.close("}");
        c.src.end_function();
    }
};

} // namespace random

namespace detail {
template <class T> struct random_traits {
    typedef typename cl_scalar_of<T>::type scalar;
    static_assert(std::is_arithmetic<scalar>::value && sizeof(T) <= 32 && (sizeof(T) & (sizeof(T) - 1)) == 0,
            "vex::Random supports scalars and short vectors of 1, 2, 4, 8, 16 or 32 bytes");
};

/// Shared machinery: a device function `T name(ulong prm1, ulong prm2)` emitted once per kernel.
template <class Impl, class T>
struct random_function {
    typedef T value_type;
    static std::string name() { return Impl::name(); }
    static void preamble(gen_context &c) {
        const std::string key = "fun:" + Impl::name();
        if (c.seen.count(key)) return;
        c.seen.insert(key);
        Impl::define(c);
    }
    template <class A1, class A2>
    typename std::enable_if<is_operand<A1>::value && is_operand<A2>::value,
        const function_call<Impl, T, as_expr_t<A1>, as_expr_t<A2>>>::type
    operator()(const A1 &idx, const A2 &seed) const {
        return function_call<Impl, T, as_expr_t<A1>, as_expr_t<A2>>(as_expr<A1>::get(idx), as_expr<A2>::get(seed));
    }
};
} // namespace detail

/// Uniform random numbers: [0, 1] for floating point types, all bit patterns for integers.
template <class T, class Generator = random::philox>
struct Random : detail::random_function<Random<T, Generator>, T>, detail::random_traits<T> {
    static std::string name() { return "random_" + type_name<T>() + "_" + Generator::name(); }
    static void define(detail::gen_context &c) {
        typedef typename cl_scalar_of<T>::type S;
        const size_t N = cl_vector_length<T>::value;
        const size_t bits = sizeof(T) < 32 ? 32 : 64, words = sizeof(T) <= 8 ? 2 : 4;
        const std::string W = bits == 32 ? "uint" : "ulong";
        Generator::define(c, words, bits);
        c.src.begin_function(type_name<T>(), name());
        c.src.begin_function_parameters();
        c.src.parameter("ulong", "prm1"); c.src.parameter("ulong", "prm2");
        c.src.end_function_parameters();
        // the output words are the bytes of the result (little endian), converted in place
        c.src.new_line() << "union { " << W << " ctr[" << words << "];";
        if (std::is_floating_point<S>::value)
            c.src << " " << (sizeof(S) == 4 ? "uint" : "ulong") << " bits[" << N << "]; " << type_name<S>() << " real[" << N << "];";
        c.src << " " << type_name<T>() << " res; } u;";
        for (size_t i = 0; i < words; i += 2)
            c.src.new_line() << "u.ctr[" << i << "] = (" << W << ")prm1; u.ctr[" << i + 1 << "] = (" << W << ")prm2;";
        c.src.new_line() << W << " key[" << Generator::key_words(words) << "];";
        c.src.new_line() << "for(int i = 0; i < " << Generator::key_words(words) << "; ++i) key[i] = 0x12345678;";
        c.src.new_line() << Generator::function_name(words, bits) << "(u.ctr, key);";
        if (std::is_same<S, float>::value)
            c.src.new_line() << "for(int i = 0; i < " << N << "; ++i) u.real[i] = u.bits[i] / 4294967295.0f;";
        else if (std::is_same<S, double>::value)
            c.src.new_line() << "for(int i = 0; i < " << N << "; ++i) u.real[i] = u.bits[i] / 18446744073709551615.0;";
        c.src.new_line() << "return u.res;";
        c.src.end_function();
    }
};

/// Normally distributed random numbers (Box-Muller on two uniforms of one generator call).
template <class T, class Generator = random::philox>
struct RandomNormal : detail::random_function<RandomNormal<T, Generator>, T> {
    static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value, "RandomNormal needs float or double");
    static std::string name() { return "random_normal_" + type_name<T>() + "_" + Generator::name(); }
    static void define(detail::gen_context &c) {
        const bool is_float = std::is_same<T, float>::value;
        const size_t n = is_float ? 2 : 4;
        Generator::define(c, n);
        c.src.begin_function(type_name<T>(), name());
        c.src.begin_function_parameters();
        c.src.parameter("ulong", "prm1"); c.src.parameter("ulong", "prm2");
        c.src.end_function_parameters();
        if (is_float) c.src.new_line() << "uint ctr[2] = {(uint)prm1, (uint)prm2};";
        else c.src.new_line() << "uint ctr[4] = {(uint)prm1, (uint)prm2, (uint)prm1, (uint)prm2};";
        c.src.new_line() << "uint key[4] = {0x12345678u, 0x12345678u, 0x12345678u, 0x12345678u};";
        c.src.new_line() << Generator::function_name(n) << "(ctr, key);";
        if (is_float) {
            c.src.new_line() << "const float u0 = ctr[0] / 4294967295.0f, u1 = ctr[1] / 4294967295.0f;";
            c.src.new_line() << "return sqrtf(-2 * logf(u0)) * cospif(2 * u1);";
        } else {
            c.src.new_line() << "const double u0 = (((ulong)ctr[1] << 32) | ctr[0]) / 18446744073709551615.0;";
            c.src.new_line() << "const double u1 = (((ulong)ctr[3] << 32) | ctr[2]) / 18446744073709551615.0;";
            c.src.new_line() << "return sqrt(-2 * log(u0)) * cospi(2 * u1);";
        }
        c.src.end_function();
    }
};

} // namespace vex
#endif
