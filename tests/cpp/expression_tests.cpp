// GPU: expression-engine terminals around the hot path -- vex::cast, vex::make_temp,
// vex::any_of / all_of, vector::reinterpret, vex::gather / scatter (reference tests:
// cast.cpp, temporary.cpp, logical.cpp, reinterpret.cpp, vector_copy.cpp:72-109).
#include "vex_test.hpp"
#include <array>
#include <numeric>

VEX_FUNCTION(double, t_sqr, (double, x), return x * x;);

TEST_CASE(casted_expression) {                                       // cast.cpp:9-16
    const size_t N = 1024;
    vex::vector<double> x(ctx, N);
    x = vex::cast<double>(5);
    check_sample(x, [](size_t, double a) { CHECK_EQUAL(a, 5.0); });
    vex::vector<int> k(ctx, N);
    k = vex::cast<int>(vex::element_index() * 0.75);                 // truncation happens on the device
    check_sample(k, [](size_t i, int a) { CHECK_EQUAL(a, int(i * 0.75)); });
    x = vex::cast<float>(1.0 / 3.0) * 3;                             // the cast fixes the deduced type
    check_sample(x, [](size_t, double a) { CHECK_EQUAL(a, double(float(1.0 / 3.0) * 3)); });
    static_assert(std::is_same<decltype(vex::cast<float>(x))::value_type, float>::value, "cast changes the value type");
}

TEST_CASE(temporary) {                                                // temporary.cpp:9-40
    const size_t n = 1024;
    std::vector<double> h = random_vector<double>(n);
    vex::vector<double> x(ctx, h), y(ctx, n);
    {
        auto s = vex::make_temp<1>(t_sqr(x) + 25);                    // deduced type
        y = s * (x + s);
        check_sample(y, [&](size_t i, double v) { double S = h[i] * h[i] + 25; CHECK_CLOSE(v, S * (h[i] + S), 1e-8); });
    }
    {
        auto s = vex::make_temp<1, double>(t_sqr(x) + 25);            // given type
        y = s * (x + s);
        check_sample(y, [&](size_t i, double v) { double S = h[i] * h[i] + 25; CHECK_CLOSE(v, S * (h[i] + S), 1e-8); });
    }
}

TEST_CASE(nested_and_reduced_temporaries) {                           // temporary.cpp:42-72
    const size_t n = 1024;
    std::vector<double> h = random_vector<double>(n);
    for (auto &v : h) v += 0.5;
    vex::vector<double> x(ctx, h), y(ctx, n);
    auto t1 = vex::make_temp<1>(log(x));
    auto t2 = vex::make_temp<2>(t1 + sin(x));
    y = t1 * t2;
    check_sample(y, [&](size_t i, double v) { double T1 = log(h[i]), T2 = T1 + sin(h[i]); CHECK_CLOSE(v, T1 * T2, 1e-8); });

    auto s2 = vex::make_temp<3>(pow(sin(x), 2));
    auto c2 = vex::make_temp<4>(pow(cos(x), 2));
    vex::Reductor<double, vex::SUM> sum(ctx);
    CHECK_CLOSE(sum(10 * (s2 + c2)), 10.0 * n, 1e-6);
}

TEST_CASE(temporaries_in_multiexpressions) {                          // temporary.cpp:74-108
    typedef std::array<double, 2> elem_t;
    const size_t n = 1024;
    std::vector<double> h = random_vector<double>(n);
    vex::vector<double> x(ctx, h);
    vex::multivector<double, 2> y(ctx, n);
    auto tmp = vex::make_temp<1>(sin(x));
    y = std::tie(tmp, sqrt(1 - tmp * tmp));
    check_sample(y, [&](size_t i, elem_t v) { CHECK_CLOSE(v[0], sin(h[i]), 1e-8); CHECK_CLOSE(v[1], cos(h[i]), 1e-8); });

    std::vector<double> h2 = random_vector<double>(2 * n);
    vex::multivector<double, 2> X(ctx, h2);
    auto tmp2 = vex::make_temp<1, double>(tan(X));                    // one temporary per component
    y = tmp2 * tmp2;
    check_sample(y, [&](size_t i, elem_t v) {
        CHECK_CLOSE(v[0], pow(tan(h2[i]), 2.0), 1e-8);
        CHECK_CLOSE(v[1], pow(tan(h2[n + i]), 2.0), 1e-8);
    });
}

TEST_CASE(logical_any_all) {                                          // logical.cpp:8-27
    const size_t N = 1024;
    vex::vector<int> x(ctx, N);
    x = vex::element_index();
    vex::any_of any_of(ctx);
    vex::all_of all_of(ctx);
    CHECK(any_of(x));
    CHECK(!any_of(0 * x));
    CHECK(any_of(x > N / 2));
    CHECK(!any_of(x < 0));
    CHECK(!all_of(x));
    CHECK(all_of((x + 1) > 0));
    CHECK(!all_of(x > N / 2));
}

TEST_CASE(reinterpret_vector) {                                       // reinterpret.cpp:8-20 (scalar types)
    const size_t n = 1024;
    vex::vector<cl_long> x(ctx, n);
    x.reinterpret<int>() = vex::element_index();                      // 2n ints in the same memory
    check_sample(x, [&](size_t i, cl_long v) {
        CHECK_EQUAL(int(v & 0xffffffff), int(2 * i));
        CHECK_EQUAL(int(v >> 32), int(2 * i + 1));
    });
    vex::vector<double> d(ctx, n);
    d = 1.0;
    auto bits = d.reinterpret<cl_ulong>();
    CHECK_EQUAL(bits.size(), n);
    CHECK_EQUAL(cl_ulong(bits[5]), cl_ulong(0x3ff0000000000000ull));
    CHECK(bits(0).raw() == reinterpret_cast<cl_ulong *>(d(0).raw()));
}

TEST_CASE(gather_scatter) {                                           // vector_copy.cpp:72-109
    const size_t n = 1 << 20, m = 100;
    std::vector<double> x = random_vector<double>(n);
    vex::vector<double> X(ctx, x);
    std::vector<size_t> i(m);
    std::uniform_int_distribution<size_t> pick(0, n - 1);
    for (auto &v : i) v = pick(test_rng());
    for (int sorted = 0; sorted < 2; ++sorted) {
        if (sorted) { std::sort(i.begin(), i.end()); i.resize(std::unique(i.begin(), i.end()) - i.begin()); }
        std::vector<double> data(i.size());
        vex::gather get(ctx, x.size(), i);
        vex::scatter put(ctx, x.size(), i);
        get(X, data);
        for (size_t p = 0; p < i.size(); ++p) CHECK(data[p] == x[i[p]]);
        vex::vector<double> Y(ctx, n);
        Y = 0;
        put(data, Y);
        for (size_t p = 0; p < i.size(); ++p) CHECK(double(Y[i[p]]) == x[i[p]]);
        vex::Reductor<double, vex::SUM> sum(ctx);
        std::vector<size_t> u = i; std::sort(u.begin(), u.end()); u.erase(std::unique(u.begin(), u.end()), u.end());
        double want = 0; for (size_t g : u) want += x[g];
        CHECK_CLOSE(sum(Y), want, 1e-10);                             // nothing else was touched
    }
}

// ---- counter-based random numbers (reference: tests/random.cpp; generators pinned by the published
//      known-answer vectors of Philox / Threefry, Salmon et al. SC'11) --------------------------------
namespace {
void host_philox2x32(uint32_t ctr[2], uint32_t key) {
    for (int r = 0; r < 10; ++r) {
        if (r) key += 0x9E3779B9u;
        uint64_t p = uint64_t(0xD256D193u) * ctr[0];
        uint32_t hi = uint32_t(p >> 32), lo = uint32_t(p);
        ctr[0] = hi ^ key ^ ctr[1]; ctr[1] = lo;
    }
}
void host_threefry2x32(uint32_t ctr[2], const uint32_t key[2]) {
    static const unsigned rot[8] = {13, 15, 26, 6, 17, 29, 16, 24};
    const uint32_t ks[3] = {key[0], key[1], 0x1BD11BDAu ^ key[0] ^ key[1]};
    ctr[0] += ks[0]; ctr[1] += ks[1];
    for (int r = 0; r < 20; ++r) {
        ctr[0] += ctr[1]; ctr[1] = (ctr[1] << rot[r % 8]) | (ctr[1] >> (32 - rot[r % 8])); ctr[1] ^= ctr[0];
        if ((r + 1) % 4 == 0) { int j = r / 4 + 1; ctr[0] += ks[j % 3]; ctr[1] += ks[(j + 1) % 3] + j; }
    }
}
}

TEST_CASE(random_generators_match_the_published_vectors) {
    {   // the host models first: Random123 known-answer vectors
        uint32_t c[2] = {0, 0}; host_philox2x32(c, 0);
        CHECK(c[0] == 0xff1dae59u && c[1] == 0x6cd10df2u);
        uint32_t t[2] = {0, 0}, k0[2] = {0, 0}; host_threefry2x32(t, k0);
        CHECK(t[0] == 0x6b200159u && t[1] == 0x99ba4efeu);
        uint32_t p[2] = {0x243f6a88u, 0x85a308d3u}, kp[2] = {0x13198a2eu, 0x03707344u}; host_threefry2x32(p, kp);
        CHECK(p[0] == 0xc4923a9cu && p[1] == 0x483df7a0u);
    }
    // the device streams: counter = ((uint)index, (uint)seed), key words 0x12345678 (vexcl/random.hpp:100-116)
    const size_t n = 4096;
    const cl_ulong seed = 0x1234567890ull;                             // only the low 32 bits enter the counter
    vex::vector<cl_uint> U(ctx, n);
    vex::vector<cl_ulong> L(ctx, n);
    vex::vector<double> D(ctx, n);
    vex::vector<float> F(ctx, n);
    vex::Random<cl_uint> ru; vex::Random<cl_ulong> rl; vex::Random<double> rd; vex::Random<float> rf;
    vex::Random<cl_ulong, vex::random::threefry> rt;
    U = ru(vex::element_index(), seed);
    L = rl(vex::element_index(), seed);
    D = rd(vex::element_index(), seed);
    F = rf(vex::element_index(), seed);
    std::vector<cl_uint> u(n); std::vector<cl_ulong> l(n), t(n); std::vector<double> d(n); std::vector<float> f(n);
    vex::copy(U, u); vex::copy(L, l); vex::copy(D, d); vex::copy(F, f);
    L = rt(vex::element_index(), seed);
    vex::copy(L, t);
    for (size_t i = 0; i < n; ++i) {
        uint32_t c[2] = {uint32_t(i), uint32_t(seed)};
        host_philox2x32(c, 0x12345678u);
        const uint64_t w = (uint64_t(c[1]) << 32) | c[0];
        CHECK(u[i] == c[0]);
        CHECK(l[i] == w);
        CHECK(d[i] == double(w) / 18446744073709551615.0);
        CHECK(f[i] == float(c[0]) / 4294967295.0f);
        uint32_t c3[2] = {uint32_t(i), uint32_t(seed)}, k3[2] = {0x12345678u, 0x12345678u};
        host_threefry2x32(c3, k3);
        CHECK(t[i] == ((uint64_t(c3[1]) << 32) | c3[0]));
    }
}

namespace {
template <class W> struct philox4_constants;
template <> struct philox4_constants<uint32_t> { static constexpr uint32_t m0 = 0xD2511F53u, m1 = 0xCD9E8D57u, w0 = 0x9E3779B9u, w1 = 0xBB67AE85u; };
template <> struct philox4_constants<uint64_t> {
    static constexpr uint64_t m0 = 0xD2E7470EE14C6C93ull, m1 = 0xCA5A826395121157ull, w0 = 0x9E3779B97F4A7C15ull, w1 = 0xBB67AE8584CAA73Bull;
};
inline void mulhilo(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) { uint64_t p = uint64_t(a) * b; hi = uint32_t(p >> 32); lo = uint32_t(p); }
inline void mulhilo(uint64_t a, uint64_t b, uint64_t &hi, uint64_t &lo) { unsigned __int128 p = (unsigned __int128)a * b; hi = uint64_t(p >> 64); lo = uint64_t(p); }
template <class W> void host_philox4(W ctr[4], W k0, W k1) {
    typedef philox4_constants<W> C;
    for (int r = 0; r < 10; ++r) {
        if (r) { k0 += C::w0; k1 += C::w1; }
        W hi0, lo0, hi1, lo1;
        mulhilo(C::m0, ctr[0], hi0, lo0); mulhilo(C::m1, ctr[2], hi1, lo1);
        const W c0 = hi1 ^ ctr[1] ^ k0, c2 = hi0 ^ ctr[3] ^ k1;
        ctr[0] = c0; ctr[1] = lo1; ctr[2] = c2; ctr[3] = lo0;
    }
}
}

TEST_CASE(random_short_vectors_follow_the_reference_streams) {       // random.hpp:85-152; tests/random.cpp:24-30
    {   // host models against the published vectors (Random123 kat_vectors: philox4x32, philox4x64)
        uint32_t a[4] = {0x243f6a88u, 0x85a308d3u, 0x13198a2eu, 0x03707344u}; host_philox4<uint32_t>(a, 0xa4093822u, 0x299f31d0u);
        CHECK(a[0] == 0xd16cfe09u && a[1] == 0x94fdccebu && a[2] == 0x5001e420u && a[3] == 0x24126ea1u);
        uint64_t b[4] = {0x243f6a8885a308d3ull, 0x13198a2e03707344ull, 0xa4093822299f31d0ull, 0x082efa98ec4e6c89ull};
        host_philox4<uint64_t>(b, 0x452821e638d01377ull, 0xbe5466cf34e90c6cull);
        CHECK(b[0] == 0xa528f45403e61d95ull && b[1] == 0x38c72dbd566e9788ull && b[2] == 0xa5a1610e72fd18b5ull && b[3] == 0x57bd43b5e52b7fe6ull);
    }
    const size_t n = 2048;
    const cl_ulong seed = 0xabcdef0123ull;                             // 64-bit counters take all of it, 32-bit ones the low half
    vex::vector<cl_float4> F(ctx, n); vex::vector<cl_double4> D(ctx, n); vex::vector<cl_int2> I(ctx, n); vex::vector<cl_double2> D2(ctx, n);
    vex::Random<cl_float4> rf; vex::Random<cl_double4> rd; vex::Random<cl_int2> ri; vex::Random<cl_double2> rd2;
    F = rf(vex::element_index(), seed);
    D = rd(vex::element_index(), seed);
    I = ri(vex::element_index(), seed);
    D2 = rd2(vex::element_index(), seed);
    std::vector<cl_float4> f(n); std::vector<cl_double4> d(n); std::vector<cl_int2> iv(n); std::vector<cl_double2> d2(n);
    vex::copy(F, f); vex::copy(D, d); vex::copy(I, iv); vex::copy(D2, d2);
    for (size_t i = 0; i < n; ++i) {
        uint32_t c[4] = {uint32_t(i), uint32_t(seed), uint32_t(i), uint32_t(seed)};
        host_philox4<uint32_t>(c, 0x12345678u, 0x12345678u);
        for (int k = 0; k < 4; ++k) CHECK(f[i].s[k] == float(c[k]) / 4294967295.0f);
        for (int k = 0; k < 2; ++k) CHECK(d2[i].s[k] == double((uint64_t(c[2 * k + 1]) << 32) | c[2 * k]) / 18446744073709551615.0);
        uint64_t w[4] = {uint64_t(i), seed, uint64_t(i), seed};
        host_philox4<uint64_t>(w, 0x12345678ull, 0x12345678ull);
        for (int k = 0; k < 4; ++k) CHECK(d[i].s[k] == double(w[k]) / 18446744073709551615.0);
        uint32_t c2[2] = {uint32_t(i), uint32_t(seed)};
        host_philox2x32(c2, 0x12345678u);
        CHECK(uint32_t(iv[i].s[0]) == c2[0] && uint32_t(iv[i].s[1]) == c2[1]);
    }
}

TEST_CASE(random_numbers_statistics_and_monte_carlo) {               // random.cpp:13-75
    const size_t N = 1 << 20;
    vex::Reductor<size_t, vex::SUM> sumi(ctx);
    vex::Reductor<double, vex::SUM> sumd(ctx);
    vex::Random<cl_double> rand3;
    vex::vector<cl_double> x3(ctx, N);
    x3 = rand3(vex::element_index(), 17);
    CHECK(sumi(x3 > 1) == 0);
    CHECK(sumi(x3 < 0) == 0);
    CHECK(std::abs(sumd(x3) / N - 0.5) < 1e-2);
    vex::RandomNormal<cl_double> rand4;
    vex::vector<cl_double> x4(ctx, N);
    x4 = rand4(vex::element_index(), 23);
    CHECK(std::abs(sumd(x4) / N) < 1e-2);                             // E X = 0
    CHECK(std::abs(sumd(fabs(x4)) / N - std::sqrt(2 / M_PI)) < 1e-2); // E |X| = sqrt(2 / pi)
    CHECK(std::abs(sumd(x4 * x4) / N - 1) < 1e-2);                    // E X^2 = 1
    vex::RandomNormal<float, vex::random::threefry> rand6;
    vex::vector<float> x6(ctx, N);
    x6 = rand6(vex::element_index(), 29);
    CHECK(std::abs(sumd(x6) / N) < 1e-2);
    vex::Random<cl_double, vex::random::threefry> rand5;
    vex::vector<cl_double> x5(ctx, N);
    x5 = rand5(vex::element_index(), 31);
    CHECK(std::abs(sumd(x5) / N - 0.5) < 1e-2);

    // pi by Monte Carlo in ONE fused kernel: two temporaries, a comparison, a reduction
    vex::Random<double, vex::random::threefry> rnd;
    vex::Reductor<size_t, vex::SUM> sum(ctx);
    auto i = vex::tag<0>(vex::element_index(0, N));
    auto x = vex::make_temp<1>(rnd(i, 1001));
    auto y = vex::make_temp<2>(rnd(i, 2002));
    double pi = 4.0 * sum((x * x + y * y) < 1) / N;
    CHECK_CLOSE(pi, M_PI, 0.5);
}
