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
