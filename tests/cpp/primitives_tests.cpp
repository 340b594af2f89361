// GPU: vex::inclusive_scan / exclusive_scan / sort / sort_by_key, the checks of
// the reference's tests/scan.cpp:9-41 and tests/sort.cpp:9-45, on the
// 2-"device" context (cross-device carry and merge are exercised).
#include "vex_test.hpp"
#include <numeric>

TEST_CASE(scan_inclusive_ints_in_place) {                            // scan.cpp:9-24
    const size_t n = 1 << 20;
    std::vector<int> x = random_vector<int>(n);
    vex::vector<int> X(ctx, x);
    vex::inclusive_scan(X, X);
    std::partial_sum(x.begin(), x.end(), x.begin());
    std::vector<int> got(n); vex::copy(X, got);
    CHECK(got == x);
}

TEST_CASE(scan_exclusive_doubles_in_place) {                         // scan.cpp:26-41
    const size_t n = 1 << 20;
    std::vector<double> x = random_vector<double>(n);
    vex::vector<double> X(ctx, x);
    vex::exclusive_scan(X, X);
    std::vector<double> want(n); want[0] = 0;
    for (size_t i = 1; i < n; ++i) want[i] = want[i - 1] + x[i - 1];
    check_sample(X, [&](size_t i, double v) { CHECK_CLOSE(v, want[i], 1e-8); });
    vex::vector<cl_uint> U(ctx, n), V(ctx, n);
    U = 3;
    vex::exclusive_scan(U, V, cl_uint(10));
    check_sample(V, [&](size_t i, cl_uint v) { CHECK_EQUAL(v, cl_uint(10 + 3 * i)); });
}

TEST_CASE(sort_floats_is_sorted) {                                   // sort.cpp:9-20
    const size_t n = 1 << 20;
    std::vector<float> k = random_vector<float>(n);
    vex::vector<float> K(ctx, k);
    vex::sort(K);
    std::vector<float> got(n); vex::copy(K, got);
    CHECK(std::is_sorted(got.begin(), got.end()));
    std::sort(k.begin(), k.end());
    CHECK(got == k);
    vex::sort(K, vex::greater<float>());
    vex::copy(K, got);
    CHECK(std::is_sorted(got.begin(), got.end(), std::greater<float>()));
}

TEST_CASE(sort_by_key_equals_stable_sort) {                          // sort.cpp:22-45
    const size_t n = 1 << 20;
    std::vector<int> k = random_vector<int>(n);
    std::vector<float> v = random_vector<float>(n);
    vex::vector<int> K(ctx, k); vex::vector<float> V(ctx, v);
    vex::sort_by_key(K, V);
    std::vector<size_t> p(n); std::iota(p.begin(), p.end(), 0);
    std::stable_sort(p.begin(), p.end(), [&](size_t a, size_t b) { return k[a] < k[b]; });
    std::vector<int> gk(n); std::vector<float> gv(n); vex::copy(K, gk); vex::copy(V, gv);
    bool same = true;
    for (size_t i = 0; i < n; ++i) same = same && gk[i] == k[p[i]] && gv[i] == v[p[i]];
    CHECK(same);
    vex::vector<cl_ulong> L(ctx, n); L = vex::element_index() * 7919 % 100003;
    vex::sort(L);
    std::vector<cl_ulong> gl(n); vex::copy(L, gl);
    CHECK(std::is_sorted(gl.begin(), gl.end()));
}

// ---- stencil convolution: the reference's tests/stencil.cpp:17-110 --------------------
static size_t clamp_index(size_t n, size_t i, long shift) {
    return std::min<size_t>(n - 1, std::max<long>(0, static_cast<long>(i) + shift));
}

static void check_stencil(size_t n, size_t width, int center) {
    std::vector<double> s = random_vector<double>(width);
    vex::stencil<double> S(ctx, s, center);
    std::vector<double> x = random_vector<double>(n);
    vex::vector<double> X(ctx, x), Y(ctx, n);
    auto conv = [&](size_t i) { double sum = 0; long k = -center; for (size_t j = 0; j < s.size(); ++j, ++k) sum += s[j] * x[clamp_index(n, i, k)]; return sum; };
    Y = 1; Y += X * S;                                               // stencil.cpp:33-45
    std::vector<double> got(n); vex::copy(Y, got);
    for (size_t i = 0; i < n; ++i) CHECK_CLOSE(got[i], 1 + conv(i), 1e-8);
    Y = 42 * (X * S);                                                // :47-56
    vex::copy(Y, got);
    for (size_t i = 0; i < n; ++i) CHECK_CLOSE(got[i], 42 * conv(i), 1e-8);
    Y = X - S * X;                                                   // mixed with a vector term, either operand order
    vex::copy(Y, got);
    for (size_t i = 0; i < n; i += 7) CHECK_SMALL(got[i] - (x[i] - conv(i)), 1e-10 * width);
}

TEST_CASE(stencil_convolution) {
    check_stencil(1024, 1, 0);
    check_stencil(1024, 3, 1);
    check_stencil(1024, 17, 0);                                      // one-sided stencils
    check_stencil(1024, 17, 16);
    check_stencil(1024, 64, 23);
    check_stencil(1 << 20, 33, 16);
    check_stencil(5000, 9000, 4500);                                 // wider than LDS: direct-read kernel
}

TEST_CASE(stencil_small_vector_and_two_stencils) {                   // stencil.cpp:59-110
    check_stencil(128, 64, 40);                                      // halos longer than a neighbour's segment
    check_stencil(40, 37, 5);
    const size_t n = 32;
    std::vector<double> s(5, 1);
    vex::stencil<double> S(ctx, s, 3);
    vex::vector<double> X(ctx, n), Y(ctx, n);
    X = 0;
    Y = X * S + X * S;
    CHECK(Y[0] == 0 && Y[16] == 0 && Y[31] == 0);
    vex::stencil<float> F(ctx, {0.25f, 0.5f, 0.25f}, 1);            // initializer list, float
    vex::vector<float> A(ctx, 1000), B(ctx, 1000);
    A = 2.0f; B = A * F;
    check_sample(B, [](size_t, float v) { CHECK_CLOSE(v, 2.0f, 1e-5); });
}
