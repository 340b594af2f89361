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
