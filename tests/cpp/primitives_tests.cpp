// GPU: vex::inclusive_scan / exclusive_scan / sort / sort_by_key, the checks of
// the reference's tests/scan.cpp:9-41 and tests/sort.cpp:9-45, on the
// 2-"device" context (cross-device carry and merge are exercised).
#include "vex_test.hpp"
#include <numeric>
#include <cstring>
#include <cmath>

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

// ---- scans with a user operator (reference API: scan.hpp:426-518, `oper.device` + host operator()) ----
namespace {
struct running_max_t { VEX_DUAL_FUNCTOR(int, (int, a)(int, b), return a > b ? a : b;) };
struct affine_t {                                                    // composition of x -> a*x + b packed in a long: NOT commutative
    VEX_DUAL_FUNCTOR(long, (long, f)(long, g),
        const long fa = f >> 32, fb = f & 0xffffffffl, ga = g >> 32, gb = g & 0xffffffffl;
        return (((fa * ga) & 0xffffl) << 32) | ((ga * fb + gb) & 0xffffl);)
};
}

TEST_CASE(scan_with_user_operators) {
    running_max_t mx; affine_t comp;
    for (size_t n : {size_t(1), size_t(777), size_t(4096), size_t(250001)}) {
        std::vector<int> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = int((i * 2654435761u) % 100000) - 50000 + int(i / 8);
        vex::vector<int> X(ctx, h), Y(ctx, n);
        const int lowest = std::numeric_limits<int>::min();
        vex::inclusive_scan(X, Y, lowest, mx);
        std::vector<int> got(n), want(n);
        vex::copy(Y, got);
        int run = lowest;
        for (size_t i = 0; i < n; ++i) { run = mx(run, h[i]); want[i] = run; }
        CHECK(got == want);
        vex::exclusive_scan(X, Y, lowest, mx);
        vex::copy(Y, got);
        run = lowest;
        for (size_t i = 0; i < n; ++i) { want[i] = run; run = mx(run, h[i]); }
        CHECK(got == want);

        std::vector<long> f(n);
        for (size_t i = 0; i < n; ++i) f[i] = (long(1 + i % 5) << 32) | long(i % 97);
        vex::vector<long> F(ctx, f), G(ctx, n);
        vex::inclusive_scan(F, G, long(1) << 32, comp);                // identity: x -> 1*x + 0
        std::vector<long> gl(n), wl(n);
        vex::copy(G, gl);
        long acc = f[0]; wl[0] = acc;
        for (size_t i = 1; i < n; ++i) { acc = comp(acc, f[i]); wl[i] = acc; }
        CHECK(gl == wl);                                               // associativity holds exactly (integer arithmetic mod 2^k)
    }
}

// ---- user comparators and tied keys: the generated merge sort (reference: tests/sort.cpp:47-200) ----
namespace {
struct low_nibble_first_t {                                          // many ties: 16 classes -- stability is visible
    VEX_DUAL_FUNCTOR(bool, (int, a)(int, b), return (a & 15) < (b & 15);)
};
struct pair_less_t {
    VEX_DUAL_FUNCTOR(bool, (int, a1)(float, a2)(int, b1)(float, b2), return (a1 == b1) ? (a2 < b2) : (a1 < b1);)
};
}

TEST_CASE(sort_with_user_comparator_is_stable_at_every_size) {
    low_nibble_first_t cmp;
    for (size_t n : {size_t(0), size_t(1), size_t(2), size_t(3), size_t(100), size_t(2047), size_t(2048), size_t(2049), size_t(4097),
                     size_t(65536), size_t(100003), size_t(1 << 20) + 13}) {
        std::vector<int> k(n);
        for (size_t i = 0; i < n; ++i) k[i] = int((i * 2654435761u) >> 7);
        std::vector<int> pos(n); std::iota(pos.begin(), pos.end(), 0);
        vex::vector<int> K(ctx, n), P(ctx, n);
        if (n) { vex::copy(k, K); vex::copy(pos, P); }
        vex::sort_by_key(K, P, cmp);
        std::vector<int> want(pos);
        // per partition the device sort is stable; across partitions the host merge is: the whole is std::stable_sort
        std::stable_sort(want.begin(), want.end(), [&](int a, int b) { return cmp(k[a], k[b]); });
        std::vector<int> gk(n), gp(n);
        if (n) { vex::copy(K, gk); vex::copy(P, gp); }
        bool same = true;
        for (size_t i = 0; i < n; ++i) same = same && gp[i] == want[i] && gk[i] == k[want[i]];
        CHECK(same);
        vex::vector<int> K2(ctx, n);
        if (n) vex::copy(k, K2);
        vex::sort(K2, cmp);                                            // keys only
        if (n) vex::copy(K2, gk);
        for (size_t i = 0; i < n; ++i) same = same && gk[i] == k[want[i]];
        CHECK(same);
    }
}

TEST_CASE(sort_tied_keys_and_tied_values) {                           // sort.cpp:85-200
    pair_less_t less;
    const size_t n = 300007;
    std::vector<int> k1(n); std::vector<float> k2(n); std::vector<cl_long> v1(n); std::vector<short> v2(n);
    for (size_t i = 0; i < n; ++i) { k1[i] = rand() % 50; k2[i] = float(rand() % 7); v1[i] = (cl_long)i * 1000003; v2[i] = short(i); }
    vex::vector<int> K1(ctx, k1); vex::vector<float> K2(ctx, k2); vex::vector<cl_long> V1(ctx, v1); vex::vector<short> V2(ctx, v2);
    std::vector<size_t> p(n); std::iota(p.begin(), p.end(), size_t(0));
    std::stable_sort(p.begin(), p.end(), [&](size_t i, size_t j) { return less(k1[i], k2[i], k1[j], k2[j]); });
    vex::sort_by_key(std::tie(K1, K2), std::tie(V1, V2), less);
    std::vector<int> g1(n); std::vector<float> g2(n); std::vector<cl_long> w1(n); std::vector<short> w2(n);
    vex::copy(K1, g1); vex::copy(K2, g2); vex::copy(V1, w1); vex::copy(V2, w2);
    bool same = true;
    for (size_t i = 0; i < n; ++i) same = same && g1[i] == k1[p[i]] && g2[i] == k2[p[i]] && w1[i] == v1[p[i]] && w2[i] == v2[p[i]];
    CHECK(same);
    // default comparator, values that the radix path cannot carry (2 bytes; two vectors): permuted afterwards
    vex::vector<int> K(ctx, k1); vex::copy(v2, V2); vex::copy(v1, V1);
    vex::sort_by_key(K, std::tie(V1, V2), vex::less<int>());
    std::iota(p.begin(), p.end(), size_t(0));
    std::stable_sort(p.begin(), p.end(), [&](size_t i, size_t j) { return k1[i] < k1[j]; });
    vex::copy(K, g1); vex::copy(V1, w1); vex::copy(V2, w2);
    same = true;
    for (size_t i = 0; i < n; ++i) same = same && g1[i] == k1[p[i]] && w1[i] == v1[p[i]] && w2[i] == v2[p[i]];
    CHECK(same);
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
    check_stencil(1001, 5, 2);                                       // sizes that are not multiples of the 4 outputs a lane folds
    check_stencil(4099, 21, 10);
    check_stencil(2050, 2, 1);
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

// ---- scan_by_key / reduce_by_key (tests/scan_by_key.cpp, tests/reduce_by_key.cpp) -------------
// Full comparison against a serial loop: bit-exact for integers; floating point sums are
// associated as a tree on the device, so they are compared with the reference's 1e-8 % tolerance.
template <class T> static bool all_close(const std::vector<T> &a, const std::vector<T> &b, double pct = 1e-8) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i) {
        double d = std::fabs(double(a[i]) - double(b[i])), m = std::max(std::fabs(double(a[i])), std::fabs(double(b[i])));
        if (d > pct * 0.01 * m) return false;
    }
    return true;
}
template <class K, class V>
static void serial_scan_by_key(const std::vector<K> &k, const std::vector<V> &v, std::vector<V> &incl, std::vector<V> &excl, V init) {
    incl.resize(v.size()); excl.resize(v.size());
    for (size_t i = 0; i < v.size(); ++i) {
        bool head = i == 0 || !(k[i - 1] == k[i]);
        incl[i] = head ? v[i] : incl[i - 1] + v[i];
        excl[i] = head ? init : init + incl[i - 1];
    }
}

TEST_CASE(scan_by_key_default_functions) {                           // scan_by_key.cpp:17-61
    std::vector<vex::backend::command_queue> queue(1, ctx.queue(0));
    for (size_t n : {size_t(1), size_t(63), size_t(1000), size_t(2048), size_t(2049), size_t(1) << 20, (size_t(1) << 20) + 77}) {
        std::vector<int> x = random_vector<int>(n), y = random_vector<int>(n);
        std::sort(x.begin(), x.end());
        if (n > 5000) for (size_t i = 3000; i < 3000 + 2 * 2048 + 100 && i < n; ++i) x[i] = x[3000];   // a run spanning whole tiles
        std::sort(x.begin(), x.end());
        vex::vector<int> ikeys(queue, x), ivals(queue, y), ovals(queue, n);
        std::vector<int> incl, excl, got(n);
        serial_scan_by_key(x, y, incl, excl, 0);
        vex::inclusive_scan_by_key(ikeys, ivals, ovals);
        vex::copy(ovals, got); CHECK(got == incl);
        vex::exclusive_scan_by_key(ikeys, ivals, ovals);
        vex::copy(ovals, got); CHECK(got == excl);
        serial_scan_by_key(x, y, incl, excl, 5);
        vex::exclusive_scan_by_key(ikeys, ivals, ovals, 5);
        vex::copy(ovals, got); CHECK(got == excl);
    }
    // doubles
    const size_t n = 300000;
    std::vector<cl_long> k(n); for (size_t i = 0; i < n; ++i) k[i] = cl_long(i / 777);
    std::vector<double> v = random_vector<double>(n), incl, excl, got(n);
    serial_scan_by_key(k, v, incl, excl, 0.0);
    vex::vector<cl_long> K(queue, k); vex::vector<double> V(queue, v), O(queue, n);
    vex::inclusive_scan_by_key(K, V, O);
    vex::copy(O, got); CHECK(all_close(got, incl));
    std::vector<double> again(n);
    vex::inclusive_scan_by_key(K, V, V);                               // in place; reproducible bit for bit
    vex::copy(V, again); CHECK(again == got);
}

TEST_CASE(by_key_single_pass_against_three_phases) {
    // Round 3: the single-pass look-back form (default) against the serial loop AND against the three deterministic phases
    // (VEXCL_SCAN_BY_KEY=tree), on runs of every length -- one element up to 40 tiles of 4096 -- with values whose sums are
    // exact (integers; doubles holding small integers), so every comparison is bit for bit.
    std::vector<vex::backend::command_queue> queue(1, ctx.queue(0));
    std::mt19937_64 rng(77);
    for (size_t n : {size_t(4095), size_t(4096), size_t(4097), size_t(1000003), size_t(5 * 1024 * 1024 + 11)}) {
        std::vector<int> k(n); std::vector<double> v(n); std::vector<int> vi(n);
        int key = 0;
        for (size_t i = 0; i < n;) {
            const size_t choices[] = {1, 2, 63, 64, 65, 777, 4096, 4097, 9000, 40 * 4096 + 5};
            size_t len = choices[rng() % 10];
            for (size_t j = 0; j < len && i < n; ++j, ++i) { k[i] = key; vi[i] = int(rng() % 1000) - 500; v[i] = double(vi[i]); }
            ++key;
        }
        vex::vector<int> K(queue, k), VI(queue, vi), OI(queue, n);
        vex::vector<double> V(queue, v), O(queue, n);
        std::vector<double> incl, excl, got(n), tree(n);
        std::vector<int> incli, excli, goti(n);
        serial_scan_by_key(k, v, incl, excl, 3.0);
        serial_scan_by_key(k, vi, incli, excli, 3);
        vex::inclusive_scan_by_key(K, V, O); vex::copy(O, got); CHECK(got == incl);
        vex::exclusive_scan_by_key(K, V, O, 3.0); vex::copy(O, got); CHECK(got == excl);
        vex::inclusive_scan_by_key(K, VI, OI); vex::copy(OI, goti); CHECK(goti == incli);
        vex::exclusive_scan_by_key(K, VI, OI, 3); vex::copy(OI, goti); CHECK(goti == excli);
        setenv("VEXCL_SCAN_BY_KEY", "tree", 1);
        vex::inclusive_scan_by_key(K, V, O); vex::copy(O, tree); CHECK(tree == incl);
        unsetenv("VEXCL_SCAN_BY_KEY");
        // reduce_by_key: run count by the keys-only kernel, sums by the single pass; outputs of the right size are re-used
        vex::vector<int> OK; vex::vector<double> OV;
        std::vector<int> uk; std::vector<double> us;
        for (size_t i = 0; i < n; ++i) { if (i == 0 || k[i - 1] != k[i]) { uk.push_back(k[i]); us.push_back(v[i]); } else us.back() += v[i]; }
        for (int rep = 0; rep < 2; ++rep) {
            const int runs = vex::reduce_by_key(K, V, OK, OV);
            CHECK_EQUAL(size_t(runs), uk.size());
            std::vector<int> gk(uk.size()); std::vector<double> gs(uk.size());
            vex::copy(OK, gk); vex::copy(OV, gs);
            CHECK(gk == uk); CHECK(gs == us);
        }
        setenv("VEXCL_SCAN_BY_KEY", "tree", 1);
        CHECK_EQUAL(size_t(vex::reduce_by_key(K, V, OK, OV)), uk.size());
        unsetenv("VEXCL_SCAN_BY_KEY");
        std::vector<double> gs(uk.size()); vex::copy(OV, gs); CHECK(gs == us);
    }
    // single precision values (two status words per tile) and 64-bit integer values (three)
    const size_t n = 700001;
    std::vector<cl_long> k(n); std::vector<float> f(n); std::vector<cl_long> l(n);
    for (size_t i = 0; i < n; ++i) { k[i] = cl_long(i / 5000); f[i] = float(int(i % 17) - 8); l[i] = cl_long(i) * 1000003; }
    vex::vector<cl_long> K(queue, k), L(queue, l), OL(queue, n); vex::vector<float> F(queue, f), OF(queue, n);
    std::vector<float> fi, fe, gf(n); std::vector<cl_long> li, le, gl(n);
    serial_scan_by_key(k, f, fi, fe, 0.f); serial_scan_by_key(k, l, li, le, cl_long(0));
    vex::inclusive_scan_by_key(K, F, OF); vex::copy(OF, gf); CHECK(gf == fi);
    vex::exclusive_scan_by_key(K, L, OL); vex::copy(OL, gl); CHECK(gl == le);
}

// Round 6: a floating-point carry that crosses SEVERAL tiles is folded serially from the nearest tile that holds a run head (or an
// inclusive prefix, which is such a fold itself): the bits no longer depend on how far the predecessors of a tile happened to be
// (scan_by_key.hpp sbk_look_back).  Values whose sums round at every step, runs from one element to 300 tiles of 16 Ki, the whole
// vector one run: twelve repetitions of the scan and of reduce_by_key must give the same bits, and they must be the serial
// loop's values up to rounding.
TEST_CASE(by_key_floating_point_carries_are_reproducible) {
    std::vector<vex::backend::command_queue> queue(1, ctx.queue(0));
    std::mt19937_64 rng(2026);
    for (int shape = 0; shape < 3; ++shape) {
        const size_t n = shape == 2 ? size_t(12) * 1024 * 1024 + 77 : size_t(20) * 1024 * 1024 + 5;
        std::vector<int> k(n); std::vector<double> v(n); std::vector<float> f(n);
        int key = 0;
        for (size_t i = 0; i < n;) {
            const size_t choices[] = {1, 5, 4096, 16384, 16385, 40000, 3 * 16384 + 1, 70 * 16384, 300 * 16384 + 3, 1000};
            size_t len = shape == 2 ? n : shape == 1 ? size_t(200) * 16384 + 11 : choices[rng() % 10];
            for (size_t j = 0; j < len && i < n; ++j, ++i) { k[i] = key; v[i] = std::ldexp(double(rng() >> 11), -53) - 0.25; f[i] = float(v[i]); }
            ++key;
        }
        vex::vector<int> K(queue, k); vex::vector<double> V(queue, v), O(queue, n); vex::vector<float> F(queue, f), OF(queue, n);
        std::vector<double> first(n), got(n), incl, excl;
        std::vector<float> ffirst(n), fgot(n);
        serial_scan_by_key(k, v, incl, excl, 0.0);
        vex::vector<int> OK; vex::vector<double> OV;
        std::vector<double> rfirst, rgot;
        size_t differing = 0, fdiffering = 0, rdiffering = 0;
        for (int rep = 0; rep < 12; ++rep) {
            vex::inclusive_scan_by_key(K, V, O); vex::copy(O, rep ? got : first);
            vex::inclusive_scan_by_key(K, F, OF); vex::copy(OF, rep ? fgot : ffirst);
            const int runs = vex::reduce_by_key(K, V, OK, OV);
            CHECK_EQUAL(runs, key);
            (rep ? rgot : rfirst).resize(runs); vex::copy(OV, rep ? rgot : rfirst);
            if (rep) {
                for (size_t i = 0; i < n; ++i) { differing += std::memcmp(&got[i], &first[i], 8) != 0; fdiffering += std::memcmp(&fgot[i], &ffirst[i], 4) != 0; }
                for (size_t i = 0; i < rfirst.size(); ++i) rdiffering += std::memcmp(&rgot[i], &rfirst[i], 8) != 0;
            }
        }
        CHECK_EQUAL(differing, size_t(0)); CHECK_EQUAL(fdiffering, size_t(0)); CHECK_EQUAL(rdiffering, size_t(0));
        double worst = 0;
        for (size_t i = 0; i < n; ++i) worst = std::max(worst, std::fabs(first[i] - incl[i]) / (1.0 + std::fabs(incl[i])));
        CHECK(worst < 1e-9);
        // the run's entry of reduce_by_key against the last element of the run in the scan (the two modes fold a lane's elements
        // differently: equal up to rounding, not bit for bit)
        size_t r = 0; double rworst = 0;
        for (size_t i = 0; i < n; ++i) if (i + 1 == n || k[i + 1] != k[i]) { rworst = std::max(rworst, std::fabs(first[i] - rfirst[r]) / (1.0 + std::fabs(first[i]))); ++r; }
        CHECK(rworst < 1e-9);
    }
}

// reduce_by_key takes the size its outputs already have as the likely number of runs and does ONE pass (scan_by_key.hpp,
// sbk::run); a wrong guess -- more runs than the outputs hold, or fewer -- must cost a second pass, never a wrong result
TEST_CASE(reduce_by_key_reuses_outputs_of_any_size) {
    std::vector<vex::backend::command_queue> queue(1, ctx.queue(0));
    const size_t n = 300007;
    vex::vector<int> OK; vex::vector<double> OV;
    const int run_lengths[] = {64, 64, 7, 7, 1000, 1, 64, 300007};
    for (int len : run_lengths) {
        std::vector<int> k(n); std::vector<double> v(n);
        for (size_t i = 0; i < n; ++i) { k[i] = int(i / size_t(len)) * 3 - 11; v[i] = double(int(i % 23) - 11) * 0.25; }
        std::vector<int> uk; std::vector<double> us;
        for (size_t i = 0; i < n; ++i) { if (i == 0 || k[i - 1] != k[i]) { uk.push_back(k[i]); us.push_back(v[i]); } else us.back() += v[i]; }
        vex::vector<int> K(queue, k); vex::vector<double> V(queue, v);
        const int runs = vex::reduce_by_key(K, V, OK, OV);
        CHECK_EQUAL(size_t(runs), uk.size());
        CHECK_EQUAL(OK.size(), uk.size()); CHECK_EQUAL(OV.size(), uk.size());
        std::vector<int> gk(uk.size()); std::vector<double> gs(uk.size());
        vex::copy(OK, gk); vex::copy(OV, gs);
        CHECK(gk == uk); CHECK(gs == us);
    }
}

VEX_FUNCTION(bool, pair_equal, (int, a1)(int, a2)(int, b1)(int, b2), return a1 == b1 && a2 == b2;);
VEX_FUNCTION(int, int_plus, (int, x)(int, y), return x + y;);
VEX_FUNCTION(int, int_max, (int, x)(int, y), return x > y ? x : y;);

TEST_CASE(scan_by_key_tuple_keys_user_functions) {                    // scan_by_key.cpp:63-124
    const size_t n = 100000;
    std::vector<vex::backend::command_queue> queue(1, ctx.queue(0));
    std::vector<int> x1(n), x2(n), y = random_vector<int>(n);
    for (size_t i = 0; i < n; ++i) { x1[i] = int(i / 100); x2[i] = int((i % 100) / 7); }
    vex::vector<int> k1(queue, x1), k2(queue, x2), ivals(queue, y), ovals(queue, n);
    auto same = [&](size_t i) { return i > 0 && x1[i - 1] == x1[i] && x2[i - 1] == x2[i]; };

    std::vector<int> want(n), got(n);
    vex::inclusive_scan_by_key(std::tie(k1, k2), ivals, ovals, pair_equal, int_plus);
    for (size_t i = 0; i < n; ++i) want[i] = same(i) ? want[i - 1] + y[i] : y[i];
    vex::copy(ovals, got); CHECK(got == want);

    vex::exclusive_scan_by_key(std::tie(k1, k2), ivals, ovals, pair_equal, int_plus);
    std::vector<int> incl = want;
    for (size_t i = 0; i < n; ++i) want[i] = same(i) ? incl[i - 1] : 0;
    vex::copy(ovals, got); CHECK(got == want);

    vex::inclusive_scan_by_key(std::tie(k1, k2), ivals, ovals, pair_equal, int_max);   // non-additive operator
    for (size_t i = 0; i < n; ++i) want[i] = same(i) ? std::max(want[i - 1], y[i]) : y[i];
    vex::copy(ovals, got); CHECK(got == want);
}

TEST_CASE(reduce_by_key_default_functions) {                          // reduce_by_key.cpp:9-43
    std::vector<vex::backend::command_queue> queue(1, ctx.queue(0));
    for (size_t n : {size_t(1), size_t(100), size_t(2048), size_t(1024 * 1024), size_t(1024 * 1024 + 3)}) {
        std::vector<int> x = random_vector<int>(n);
        std::vector<double> y = random_vector<double>(n);
        std::sort(x.begin(), x.end());
        vex::vector<int> ikeys(queue, x), okeys;
        vex::vector<double> ivals(queue, y), ovals;
        int num_keys = vex::reduce_by_key(ikeys, ivals, okeys, ovals);

        std::vector<int> uk; std::vector<double> us;
        for (size_t i = 0; i < n; ++i) {
            if (i == 0 || x[i - 1] != x[i]) { uk.push_back(x[i]); us.push_back(y[i]); } else us.back() += y[i];
        }
        CHECK_EQUAL(size_t(num_keys), uk.size());
        CHECK_EQUAL(okeys.size(), uk.size());
        CHECK_EQUAL(ovals.size(), uk.size());
        std::vector<int> gk(uk.size()); std::vector<double> gs(uk.size());
        vex::copy(okeys, gk); vex::copy(ovals, gs);
        CHECK(gk == uk);
        CHECK(all_close(gs, us));
    }
    vex::vector<int> ek(queue, 0), ok; vex::vector<double> ev(queue, 0), ov;
    CHECK_EQUAL(vex::reduce_by_key(ek, ev, ok, ov), 0);              // empty input
    // all keys distinct / all keys equal
    const size_t n = 10000;
    std::vector<int> x(n); std::iota(x.begin(), x.end(), 0);
    std::vector<double> y = random_vector<double>(n);
    vex::vector<int> ik(queue, x); vex::vector<double> iv(queue, y);
    CHECK_EQUAL(vex::reduce_by_key(ik, iv, ok, ov), int(n));
    std::vector<double> gs(n); vex::copy(ov, gs); CHECK(gs == y);
    ik = 7;
    CHECK_EQUAL(vex::reduce_by_key(ik, iv, ok, ov), 1);
    CHECK_EQUAL(int(ok[0]), 7);
    CHECK_CLOSE(double(ov[0]), std::accumulate(y.begin(), y.end(), 0.0), 1e-8);
}

VEX_FUNCTION(bool, key2_equal, (cl_int, a1)(cl_long, a2)(cl_int, b1)(cl_long, b2), return (a1 == b1) && (a2 == b2););
VEX_FUNCTION(double, dbl_plus, (double, x)(double, y), return x + y;);

TEST_CASE(reduce_by_key_tuple_keys) {                                 // reduce_by_key.cpp:45-129
    const size_t n = 1000 * 1000;
    std::vector<vex::backend::command_queue> queue(1, ctx.queue(0));
    std::vector<cl_int> k1(n); std::vector<cl_long> k2(n);
    {
        std::vector<cl_int> a = random_vector<cl_int>(n); std::vector<cl_long> b = random_vector<cl_long>(n);
        std::vector<size_t> idx(n); std::iota(idx.begin(), idx.end(), 0);
        std::sort(idx.begin(), idx.end(), [&](size_t i, size_t j) { return std::make_tuple(a[i], b[i]) < std::make_tuple(a[j], b[j]); });
        for (size_t i = 0; i < n; ++i) { k1[i] = a[idx[i]]; k2[i] = b[idx[i]]; }
    }
    std::vector<double> y = random_vector<double>(n);
    vex::vector<cl_int> ikey1(queue, k1), okey1; vex::vector<cl_long> ikey2(queue, k2), okey2;
    vex::vector<double> ivals(queue, y), ovals;
    int num_keys = vex::reduce_by_key(std::tie(ikey1, ikey2), ivals, std::tie(okey1, okey2), ovals, key2_equal, dbl_plus);

    std::vector<cl_int> rk1; std::vector<cl_long> rk2; std::vector<double> rsum;
    for (size_t i = 0; i < n; ++i) {
        if (i > 0 && k1[i - 1] == k1[i] && k2[i - 1] == k2[i]) rsum.back() += y[i];
        else { rk1.push_back(k1[i]); rk2.push_back(k2[i]); rsum.push_back(y[i]); }
    }
    CHECK_EQUAL(size_t(num_keys), rsum.size());
    CHECK_EQUAL(okey1.size(), rsum.size()); CHECK_EQUAL(okey2.size(), rsum.size()); CHECK_EQUAL(ovals.size(), rsum.size());
    std::vector<cl_int> g1(rsum.size()); std::vector<cl_long> g2(rsum.size()); std::vector<double> gs(rsum.size());
    vex::copy(okey1, g1); vex::copy(okey2, g2); vex::copy(ovals, gs);
    CHECK(g1 == rk1); CHECK(g2 == rk2); CHECK(all_close(gs, rsum));
}
