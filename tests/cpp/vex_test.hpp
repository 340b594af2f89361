// Minimal test harness for the C++ API tests (the reference uses Boost.Test,
// absent here).  Mirrors tests/context_setup.hpp of the reference: a global
// Context(DoublePrecision && Env); when only ONE device is present a second
// context on the same device is added, so that partitioning, ghost exchange and
// multi-device scan/sort/reduce run on a 2-"device" context (:24-39).
#ifndef VEX_TEST_HPP
#define VEX_TEST_HPP
#include <cmath>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <random>
#include <string>
#include <vector>
#include <vexcl/vexcl.hpp>

struct test_registry {
    struct entry { std::string name; std::function<void()> fn; };
    static std::vector<entry> &all() { static std::vector<entry> v; return v; }
    static int &failures() { static int f = 0; return f; }
    static int &checks() { static int c = 0; return c; }
};
struct test_registrar { test_registrar(const char *n, std::function<void()> f) { test_registry::all().push_back({n, f}); } };

#define TEST_CASE(name) static void name(); static test_registrar reg_##name(#name, name); static void name()
#define CHECK(cond) do { ++test_registry::checks(); if (!(cond)) { ++test_registry::failures(); \
    std::cerr << __FILE__ << ":" << __LINE__ << ": CHECK failed: " #cond << std::endl; } } while (0)
#define CHECK_EQUAL(a, b) do { ++test_registry::checks(); auto va = (a); auto vb = (b); if (!(va == vb)) { ++test_registry::failures(); \
    std::cerr << __FILE__ << ":" << __LINE__ << ": " #a " == " #b " failed: " << va << " != " << vb << std::endl; } } while (0)
// BOOST_CHECK_CLOSE semantics: tolerance in PERCENT
#define CHECK_CLOSE(a, b, pct) do { ++test_registry::checks(); double va = (a), vb = (b); \
    double d = std::fabs(va - vb), m = std::max(std::fabs(va), std::fabs(vb)); \
    if (!(d <= (pct) * 0.01 * m || d == 0)) { ++test_registry::failures(); \
    std::cerr << __FILE__ << ":" << __LINE__ << ": " #a " ~ " #b " failed: " << va << " vs " << vb << std::endl; } } while (0)
#define CHECK_SMALL(a, tol) do { ++test_registry::checks(); double va = (a); if (!(std::fabs(va) <= (tol))) { ++test_registry::failures(); \
    std::cerr << __FILE__ << ":" << __LINE__ << ": |" #a "| <= " #tol " failed: " << va << std::endl; } } while (0)

inline vex::Context &make_context() {
    static std::unique_ptr<vex::Context> ctx;
    if (!ctx) {
        ctx.reset(new vex::Context(vex::Filter::DoublePrecision && vex::Filter::Env));
        if (ctx->size() == 1 && !std::getenv("VEX_TEST_SINGLE_DEVICE")) {
            vex::Context second(vex::Filter::DoublePrecision && vex::Filter::Env);
            std::vector<std::pair<vex::backend::context, vex::backend::command_queue>> both;
            both.push_back(std::make_pair(ctx->context(0), ctx->queue(0)));
            both.push_back(std::make_pair(second.context(0), second.queue(0)));
            ctx.reset(new vex::Context(both));
        }
    }
    return *ctx;
}
#define ctx (make_context())

inline std::mt19937_64 &test_rng() { static std::mt19937_64 r(20240917); return r; }

template <class T> std::vector<T> random_vector(size_t n) {
    std::vector<T> x(n);
    if (std::is_floating_point<T>::value) { std::uniform_real_distribution<double> d(0, 1); for (auto &v : x) v = static_cast<T>(d(test_rng())); }
    else { std::uniform_int_distribution<long> d(0, 100); for (auto &v : x) v = static_cast<T>(d(test_rng())); }
    return x;
}

// tests/random_matrix.hpp shape: width in [0, nnz_per_row-1], distinct sorted columns
template <class RT, class CT, class VT>
void random_matrix(size_t n, size_t m, size_t nnz_per_row, std::vector<RT> &row, std::vector<CT> &col, std::vector<VT> &val) {
    row.clear(); col.clear();
    std::uniform_int_distribution<size_t> rw(0, nnz_per_row - 1), rc(0, m - 1);
    row.push_back(0);
    for (size_t k = 0; k < n; ++k) {
        size_t width = rw(test_rng());
        std::set<CT> cs;
        while (cs.size() < width) cs.insert(static_cast<CT>(rc(test_rng())));
        for (auto c : cs) col.push_back(c);
        row.push_back(static_cast<RT>(col.size()));
    }
    val = random_vector<VT>(col.size());
}

// reads 32 random elements through vector::operator[] (tests/context_setup.hpp:53-81)
template <class V, class F> void check_sample(const V &v, F f) {
    if (!v.size()) return;
    std::uniform_int_distribution<size_t> d(0, v.size() - 1);
    for (int i = 0; i < 32; ++i) { size_t idx = d(test_rng()); f(idx, static_cast<typename V::value_type>(v[idx])); }
}
template <class V1, class V2, class F> void check_sample(const V1 &v1, const V2 &v2, F f) {
    if (!v1.size()) return;
    std::uniform_int_distribution<size_t> d(0, v1.size() - 1);
    for (int i = 0; i < 32; ++i) { size_t idx = d(test_rng()); f(idx, static_cast<typename V1::value_type>(v1[idx]), static_cast<typename V2::value_type>(v2[idx])); }
}

#ifndef VEX_TEST_NO_MAIN
int main(int argc, char **argv) {
    try {
        bool need_device = true;
#ifdef VEX_TEST_CPU_ONLY
        need_device = false;
#endif
        if (need_device) {
            if (!ctx) { std::cerr << "no compute devices" << std::endl; return 2; }
            std::cout << ctx << std::endl;
        }
        std::string only = argc > 1 ? argv[1] : "";
        for (auto &t : test_registry::all()) {
            if (!only.empty() && t.name != only) continue;
            int before = test_registry::failures();
            t.fn();
            std::cout << (test_registry::failures() == before ? "[ ok ] " : "[FAIL] ") << t.name << std::endl;
        }
    } catch (const vex::error &e) {
        std::cerr << "vex::error: " << e.what() << std::endl; return 3;
    } catch (const std::exception &e) {
        std::cerr << "exception: " << e.what() << std::endl; return 3;
    }
    std::cout << test_registry::checks() << " checks, " << test_registry::failures() << " failures" << std::endl;
    return test_registry::failures() ? 1 : 0;
}
#endif
#endif
