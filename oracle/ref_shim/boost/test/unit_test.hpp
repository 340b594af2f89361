// A minimal stand-in for <boost/test/unit_test.hpp> (Boost is absent from this image), so that the
// REFERENCE'S OWN test sources (/root/reference/tests/*.cpp) compile, unchanged and in place, against
// this repository's vexcl/ headers (oracle/build_ref.sh -> oracle/_ref/).  Test infrastructure only.
// Semantics follow Boost.Test: CHECK records a failure and continues, REQUIRE aborts the test case,
// BOOST_CHECK_CLOSE takes its tolerance in PERCENT, a global fixture lives for the whole run, a
// fixture suite constructs the fixture once per test case.
#ifndef VEX_REF_SHIM_BOOST_TEST_HPP
#define VEX_REF_SHIM_BOOST_TEST_HPP
#include <cmath>
#include <cstdlib>
#include <exception>
#include <functional>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include <tuple>
#include <boost/math/constants/constants.hpp>   // the reference's constants.hpp makes these visible to its tests
// boost::fusion::vector_tie, which the reference's headers bring in for tuples of keys / values: the
// tuple type of this implementation is std::tuple.
namespace boost { namespace fusion {
template <class... T> std::tuple<T &...> vector_tie(T &... t) { return std::tuple<T &...>(t...); }
} }

// boost::proto::display_expr(as_child(e)) in tests/deduce.cpp prints the Proto tree of an expression; the
// expression templates here are not Proto trees, so the print is a no-op (the test's assertions are on types).
namespace boost { namespace proto {
template <class T> const T &as_child(const T &t) { return t; }
template <class T> void display_expr(const T &) {}
} }

namespace shim {
struct state {
    static int &failures() { static int f = 0; return f; }
    static int &checks() { static int c = 0; return c; }
    struct entry { std::string name; std::function<void()> run; };
    static std::vector<entry> &tests() { static std::vector<entry> t; return t; }
    static std::vector<std::function<void *()>> &fixtures() { static std::vector<std::function<void *()>> f; return f; }
};
struct require_failed : std::exception {};
struct registrar { registrar(const char *n, std::function<void()> f) { state::tests().push_back({n, f}); } };
struct fixture_registrar { fixture_registrar(std::function<void *()> f) { state::fixtures().push_back(f); } };
inline void fail(const char *file, int line, const std::string &what) {
    ++state::failures();
    std::cerr << file << ":" << line << ": error: " << what << std::endl;
}
template <class A, class B> std::string show2(const A &a, const B &b) { std::ostringstream s; s << " [" << a << " != " << b << "]"; return s.str(); }
struct no_fixture {};
}

#define BOOST_GLOBAL_FIXTURE(F) static shim::fixture_registrar shim_fixture_##F([]() -> void * { return new F(); })
// a suite is a namespace (as in Boost.Test); the fixture of the innermost suite is found by name lookup
typedef shim::no_fixture shim_suite_fixture;
#define BOOST_FIXTURE_TEST_SUITE(name, F) namespace name { typedef F shim_suite_fixture;
#define BOOST_AUTO_TEST_SUITE(name) namespace name {
#define BOOST_AUTO_TEST_SUITE_END() }
#define BOOST_VERSION 106500

#define BOOST_AUTO_TEST_CASE(name)                                                               \
    struct shim_case_##name : shim_suite_fixture { void test_method(); };                        \
    static shim::registrar shim_reg_##name(#name, []() { shim_case_##name t; t.test_method(); });\
    void shim_case_##name::test_method()

#define BOOST_CHECK(cond) do { ++shim::state::checks(); if (!(cond)) shim::fail(__FILE__, __LINE__, "check " #cond " has failed"); } while (0)
#define BOOST_REQUIRE(cond) do { ++shim::state::checks(); if (!(cond)) { shim::fail(__FILE__, __LINE__, "critical check " #cond " has failed"); throw shim::require_failed(); } } while (0)
#define BOOST_CHECK_EQUAL(a, b) do { ++shim::state::checks(); auto &&shim_a = (a); auto &&shim_b = (b); \
    if (!(shim_a == shim_b)) shim::fail(__FILE__, __LINE__, "check " #a " == " #b " has failed" + shim::show2(shim_a, shim_b)); } while (0)
#define BOOST_CHECK_CLOSE(a, b, pct) do { ++shim::state::checks(); const double shim_a = static_cast<double>(a), shim_b = static_cast<double>(b); \
    const double shim_d = std::fabs(shim_a - shim_b); \
    const bool shim_ok = shim_d == 0 || (shim_d <= (pct) * 0.01 * std::fabs(shim_a) && shim_d <= (pct) * 0.01 * std::fabs(shim_b)); \
    if (!shim_ok) shim::fail(__FILE__, __LINE__, "difference between " #a " and " #b " exceeds " #pct "%" + shim::show2(shim_a, shim_b)); } while (0)
#define BOOST_CHECK_SMALL(a, tol) do { ++shim::state::checks(); const double shim_a = static_cast<double>(a); \
    if (!(std::fabs(shim_a) <= (tol))) shim::fail(__FILE__, __LINE__, "absolute value of " #a " exceeds " #tol + shim::show2(shim_a, tol)); } while (0)
#define BOOST_CHECK_THROW(expr, exc) do { ++shim::state::checks(); bool shim_thrown = false; try { expr; } catch (const exc &) { shim_thrown = true; } catch (...) {} \
    if (!shim_thrown) shim::fail(__FILE__, __LINE__, "exception " #exc " is expected"); } while (0)
#define BOOST_CHECK_NO_THROW(expr) do { ++shim::state::checks(); try { expr; } catch (...) { shim::fail(__FILE__, __LINE__, "unexpected exception thrown by " #expr); } } while (0)

int main(int argc, char **argv) {
    std::vector<void *> keep;
    try {
        for (auto &f : shim::state::fixtures()) keep.push_back(f());
    } catch (const std::exception &e) {
        std::cerr << "fixture setup failed: " << e.what() << std::endl;
        return 3;
    }
    const std::string only = argc > 1 ? argv[1] : "";
    int ran = 0;
    for (auto &t : shim::state::tests()) {
        if (!only.empty() && t.name != only) continue;
        const int before = shim::state::failures();
        try { t.run(); }
        catch (const shim::require_failed &) {}
        catch (const std::exception &e) { shim::fail(t.name.c_str(), 0, std::string("uncaught exception: ") + e.what()); }
        catch (...) { shim::fail(t.name.c_str(), 0, "uncaught exception of unknown type"); }
        ++ran;
        std::cout << (shim::state::failures() == before ? "[ ok ] " : "[FAIL] ") << t.name << std::endl;
    }
    std::cout << "*** " << ran << " test cases, " << shim::state::checks() << " assertions, " << shim::state::failures() << " failures" << std::endl;
    return shim::state::failures() ? 1 : 0;
}
#endif
