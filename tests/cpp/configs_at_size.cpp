// Every BASELINE.json configuration VERIFIED at its stated size through the vex:: API (not only timed):
//   configs[1]  a = b*c + sin(d), n = 1e8 doubles       every element against the host's libm, |d| <= 1e-14 (|b c| + |sin d|)
//   configs[4]  vex::sort of 1e9 uint32 keys            sorted + the same multiset (sum, xor, sum of squares mod 2^64 of the keys)
//               vex::sort_by_key, 2.5e8 keys            keys[perm] == sorted keys, perm is a permutation, equal keys keep their order
//               vex::inclusive_scan of 1e9 uint32       out[0] == in[0], out[i] - out[i-1] == in[i] for every i (=> out is THE scan),
//                                                       out[n-1] == sum(in) mod 2^32
// Semantics: /root/reference/tests/sort.cpp:9-45, tests/scan.cpp:9-41, tests/vector_arithmetics.cpp.  Inputs are made on the
// host from a counter hash (nothing the device computes is trusted for the check); results are copied back and checked by
// host threads.  One device (the reference's multi-device sort is a host merge, outside the measured configuration).
// Usage: configs_at_size [scale = 1.0]   (scale < 1 shrinks every size; the GPU test runs scale 1)
#define VEX_TEST_NO_MAIN
#include "vex_test.hpp"
#undef ctx
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <thread>

static inline uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

template <class F> static void parallel(size_t n, F f) {
    const unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back([=] { f(t, n * t / nt, n * (t + 1) / nt); });
    for (auto &x : th) x.join();
}

struct multiset_sig { uint64_t sum = 0, x = 0, sq = 0; bool operator==(const multiset_sig &o) const { return sum == o.sum && x == o.x && sq == o.sq; } };
static multiset_sig signature(const std::vector<cl_uint> &k) {
    std::vector<multiset_sig> part(32);
    parallel(k.size(), [&](unsigned t, size_t a, size_t b) {
        multiset_sig s;
        for (size_t i = a; i < b; ++i) { const uint64_t v = k[i]; s.sum += v; s.x ^= v; s.sq += v * v; }
        part[t] = s;
    });
    multiset_sig s;
    for (auto &p : part) { s.sum += p.sum; s.x ^= p.x; s.sq += p.sq; }
    return s;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const double scale = argc > 1 ? std::atof(argv[1]) : 1.0;
    try {
        vex::Context ctx(vex::Filter::DoublePrecision && vex::Filter::Env && vex::Filter::Count(1));
        if (!ctx) { std::cerr << "no compute devices" << std::endl; return 2; }
        std::cout << ctx << std::endl;

        {   // ---- configs[1]: fused elementwise at n = 1e8
            const size_t n = (size_t)(1e8 * scale);
            std::vector<double> b(n), c(n), d(n), a(n);
            parallel(n, [&](unsigned, size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    b[i] = (double)(mix(3 * i) >> 11) * 0x1p-53 * 4 - 2;            // [-2, 2)
                    c[i] = (double)(mix(3 * i + 1) >> 11) * 0x1p-53 * 4 - 2;
                    d[i] = ((double)(mix(3 * i + 2) >> 11) * 0x1p-53 - 0.5) * 200;   // [-100, 100): argument reduction is exercised
                }
            });
            vex::vector<double> A(ctx, n), B(ctx, b), C(ctx, c), D(ctx, d);
            A = B * C + sin(D);
            vex::copy(A, a);
            std::vector<double> worst(32, 0.0); std::vector<size_t> bad(32, 0);
            parallel(n, [&](unsigned t, size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    const double s = std::sin(d[i]), want = b[i] * c[i] + s;
                    const double tol = 1e-14 * (std::fabs(b[i] * c[i]) + std::fabs(s));
                    const double err = std::fabs(a[i] - want);
                    if (!(err <= tol)) ++bad[t];
                    if (tol > 0) worst[t] = std::max(worst[t], err / tol);
                }
            });
            size_t nbad = 0; double w = 0;
            for (int t = 0; t < 32; ++t) { nbad += bad[t]; w = std::max(w, worst[t]); }
            ++test_registry::checks();
            if (nbad) { ++test_registry::failures(); std::cerr << "elementwise: " << nbad << " of " << n << " elements outside 1e-14 * sum|terms|" << std::endl; }
            std::printf("{\"config\": 1, \"what\": \"a = b*c + sin(d)\", \"n\": %zu, \"elements_checked\": %zu, \"outside_tolerance\": %zu, \"worst_err_over_tol\": %.3g, \"tolerance\": \"1e-14 * (|b c| + |sin d|) per element, against the host's libm\"}\n", n, n, nbad, w);
        }

        {   // ---- configs[4]: sort of 1e9 uint32 keys
            const size_t n = (size_t)(1e9 * scale);
            std::vector<cl_uint> k(n);
            parallel(n, [&](unsigned, size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) k[i] = (cl_uint)mix(i ^ 0x5bd1e995ull); });
            const multiset_sig before = signature(k);
            vex::vector<cl_uint> K(ctx, k);
            double t0 = now();
            vex::sort(K);
            ctx.finish();
            const double sort_s = now() - t0;
            std::vector<cl_uint> got(n);
            vex::copy(K, got);
            const multiset_sig after = signature(got);
            std::vector<size_t> inv(32, 0);
            parallel(n, [&](unsigned t, size_t lo, size_t hi) { for (size_t i = std::max<size_t>(lo, 1); i < hi; ++i) if (got[i - 1] > got[i]) ++inv[t]; });
            size_t ninv = 0; for (auto v : inv) ninv += v;
            CHECK_EQUAL(ninv, size_t(0));
            CHECK(before == after);
            std::printf("{\"config\": 4, \"what\": \"vex::sort uint32\", \"n\": %zu, \"inversions\": %zu, \"multiset_equal\": %s, \"sum\": %llu, \"xor\": %llu, \"sum_sq\": %llu, \"wall_s_incl_jit\": %.4f}\n",
                        n, ninv, before == after ? "true" : "false", (unsigned long long)after.sum, (unsigned long long)after.x, (unsigned long long)after.sq, sort_s);

            // ---- inclusive scan of the same 1e9 keys (mod 2^32)
            vex::vector<cl_uint> IN(ctx, k), OUT(ctx, n);
            vex::inclusive_scan(IN, OUT);
            vex::copy(OUT, got);
            std::vector<size_t> bad(32, 0); std::vector<uint64_t> tot(32, 0);
            parallel(n, [&](unsigned t, size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    const cl_uint prev = i ? got[i - 1] : 0u;
                    if ((cl_uint)(got[i] - prev) != k[i]) ++bad[t];
                    tot[t] += k[i];
                }
            });
            size_t nbad = 0; uint64_t total = 0; for (int t = 0; t < 32; ++t) { nbad += bad[t]; total += tot[t]; }
            CHECK_EQUAL(nbad, size_t(0));
            CHECK_EQUAL(got[n - 1], (cl_uint)total);
            std::printf("{\"config\": 4, \"what\": \"vex::inclusive_scan uint32\", \"n\": %zu, \"differences_wrong\": %zu, \"last\": %u, \"sum_mod_2_32\": %u}\n",
                        n, nbad, got[n - 1], (cl_uint)total);
        }

        {   // ---- sort_by_key: 2.5e8 keys with heavy duplication (65536 distinct), values = original positions
            const size_t n = (size_t)(2.5e8 * scale);
            std::vector<cl_uint> k(n), v(n);
            parallel(n, [&](unsigned, size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) { k[i] = (cl_uint)(mix(i + 77) & 0xFFFF) * 65537u; v[i] = (cl_uint)i; } });
            vex::vector<cl_uint> K(ctx, k), V(ctx, v);
            vex::sort_by_key(K, V);
            std::vector<cl_uint> gk(n), gv(n);
            vex::copy(K, gk); vex::copy(V, gv);
            std::vector<size_t> bad(32, 0);
            std::vector<unsigned char> seen(n, 0);
            parallel(n, [&](unsigned t, size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    if (gv[i] >= n || k[gv[i]] != gk[i]) { ++bad[t]; continue; }          // keys[perm] == sorted keys
                    if (i) {
                        if (gk[i - 1] > gk[i]) ++bad[t];                                  // sorted
                        else if (gk[i - 1] == gk[i] && gv[i - 1] >= gv[i]) ++bad[t];      // stable (and no value twice within a run)
                    }
                }
            });
            for (size_t i = 0; i < n; ++i) if (gv[i] < n) seen[gv[i]] = 1;                // a permutation: every position once
            size_t nbad = 0, nseen = 0; for (auto b : bad) nbad += b; for (auto s : seen) nseen += s;
            CHECK_EQUAL(nbad, size_t(0));
            CHECK_EQUAL(nseen, n);
            std::printf("{\"config\": 4, \"what\": \"vex::sort_by_key uint32 -> uint32\", \"n\": %zu, \"violations\": %zu, \"distinct_positions\": %zu}\n", n, nbad, nseen);
        }
    } catch (const vex::error &e) {
        std::cerr << "vex::error: " << e.what() << std::endl; return 3;
    } catch (const std::exception &e) {
        std::cerr << "exception: " << e.what() << std::endl; return 3;
    }
    std::cout << test_registry::checks() << " checks, " << test_registry::failures() << " failures" << std::endl;
    return test_registry::failures() ? 1 : 0;
}
