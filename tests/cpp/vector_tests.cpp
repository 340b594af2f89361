// GPU: vex::vector / expression / Reductor behaviour the reference pins in
// tests/vector_create.cpp, vector_copy.cpp, vector_arithmetics.cpp,
// custom_kernel.cpp, events.cpp, threads.cpp -- on the 2-"device" context.
#include "vex_test.hpp"
#include <thread>

TEST_CASE(vector_create_and_copy) {                                  // vector_create.cpp:7-211
    const size_t N = 1024;
    vex::vector<double> empty;
    CHECK(empty.size() == 0 && empty.end() - empty.begin() == 0);
    vex::vector<double> x(ctx, N);
    CHECK_EQUAL(x.size(), N);
    CHECK_EQUAL(x.nparts(), ctx.size());
    CHECK_EQUAL(size_t(x.end() - x.begin()), N);
    size_t total = 0; for (unsigned d = 0; d < x.nparts(); ++d) total += x.part_size(d);
    CHECK_EQUAL(total, N);

    std::vector<double> h = random_vector<double>(N);
    vex::vector<double> y(ctx, h);                                  // from host vector
    check_sample(y, [&](size_t i, double v) { CHECK_EQUAL(v, h[i]); });
    vex::vector<double> z(ctx, N, h.data());                        // from host pointer
    check_sample(z, [&](size_t i, double v) { CHECK_EQUAL(v, h[i]); });

    vex::vector<double> c(y);                                       // deep copy
    CHECK(c(0).raw() != y(0).raw());
    y = 0;
    check_sample(c, [&](size_t i, double v) { CHECK_EQUAL(v, h[i]); });

    double *before = c(0).raw();
    vex::vector<double> m(std::move(c));                            // move keeps the buffer
    CHECK(m(0).raw() == before && c.size() == 0);
    c = std::move(m);
    CHECK(c(0).raw() == before);
    c -= z;                                                         // buffer identity stable (vector_create.cpp:152-175)
    CHECK(c(0).raw() == before);
    check_sample(c, [&](size_t, double v) { CHECK_EQUAL(v, 0.0); });

    vex::vector<double> e = z * 2;                                  // expression constructor
    CHECK_EQUAL(e.size(), N);
    check_sample(e, [&](size_t i, double v) { CHECK_EQUAL(v, 2 * h[i]); });

    vex::vector<int> tiny(ctx, 5);                                  // some devices are empty (vector_create.cpp:189-194)
    tiny = 42;
    for (size_t i = 0; i < 5; ++i) CHECK_EQUAL(int(tiny[i]), 42);
    x.resize(ctx, 2 * N); CHECK_EQUAL(x.size(), 2 * N);
    vex::swap(x, e); CHECK_EQUAL(x.size(), N);
}

TEST_CASE(copy_and_element_access) {                                 // vector_copy.cpp
    const size_t N = 1 << 16;
    std::vector<double> h = random_vector<double>(N), back(N);
    vex::vector<double> x(ctx, N);
    vex::copy(h, x);
    vex::copy(x, back);
    CHECK(h == back);
    vex::copy(h.begin(), h.begin() + 100, x.begin() + 500);
    vex::copy(x.begin() + 500, x.begin() + 600, back.begin());
    for (size_t i = 0; i < 100; ++i) CHECK_EQUAL(back[i], h[i]);
    x[N - 1] = 7.5; CHECK_EQUAL(double(x[N - 1]), 7.5);
    x[0] = x[N - 1]; CHECK_EQUAL(double(x[0]), 7.5);
    double s = 0; for (auto it = x.begin(); it != x.begin() + 4; ++it) s += *it;
    CHECK(s > 7.5);
    bool thrown = false; try { x.at(N); } catch (const std::out_of_range &) { thrown = true; } CHECK(thrown);
    auto mapped = x.map(0); mapped[1] = 99.0; mapped.reset();       // unmap writes back
    CHECK_EQUAL(double(x[1]), 99.0);
}

TEST_CASE(arithmetics) {                                             // vector_arithmetics.cpp:33-48
    const size_t N = 1 << 20;
    vex::vector<double> x(ctx, N);
    vex::vector<double> y(ctx, random_vector<double>(N));
    vex::vector<double> z(ctx, random_vector<double>(N));
    x = 5 * sin(y) + z;
    check_sample(x, y, [&](size_t, double a, double b) { (void)a; (void)b; });
    std::vector<double> hx(N), hy(N), hz(N);
    vex::copy(x, hx); vex::copy(y, hy); vex::copy(z, hz);
    for (size_t i = 0; i < N; i += 997) CHECK_CLOSE(hx[i], 5 * std::sin(hy[i]) + hz[i], 1e-12);
    x = pow(sin(y), 2.0) + pow(cos(y), 2.0);                        // :101-111
    check_sample(x, [](size_t, double v) { CHECK_CLOSE(v, 1.0, 1e-8); });
    x = 17; x += 3; x -= 5; x *= 2; x /= 6;
    check_sample(x, [](size_t, double v) { CHECK_CLOSE(v, 5.0, 1e-12); });
    vex::vector<int> k(ctx, N); k = 13; k %= 5; k <<= 2; k |= 1; k ^= 3; k &= 14; k >>= 1;
    check_sample(k, [](size_t, int v) { CHECK_EQUAL(v, ((((13 % 5) << 2 | 1) ^ 3) & 14) >> 1); });
}

VEX_FUNCTION(double, squared_radius, (double, x)(double, y), return x * x + y * y;);
VEX_FUNCTION(double, times2, (double, x), return x * 2;);
VEX_FUNCTION(double, times4, (double, x), return x * 4;);
VEX_FUNCTION_D(double, chained, (double, x), (times2)(times4), return times2(x) + times4(x););

TEST_CASE(user_functions_and_ternary) {                              // :113-193, :238-250
    const size_t N = 1 << 16;
    vex::vector<double> x(ctx, random_vector<double>(N)), y(ctx, random_vector<double>(N)), r(ctx, N);
    std::vector<double> hx(N), hy(N); vex::copy(x, hx); vex::copy(y, hy);
    r = squared_radius(x, y);
    check_sample(r, [&](size_t i, double v) { CHECK_CLOSE(v, hx[i] * hx[i] + hy[i] * hy[i], 1e-12); });
    r = times2(x) + times4(y);                                      // same signature, distinct names (:131-145)
    check_sample(r, [&](size_t i, double v) { CHECK_CLOSE(v, 2 * hx[i] + 4 * hy[i], 1e-12); });
    r = chained(squared_radius(x, y));                              // nested (:147-193)
    check_sample(r, [&](size_t i, double v) { CHECK_CLOSE(v, 6 * (hx[i] * hx[i] + hy[i] * hy[i]), 1e-12); });
    r = vex::if_else(x < 0.5, y, -x);
    check_sample(r, [&](size_t i, double v) { CHECK_EQUAL(v, hx[i] < 0.5 ? hy[i] : -hx[i]); });
    vex::vector<size_t> idx(ctx, N);
    idx = vex::tag<1>(vex::element_index()) * 2 + vex::tag<1>(vex::element_index());   // :271-284, global indices
    check_sample(idx, [&](size_t i, size_t v) { CHECK_EQUAL(v, 3 * i); });
    vex::backend::push_program_header(ctx.queue(0), "#define THE_ANSWER 42\n");        // :195-212
    for (unsigned d = 1; d < ctx.size(); ++d) vex::backend::push_program_header(ctx.queue(d), "#define THE_ANSWER 42\n");
    VEX_FUNCTION(int, answer, (int, x), return x * THE_ANSWER;);
    vex::vector<int> a(ctx, N); a = answer(1);
    check_sample(a, [](size_t, int v) { CHECK_EQUAL(v, 42); });
    for (unsigned d = 0; d < ctx.size(); ++d) vex::backend::pop_program_header(ctx.queue(d));
}

TEST_CASE(reductions) {                                              // :66-99
    const size_t N = 1 << 20;
    std::vector<double> h = random_vector<double>(N);
    for (auto &v : h) v = (v - 0.5) * 1e8;
    vex::vector<double> x(ctx, h);
    vex::Reductor<double, vex::SUM> sum(ctx);
    vex::Reductor<double, vex::SUM_Kahan> ksum(ctx);
    vex::Reductor<double, vex::MIN> vmin(ctx);
    vex::Reductor<double, vex::MAX> vmax(ctx);
    vex::Reductor<double, vex::MIN_MAX> minmax(ctx);
    double exact = 0, c = 0, scale = 0;
    for (double v : h) { double y = v - c, t = exact + y; c = (t - exact) - y; exact = t; scale += std::fabs(v); }
    CHECK(std::fabs(sum(x) - exact) <= 1e-10 * scale);
    CHECK(std::fabs(ksum(x) - exact) <= 1e-12 * scale);
    CHECK_EQUAL(vmin(x), *std::min_element(h.begin(), h.end()));
    CHECK_EQUAL(vmax(x), *std::max_element(h.begin(), h.end()));
    auto mm = minmax(x);
    CHECK_EQUAL(mm.s[0], *std::min_element(h.begin(), h.end()));
    CHECK_EQUAL(mm.s[1], *std::max_element(h.begin(), h.end()));
    CHECK_EQUAL(vmax(fabs(x - x)), 0.0);
    double dot = 0; for (double v : h) dot += v * v;
    CHECK_CLOSE(sum(x * x), dot, 1e-8);
    vex::Reductor<size_t, vex::SUM> isum(ctx);                      // pure index expression (reductor.hpp:323-325)
    CHECK_EQUAL(isum(vex::element_index(0, 1000)), size_t(999 * 1000 / 2));
    vex::vector<int> k(ctx, 1000); k = 3;
    vex::Reductor<int, vex::SUM> ksumi(ctx);
    CHECK_EQUAL(ksumi(k * 2), 6000);
}

TEST_CASE(reductions_back_to_back_never_fold_a_stale_partial) {
    // The single-launch reduction (reductor.hpp:302-439 in one kernel): a workgroup's partial must be visible to the workgroup
    // that arrives last -- ordered by RELEASE arrivals / ACQUIRE in the closing workgroup (VEXCL_REDUCTOR_ORDER=release, the
    // default; `relaxed` and `two_launch` run the same test through tests/test_cpp_api.py).  Every reduction here changes EVERY
    // workgroup's partial (the multiplier changes), lengths alternate so the number of arrivals changes too, and each result
    // is compared exactly (integers): a fold that picked up one partial of the previous reduction gives a different number.
    const size_t N = (1 << 20) + 77;
    std::vector<long> h(N);
    for (size_t i = 0; i < N; ++i) h[i] = (long)((i * 2654435761ull) % 2003) - 1001;
    vex::vector<long> x(ctx, h);
    vex::Reductor<long, vex::SUM> sum(ctx);
    vex::Reductor<long, vex::MIN_MAX> minmax(ctx);
    const size_t lens[3] = {N, N / 3 + 5, 70001};
    long S[3] = {0, 0, 0}, lo[3], hi[3];
    for (int l = 0; l < 3; ++l) {
        lo[l] = h[0]; hi[l] = h[0];
        for (size_t i = 0; i < lens[l]; ++i) { S[l] += h[i]; lo[l] = std::min(lo[l], h[i]); hi[l] = std::max(hi[l], h[i]); }
    }
    const char *e = std::getenv("VEX_TEST_REDUCE_STRESS");
    const long rounds = e ? std::atol(e) : 20000;
    long wrong = 0;
    std::vector<std::unique_ptr<vex::vector<long>>> xs;
    for (int l = 0; l < 3; ++l) xs.emplace_back(new vex::vector<long>(ctx, std::vector<long>(h.begin(), h.begin() + lens[l])));
    for (long k = 1; k <= rounds; ++k) {
        const int l = (int)(k % 3);
        const long got = sum(*xs[l] * k + 1);
        if (got != S[l] * k + (long)lens[l]) ++wrong;
        if (k % 16 == 0) {
            auto mm = minmax(*xs[l] + k);
            if (mm.s[0] != lo[l] + k || mm.s[1] != hi[l] + k) ++wrong;
        }
    }
    CHECK_EQUAL(wrong, 0L);
}

TEST_CASE(custom_kernel) {                                           // custom_kernel.cpp:7-91
    const cl_ulong n = 1024;
    std::vector<vex::command_queue> queue(1, ctx.queue(0));
    vex::vector<int> x(queue, n);
    {
        vex::backend::source_generator src(queue[0]);
        src.begin_kernel("zeros");
        src.begin_kernel_parameters();
        src.parameter<size_t>("n");
        src.parameter<int*>("x");
        src.end_kernel_parameters();
        src.grid_stride_loop("idx", "n").open("{");
        src.new_line() << "x[idx] = 0;";
        src.close("}");
        src.end_kernel();
        vex::backend::kernel zeros(queue[0], src.str(), "zeros");
        zeros(queue[0], n, x(0));
        check_sample(x, [](size_t, int v) { CHECK_EQUAL(v, 0); });
    }
    {
        vex::backend::source_generator src(queue[0]);
        for (const char *name : {"ones", "twos"}) {
            src.begin_kernel(name);
            src.begin_kernel_parameters();
            src.parameter<size_t>("n");
            src.parameter<int*>("x");
            src.end_kernel_parameters();
            src.grid_stride_loop("idx", "n").open("{");
            src.new_line() << "x[idx] = " << (std::string(name) == "ones" ? 1 : 2) << ";";
            src.close("}");
            src.end_kernel();
        }
        auto program = vex::backend::build_sources(queue[0], src.str());
        vex::backend::kernel ones(queue[0], program, "ones"), twos(queue[0], program, "twos");
        ones(queue[0], n, x(0));
        check_sample(x, [](size_t, int v) { CHECK_EQUAL(v, 1); });
        twos.push_arg(n); twos.push_arg(x(0)); twos(queue[0]);
        check_sample(x, [](size_t, int v) { CHECK_EQUAL(v, 2); });
    }
}

TEST_CASE(events_and_queues) {                                       // events.cpp:9-104
    const size_t n = 1 << 20;
    std::vector<vex::command_queue> q1(1, ctx.queue(0));
    std::vector<vex::command_queue> q2(1, vex::backend::duplicate_queue(ctx.queue(0)));
    vex::vector<int> x(q1, n), y(q1, n);
    x = 1;
    auto e = vex::backend::enqueue_marker(q1[0]);
    vex::backend::enqueue_barrier(q2[0], vex::backend::wait_list(1, e));
    vex::vector<int> y2(q2[0], y(0));                                // same buffer, other queue
    y2 = x * 2;
    q2[0].finish();
    check_sample(y, [](size_t, int v) { CHECK_EQUAL(v, 2); });
    std::vector<int> h(n);
    vex::copy(q2, y, h);
    CHECK_EQUAL(h[n / 2], 2);
}

TEST_CASE(threads_one_queue_each) {                                  // threads.cpp:9-35
    const size_t n = 1 << 20;
    std::vector<std::thread> pool;
    std::vector<long> results(ctx.size(), 0);
    for (unsigned d = 0; d < ctx.size(); ++d)
        pool.emplace_back([&, d]() {
            std::vector<vex::command_queue> q(1, ctx.queue(d));
            vex::vector<int> x(q, n);
            x = int(d + 1);
            vex::Reductor<long, vex::SUM> sum(q);
            results[d] = sum(x);
        });
    for (auto &t : pool) t.join();
    for (unsigned d = 0; d < ctx.size(); ++d) CHECK_EQUAL(results[d], long(n * (d + 1)));
}

TEST_CASE(kernel_cache_is_per_type_not_per_value) {                  // values never recompile
    uint64_t c0 = 0, c1 = 0, h0 = 0, h1 = 0;
    vex::vector<double> x(ctx, 1000), y(ctx, 1000);
    y = 1;
    x = 2.0 * y + 0.25;
    vexhip_jit_stats(&c0, &h0);
    for (int i = 0; i < 10; ++i) x = double(i) * y + 0.5 * i;       // same type, other scalars
    vexhip_jit_stats(&c1, &h1);
    CHECK_EQUAL(c1 + h1, c0 + h0);
    check_sample(x, [](size_t, double v) { CHECK_EQUAL(v, 9.0 + 4.5); });
}
