// GPU: vex::multivector / vex::tie behaviour the reference pins in
// tests/multivector_create.cpp, tests/multivector_arithmetics.cpp and the
// multivector cases of tests/spmv.cpp and tests/reduce -- on the 2-"device" context.
#include "vex_test.hpp"
#include <array>
#include <numeric>

VEX_FUNCTION(double, mv_sqr_sum, (double, x)(double, y), return x * x + y * y;);

TEST_CASE(multivector_create) {                                       // multivector_create.cpp:7-110
    const size_t n = 1024, m = 3;
    typedef std::array<double, m> elem_t;

    vex::multivector<double, m> empty;
    CHECK_EQUAL(empty.size(), size_t(0));

    vex::multivector<double, m> x(ctx, n);
    CHECK_EQUAL(x.size(), n);
    CHECK_EQUAL(x(0).nparts(), ctx.size());
    CHECK_EQUAL(size_t(x.end() - x.begin()), n);

    std::vector<double> h = random_vector<double>(n * m);
    vex::multivector<double, m> y(ctx, h);                            // components one after another
    check_sample(y, [&](size_t i, elem_t a) { for (size_t k = 0; k < m; ++k) CHECK_EQUAL(a[k], h[k * n + i]); });

    vex::multivector<double, m> c(y);                                 // deep copy
    CHECK(c(0)(0).raw() != y(0)(0).raw());
    y = 0;
    check_sample(c, [&](size_t i, elem_t a) { for (size_t k = 0; k < m; ++k) CHECK_EQUAL(a[k], h[k * n + i]); });
    check_sample(y, [&](size_t, elem_t a) { for (size_t k = 0; k < m; ++k) CHECK_EQUAL(a[k], 0.0); });

    double *before = c(1)(0).raw();
    vex::multivector<double, m> mv(std::move(c));                     // move keeps the buffers
    CHECK(mv(1)(0).raw() == before);

    std::vector<double> back(n * m);
    vex::copy(mv, back);
    CHECK(back == h);
    vex::copy(back, y);
    check_sample(y, [&](size_t i, elem_t a) { for (size_t k = 0; k < m; ++k) CHECK_EQUAL(a[k], h[k * n + i]); });

    elem_t e = {{1.5, 2.5, 3.5}};                                     // element proxy write
    y[17] = e;
    elem_t r = y[17];
    CHECK(r == e);

    x.resize(ctx, 2 * n);
    CHECK_EQUAL(x.size(), 2 * n);

    vex::multivector<double, m> fromexpr = 2 * mv;                    // expression constructor
    CHECK_EQUAL(fromexpr.size(), n);
    check_sample(fromexpr, [&](size_t i, elem_t a) { for (size_t k = 0; k < m; ++k) CHECK_EQUAL(a[k], 2 * h[k * n + i]); });
}

TEST_CASE(multivector_arithmetics) {                                  // multivector_arithmetics.cpp:10-57
    typedef std::array<double, 4> elem_t;
    const size_t n = 1024;
    vex::multivector<double, 4> x(ctx, n), y(ctx, random_vector<double>(n * 4)), z(ctx, random_vector<double>(n * 4));

    std::array<int, 4> v = {{6, 7, 8, 9}};
    x = v;
    check_sample(x, [&](size_t, elem_t a) { for (size_t k = 0; k < 4; ++k) CHECK_EQUAL(a[k], v[k]); });

    vex::multivector<double, 4> w = x + y;
    x = 2 * std::cos(0.0) * y - z;      // host scalar folded before it reaches the expression
    check_sample(x, y, [&](size_t idx, elem_t a, elem_t b) {
        elem_t c = z[idx];
        for (size_t k = 0; k < 4; ++k) CHECK_CLOSE(a[k], 2 * b[k] - c[k], 1e-10);
    });

    x = std::make_tuple(1, 2, 3, 4) * y + z;
    check_sample(x, y, [&](size_t idx, elem_t a, elem_t b) {
        elem_t c = z[idx];
        for (size_t k = 0; k < 4; ++k) CHECK_CLOSE(a[k], (k + 1) * b[k] + c[k], 1e-10);
    });

    x = sin(y) * cos(z) + pow(y, 2.0);
    check_sample(x, y, [&](size_t idx, elem_t a, elem_t b) {
        elem_t c = z[idx];
        for (size_t k = 0; k < 4; ++k) CHECK_CLOSE(a[k], sin(b[k]) * cos(c[k]) + b[k] * b[k], 1e-8);
    });

    x = mv_sqr_sum(y, z);                                             // user function on multivectors
    check_sample(x, y, [&](size_t idx, elem_t a, elem_t b) {
        elem_t c = z[idx];
        for (size_t k = 0; k < 4; ++k) CHECK_CLOSE(a[k], b[k] * b[k] + c[k] * c[k], 1e-10);
    });

    x = if_else(y > 0.5, y, -z);                                      // ternary over components
    check_sample(x, y, [&](size_t idx, elem_t a, elem_t b) {
        elem_t c = z[idx];
        for (size_t k = 0; k < 4; ++k) CHECK_EQUAL(a[k], b[k] > 0.5 ? b[k] : -c[k]);
    });
}

TEST_CASE(multiexpressions) {                                         // multivector_arithmetics.cpp:59-94
    typedef std::array<double, 2> elem_t;
    const size_t n = 1024;
    vex::multivector<double, 2> x(ctx, n), y(ctx, random_vector<double>(n * 2));

    x = std::tie(sin(y(0)) + cos(y(1)), cos(y(0)) + sin(y(1)));
    check_sample(x, y, [&](size_t, elem_t a, elem_t b) {
        CHECK_CLOSE(a[0], sin(b[0]) + cos(b[1]), 1e-8);
        CHECK_CLOSE(a[1], cos(b[0]) + sin(b[1]), 1e-8);
    });

    // rotation: both components read both sources; all reads precede the stores
    std::vector<double> before(2 * n);
    vex::copy(y, before);
    const double alpha = 0.3;
    y = std::tie(y(0) * cos(alpha) - y(1) * sin(alpha), y(0) * sin(alpha) + y(1) * cos(alpha));
    check_sample(y, [&](size_t i, elem_t a) {
        CHECK_CLOSE(a[0], before[i] * cos(alpha) - before[n + i] * sin(alpha), 1e-8);
        CHECK_CLOSE(a[1], before[i] * sin(alpha) + before[n + i] * cos(alpha), 1e-8);
    });
}

TEST_CASE(tied_vectors) {                                             // multivector_arithmetics.cpp:96-120
    const size_t n = 1024;
    std::vector<double> hx = random_vector<double>(n), hy = random_vector<double>(n);
    vex::vector<double> X(ctx, hx), Y(ctx, hy), A(ctx, n), B(ctx, n);

    vex::tie(A, B) = std::tie(X + Y, X - Y);
    check_sample(A, B, [&](size_t i, double a, double b) {
        CHECK_CLOSE(a, hx[i] + hy[i], 1e-12);
        CHECK_CLOSE(b, hx[i] - hy[i], 1e-12);
    });

    vex::tie(X, Y) = std::tie(Y, X);                                  // swap: loads before stores
    check_sample(X, Y, [&](size_t i, double a, double b) { CHECK_EQUAL(a, hy[i]); CHECK_EQUAL(b, hx[i]); });

    vex::vector<int> I(ctx, n);                                       // heterogeneous element types
    vex::tie(A, I) = std::tie(2 * X, vex::element_index());
    check_sample(A, I, [&](size_t i, double a, int k) { CHECK_EQUAL(a, 2 * hy[i]); CHECK_EQUAL(size_t(k), i); });

    vex::tie(A, B) += std::make_tuple(1, 2);
    check_sample(A, B, [&](size_t i, double a, double b) {
        CHECK_CLOSE(a, 2 * hy[i] + 1, 1e-12);
        CHECK_CLOSE(b, hx[i] - hy[i] + 2, 1e-12);
    });
}

TEST_CASE(multivector_reduction) {                                    // multivector_arithmetics.cpp:122-149
    const size_t n = 1 << 16;
    std::vector<double> h = random_vector<double>(2 * n);
    vex::multivector<double, 2> x(ctx, h);
    vex::Reductor<double, vex::SUM> sum(ctx);
    vex::Reductor<double, vex::MIN> mn(ctx);
    vex::Reductor<double, vex::MAX> mx(ctx);

    std::array<double, 2> s = sum(x), lo = mn(x), hi = mx(x);
    for (size_t k = 0; k < 2; ++k) {
        CHECK_CLOSE(s[k], std::accumulate(h.begin() + k * n, h.begin() + (k + 1) * n, 0.0), 1e-6);
        CHECK_EQUAL(lo[k], *std::min_element(h.begin() + k * n, h.begin() + (k + 1) * n));
        CHECK_EQUAL(hi[k], *std::max_element(h.begin() + k * n, h.begin() + (k + 1) * n));
    }
    std::array<double, 2> s2 = sum(x * x + 1);
    for (size_t k = 0; k < 2; ++k) {
        double ref = 0; for (size_t i = 0; i < n; ++i) ref += h[k * n + i] * h[k * n + i] + 1;
        CHECK_CLOSE(s2[k], ref, 1e-6);
    }
}

TEST_CASE(multivector_element_index_and_compound) {                   // multivector_arithmetics.cpp:151-215
    typedef std::array<double, 2> elem_t;
    const size_t n = 1024;
    vex::multivector<double, 2> x(ctx, n), y(ctx, random_vector<double>(n * 2));

    x = 0.5 * vex::element_index();
    check_sample(x, [&](size_t i, elem_t a) { CHECK_EQUAL(a[0], 0.5 * i); CHECK_EQUAL(a[1], 0.5 * i); });

    x = std::tie(sin(0.5 * vex::element_index()), cos(0.5 * vex::element_index()));
    check_sample(x, [&](size_t i, elem_t a) { CHECK_CLOSE(a[0], sin(0.5 * i), 1e-6); CHECK_CLOSE(a[1], cos(0.5 * i), 1e-6); });

    x = 0;
    x += sin(2 * y);
    check_sample(x, y, [&](size_t, elem_t a, elem_t b) { for (size_t k = 0; k < 2; ++k) CHECK_CLOSE(a[k], sin(2 * b[k]), 1e-8); });
    x = 0;
    x -= sin(2 * y);
    check_sample(x, y, [&](size_t, elem_t a, elem_t b) { for (size_t k = 0; k < 2; ++k) CHECK_CLOSE(a[k], -sin(2 * b[k]), 1e-8); });
    x = 1;
    x *= std::tie(y(1), sin(y(0)));
    check_sample(x, y, [&](size_t, elem_t a, elem_t b) { CHECK_CLOSE(a[0], b[1], 1e-8); CHECK_CLOSE(a[1], sin(b[0]), 1e-8); });

    x = std::integral_constant<int, 42>();                            // multivector_arithmetics.cpp:217-238
    check_sample(x, [&](size_t, elem_t a) { CHECK_EQUAL(a[0], 42.0); CHECK_EQUAL(a[1], 42.0); });
    x = sin(vex::constants::e() * vex::element_index());
    check_sample(x, [&](size_t i, elem_t a) { for (size_t k = 0; k < 2; ++k) CHECK_CLOSE(a[k], sin(std::exp(1.0) * i), 1e-8); });

    vex::multivector<double, 2> small(ctx, 16);                       // multivector_arithmetics.cpp:240-249
    bool thrown = false;
    try { small = y; } catch (const std::runtime_error &) { thrown = true; }
    CHECK(thrown);
}

TEST_CASE(multivector_spmv) {                                         // spmv.cpp:262-305
    typedef std::array<double, 2> elem_t;
    const size_t n = 1024;
    std::vector<size_t> row, col; std::vector<double> val;
    random_matrix(n, n, 16, row, col, val);
    std::vector<double> x = random_vector<double>(n * 2);

    vex::SpMat<double> A(ctx, n, n, row.data(), col.data(), val.data());
    vex::multivector<double, 2> X(ctx, x), Y(ctx, n);

    auto rowsum = [&](size_t i, size_t k) {
        double s = 0;
        for (size_t j = row[i]; j < row[i + 1]; ++j) s += val[j] * x[k * n + col[j]];
        return s;
    };

    Y = A * X;
    check_sample(Y, [&](size_t i, elem_t a) { for (size_t k = 0; k < 2; ++k) CHECK_CLOSE(a[k], rowsum(i, k), 1e-8); });
    Y = X + A * X;
    check_sample(Y, [&](size_t i, elem_t a) { for (size_t k = 0; k < 2; ++k) CHECK_CLOSE(a[k], x[k * n + i] + rowsum(i, k), 1e-8); });
    Y -= 2 * (A * X);
    check_sample(Y, [&](size_t i, elem_t a) { for (size_t k = 0; k < 2; ++k) CHECK_CLOSE(a[k] + 1, 1 + x[k * n + i] - rowsum(i, k), 1e-8); });
}

TEST_CASE(multivector_inline_spmv) {                                  // spmv.cpp:307-345
    typedef std::array<double, 2> elem_t;
    const size_t n = 1024;
    std::vector<vex::backend::command_queue> queue(1, ctx.queue(0));
    std::vector<size_t> row, col; std::vector<double> val;
    random_matrix(n, n, 16, row, col, val);
    std::vector<double> x = random_vector<double>(n * 2);
    vex::SpMat<double> A(queue, n, n, row.data(), col.data(), val.data());
    vex::multivector<double, 2> X(queue, x), Y(queue, n);

    Y = cos(vex::make_inline(A * X));
    check_sample(Y, [&](size_t i, elem_t a) {
        for (size_t k = 0; k < 2; ++k) {
            double s = 0;
            for (size_t j = row[i]; j < row[i + 1]; ++j) s += val[j] * x[k * n + col[j]];
            CHECK_CLOSE(a[k], cos(s), 1e-8);
        }
    });
}

TEST_CASE(multivector_spmv_single_device_fused) {                     // one pass over the matrix for all components
    typedef std::array<double, 3> elem_t;
    std::vector<vex::backend::command_queue> queue(1, ctx.queue(0));
    // (a) 3-D Poisson (1-byte diagonal codes), (b) random rows with a CSR tail (32-bit columns)
    for (int kind = 0; kind < 2; ++kind) {
        std::vector<int> row, col; std::vector<double> val;
        size_t n;
        if (kind == 0) {
            const int m = 20; n = size_t(m) * m * m;
            row.push_back(0);
            for (int k = 0; k < m; ++k) for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) {
                int idx = (k * m + j) * m + i;
                if (i == 0 || i == m - 1 || j == 0 || j == m - 1 || k == 0 || k == m - 1) { col.push_back(idx); val.push_back(1); }
                else {
                    int nb[7] = {idx - m * m, idx - m, idx - 1, idx, idx + 1, idx + m, idx + m * m};
                    for (int q = 0; q < 7; ++q) { col.push_back(nb[q]); val.push_back(q == 3 ? 6.0 : -1.0 - 0.01 * q); }
                }
                row.push_back((int)col.size());
            }
        } else {
            n = 3000;
            random_matrix(n, n, 16, row, col, val);
        }
        std::vector<double> x = random_vector<double>(n * 3), y0 = random_vector<double>(n * 3);
        vex::SpMat<double, int, int> A(queue, n, n, row.data(), col.data(), val.data());
        vex::multivector<double, 3> X(queue, x), Y(queue, y0), Z(queue, n);
        auto rowsum = [&](size_t i, size_t k) {
            double s = 0;
            for (int j = row[i]; j < row[i + 1]; ++j) s += val[j] * x[k * n + col[j]];
            return s;
        };
        Z = A * X;
        for (size_t k = 0; k < 3; ++k) {                               // bit-identical to the single-vector product
            vex::vector<double> z(queue, n);
            z = A * X(k);
            std::vector<double> a(n), b(n);
            vex::copy(z, a); vex::copy(Z(k), b);
            CHECK(a == b);
        }
        check_sample(Z, [&](size_t i, elem_t a) { for (size_t k = 0; k < 3; ++k) CHECK_CLOSE(a[k], rowsum(i, k), 1e-8); });
        Y += 2 * (A * X) - X;
        check_sample(Y, [&](size_t i, elem_t a) {
            for (size_t k = 0; k < 3; ++k) CHECK_CLOSE(a[k] + 100, 100 + y0[k * n + i] + 2 * rowsum(i, k) - x[k * n + i], 1e-8);
        });
    }
}
