// GPU: vector views -- gslice / slicer / range, slices as lvalues, permutation,
// reduce<RDC> along dimensions, reshape (reference: tests/vector_view.cpp).
#include "vex_test.hpp"
#include <numeric>
#include <valarray>

TEST_CASE(vector_view_1d) {                                           // vector_view.cpp:12-32
    const size_t N = 1024;
    std::vector<vex::command_queue> queue(1, ctx.queue(0));
    std::vector<double> x = random_vector<double>(2 * N);
    vex::vector<double> X(queue, x), Y(queue, N);
    size_t size = N, stride = 2;
    vex::gslice<1> slice(0, &size, &stride);
    Y = slice(X);
    check_sample(Y, [&](size_t i, double v) { CHECK(v == x[i * 2]); });
    Y = slice(X * X);                                                 // a slice of an expression
    check_sample(Y, [&](size_t i, double v) { CHECK(v == x[i * 2] * x[i * 2]); });
    Y = 2 * slice(X * X) + slice(X);                                  // two views in one kernel
    check_sample(Y, [&](size_t i, double v) { CHECK_CLOSE(v, 2 * x[i * 2] * x[i * 2] + x[i * 2], 1e-12); });
}

TEST_CASE(vector_view_2d_and_slicer) {                                // vector_view.cpp:34-109
    using vex::range; using vex::_;
    const size_t N = 32;
    std::vector<vex::command_queue> queue(1, ctx.queue(0));
    std::valarray<double> x(N * N);
    std::iota(&x[0], &x[N * N], 0);
    size_t start = 2 * N + 4, size[] = {5, 4}, stride[] = {2 * N, 2};    // every even point of block [(2,4) - (10,10)]
    std::gslice std_slice(start, std::valarray<size_t>(size, 2), std::valarray<size_t>(stride, 2));
    std::valarray<double> y = x[std_slice];
    vex::vector<double> X(queue, N * N, &x[0]), Y(queue, size[0] * size[1]), Z(queue, N);
    vex::gslice<2> vex_slice(start, size, stride);
    Y = vex_slice(X);
    check_sample(Y, [&](size_t i, double v) { CHECK_EQUAL(v, y[i]); });

    size_t dim[2] = {N, N};
    vex::slicer<2> slicer(dim);
    Y = 0;
    Y = slicer[range(2, 2, 11)][range(4, 2, 11)](X);
    check_sample(Y, [&](size_t i, double v) { CHECK_EQUAL(v, y[i]); });
    Z = slicer[5](X);                                                 // fifth row
    check_sample(Z, [&](size_t i, double v) { CHECK_EQUAL(v, x[5 * N + i]); });
    Z = slicer[_][5](X);                                              // fifth column
    check_sample(Z, [&](size_t i, double v) { CHECK_EQUAL(v, x[5 + N * i]); });
}

TEST_CASE(negative_stride) {                                          // vector_view.cpp:111-136
    std::vector<vex::command_queue> queue(1, ctx.queue(0));
    float v[] = {0, 5, 1, 4, 2, 3, 3, 2, 4, 1, 5, 0};
    const size_t rows = 6, cols = 2;
    vex::vector<float> x(queue, rows * cols, v), z(queue, rows / 2 * cols);
    vex::slicer<2> slice(vex::extents[rows][cols]);
    z = slice[vex::range(5, -2, 0)](x);
    for (size_t i = 0; i < rows / 2; ++i)
        for (size_t j = 0; j < cols; ++j)
            CHECK_EQUAL(static_cast<float>(z[i * cols + j]), v[(rows - i * 2 - 1) * cols + j]);
}

TEST_CASE(reduce_slice_and_assign_to_view) {                          // vector_view.cpp:187-247
    using vex::range; using vex::_;
    const size_t N = 1024;
    std::vector<vex::command_queue> queue(1, ctx.queue(0));
    vex::vector<int> X(queue, N);
    vex::Reductor<int, vex::SUM> sum(queue);
    vex::slicer<1> slice(&N);
    X = 1;
    CHECK_EQUAL(static_cast<int>(N / 2), sum(slice[range(0, 2, N)](X)));

    const size_t m = 32, n = m * m;
    vex::vector<int> x(queue, n);
    vex::slicer<1> slicer1(vex::extents[n]);
    vex::slicer<2> slicer2(vex::extents[m][m]);
    x = 1;
    slicer1[range(1, 2, n)](x) = 2;
    check_sample(x, [&](size_t i, int v) { CHECK_EQUAL(v, static_cast<int>(i % 2 + 1)); });
    for (size_t i = 0; i < m; ++i) slicer2[_][i](x) = i;
    check_sample(x, [&](size_t i, int v) { CHECK_EQUAL(v, static_cast<int>(i % m)); });
    slicer2[3](x) += 100;                                             // compound assignment to a row
    for (size_t i = 0; i < m; ++i) { CHECK_EQUAL(int(x[3 * m + i]), int(i) + 100); CHECK_EQUAL(int(x[4 * m + i]), int(i)); }

    vex::vector<size_t> I(queue, m);
    I = vex::element_index() * m;
    auto first_col = vex::permutation(I);
    first_col(x) = 42;
    for (size_t i = 0; i < m; ++i) CHECK_EQUAL(int(x[i * m]), 42);
}

TEST_CASE(slice_reductors) {                                          // vector_view.cpp:249-334
    std::vector<vex::command_queue> queue(1, ctx.queue(0));
    using vex::extents; using vex::_;
    {
        vex::vector<int> x(queue, 32), y(queue, 1);
        x = 1;
        y = vex::reduce<vex::SUM>(extents[32], x, 0);
        CHECK_EQUAL(int(y[0]), 32);
        x = 2;
        y = vex::reduce<vex::SUM>(extents[4][8], x, extents[0][1]);
        CHECK_EQUAL(int(y[0]), 64);
    }
    {
        vex::vector<int> x(queue, 32 * 32), y(queue, 32);
        vex::slicer<2> slice(extents[32][32]);
        int isum = 0, i2sum = 0;
        for (int i = 0; i < 32; ++i) { slice[i](x) = i; isum += i; i2sum += i * i; }
        y = vex::reduce<vex::SUM>(slice[_][_](x), 1);
        for (int i = 0; i < 32; ++i) CHECK_EQUAL(int(y[i]), i * 32);
        y = vex::reduce<vex::SUM>(slice[_][_](x), 0);
        for (size_t i = 0; i < 32; ++i) CHECK_EQUAL(int(y[i]), isum);
        auto t = vex::make_temp<1>(x);
        y = vex::reduce<vex::SUM>(slice[_][_], t * t, 1);
        for (int i = 0; i < 32; ++i) CHECK_EQUAL(int(y[i]), i * i * 32);
        y = vex::reduce<vex::SUM>(slice[_][_], t * t, 0);
        for (size_t i = 0; i < 32; ++i) CHECK_EQUAL(int(y[i]), i2sum);
    }
    {
        vex::vector<int> x(queue, 32 * 32 * 32), y(queue, 32);
        vex::slicer<3> slice(extents[32][32][32]);
        x = 1;
        auto test = [&](size_t d1, size_t d2) {
            std::array<size_t, 2> dim = {{d1, d2}};
            y = vex::reduce<vex::SUM>(slice[_][_][_](x), dim);
            check_sample(y, [&](size_t, int s) { CHECK_EQUAL(s, 1024); });
        };
        test(0, 1); test(1, 2); test(0, 2);
    }
}

TEST_CASE(nested_reduce_and_reshape) {                                // vector_view.cpp:336-393
    using vex::extents; using vex::_;
    const size_t n = 32;
    std::vector<vex::command_queue> queue(1, ctx.queue(0));
    std::vector<double> X = random_vector<double>(n * n * n);
    vex::vector<double> x(queue, X), y(queue, n);
    vex::slicer<2> s2(extents[n][n]);
    vex::slicer<3> s3(extents[n][n][n]);
    y = vex::reduce<vex::MAX>(s2[_], vex::reduce<vex::SUM>(s3[_], sin(x), 2), 1);
    check_sample(y, [&](size_t k, double Y) {
        double ms = -std::numeric_limits<double>::max();
        for (size_t j = 0, idx = k * n * n; j < n; ++j) {
            double sum = 0;
            for (size_t i = 0; i < n; ++i, ++idx) sum += sin(X[idx]);
            ms = std::max(ms, sum);
        }
        CHECK_CLOSE(ms, Y, 1e-8);
    });

    auto dim_out = vex::make_array<size_t>(4, 2);
    auto dim_in = vex::make_array<size_t>(1, 0);
    vex::vector<int> a(queue, 8);
    a = vex::element_index();
    vex::vector<int> b = vex::reshape(a, dim_out, dim_in);            // transpose of a 2 x 4 array
    check_sample(b, [&](size_t k, int v) {
        size_t i = k % dim_out[1], j = k / dim_out[1];
        CHECK_EQUAL(i, size_t(v) / dim_out[0]);
        CHECK_EQUAL(j, size_t(v) % dim_out[0]);
    });
}
