// GPU: n-D views of vectors and expressions -- vex::gslice, vex::slicer with range / _,
// slices as lvalues, vex::permutation as an lvalue, vex::reduce<RDC> along dimensions,
// vex::reshape.  Behaviour pinned by the reference's tests/vector_view.cpp (cases named
// below); here every view is compared with a host model of the same index arithmetic.
#include "vex_test.hpp"
#include <array>
#include <numeric>

namespace {

// host model of a generalized slice: element k of the view, row-major over len[], last dimension fastest
template <size_t D>
size_t slice_position(size_t k, size_t start, const std::array<size_t, D> &len, const std::array<long, D> &stride) {
    long pos = (long)start;
    for (size_t d = D; d-- > 0;) { pos += (long)(k % len[d]) * stride[d]; k /= len[d]; }
    return (size_t)pos;
}

std::vector<vex::command_queue> one_queue() { return std::vector<vex::command_queue>(1, ctx.queue(0)); }

template <class T> std::vector<T> download(const vex::vector<T> &v) { std::vector<T> h(v.size()); vex::copy(v, h); return h; }

} // namespace

TEST_CASE(strided_slices_of_vectors_and_expressions) {                // vector_view.cpp: vector_view_1d, vector_view_2
    auto q = one_queue();
    const size_t n = 3000;
    std::vector<double> h = random_vector<double>(n);
    vex::vector<double> X(q, h);

    // 1-D: every third element from position 7
    std::array<size_t, 1> len1 = {{900}}; std::array<long, 1> str1 = {{3}};
    vex::gslice<1> s1(7, len1, str1);
    vex::vector<double> Y(q, len1[0]);
    Y = s1(X);
    auto got = download(Y);
    for (size_t k = 0; k < got.size(); ++k) CHECK(got[k] == h[slice_position(k, 7, len1, str1)]);
    Y = s1(X * X + 1);                                                // the EXPRESSION is evaluated at the mapped position
    got = download(Y);
    for (size_t k = 0; k < got.size(); ++k) { double v = h[7 + 3 * k]; CHECK_CLOSE(got[k], v * v + 1, 1e-12); }   // the device fuses the multiply-add
    Y = s1(X) - 2 * s1(sqrt(X));                                      // two views, one kernel
    got = download(Y);
    for (size_t k = 0; k < got.size(); k += 7) { double v = h[7 + 3 * k]; CHECK_CLOSE(got[k] + 10, v - 2 * std::sqrt(v) + 10, 1e-12); }

    // 2-D: a 6 x 9 block of a 40 x 75 array, rows 2 apart, columns 5 apart, from element (3, 11)
    const size_t rows = 40, cols = 75;
    std::array<size_t, 2> len2 = {{6, 9}}; std::array<long, 2> str2 = {{2 * (long)cols, 5}};
    vex::gslice<2> s2(3 * cols + 11, len2, str2);
    vex::vector<double> B(q, 54);
    B = s2(X);
    got = download(B);
    for (size_t k = 0; k < 54; ++k) CHECK(got[k] == h[slice_position(k, 3 * cols + 11, len2, str2)]);
    (void)rows;
}

TEST_CASE(slicer_ranges_rows_columns_and_negative_strides) {          // vector_view.cpp: vector_slicer_2d, negative_stride
    using vex::range; using vex::_;
    auto q = one_queue();
    const size_t R = 24, C = 36;
    std::vector<int> h(R * C);
    std::iota(h.begin(), h.end(), 0);
    vex::vector<int> X(q, h);
    vex::slicer<2> at(vex::extents[R][C]);

    vex::vector<int> row(q, C), colv(q, R);
    row = at[9](X);
    auto got = download(row);
    for (size_t j = 0; j < C; ++j) CHECK_EQUAL(got[j], h[9 * C + j]);
    colv = at[_][17](X);
    got = download(colv);
    for (size_t i = 0; i < R; ++i) CHECK_EQUAL(got[i], h[i * C + 17]);

    // rows 4, 7, 10, ..., 22 and columns 30, 26, ..., 2 (a negative stride walks backwards)
    vex::vector<int> blk(q, 7 * 8);
    blk = at[range(4, 3, 23)][range(30, -4, 0)](X);
    got = download(blk);
    for (size_t a = 0; a < 7; ++a) for (size_t b = 0; b < 8; ++b) CHECK_EQUAL(got[a * 8 + b], h[(4 + 3 * a) * C + (30 - 4 * b)]);

    // the reference's case: every second row of a 6 x 2 array, from the last one upwards
    float v[] = {0, 5, 1, 4, 2, 3, 3, 2, 4, 1, 5, 0};
    vex::vector<float> x(q, 12, v), z(q, 6);
    vex::slicer<2> s62(vex::extents[6][2]);
    z = s62[range(5, -2, 0)](x);
    for (size_t i = 0; i < 3; ++i) for (size_t j = 0; j < 2; ++j) CHECK_EQUAL(float(z[i * 2 + j]), v[(5 - 2 * i) * 2 + j]);
}

TEST_CASE(views_as_left_hand_sides) {                                 // vector_view.cpp: reduce_slice, assign_to_view
    using vex::range; using vex::_;
    auto q = one_queue();
    const size_t m = 20, n = m * m;
    vex::vector<int> x(q, n);
    vex::slicer<1> flat(vex::extents[n]);
    vex::slicer<2> grid(vex::extents[m][m]);

    x = 1;
    flat[range(1, 2, n)](x) = 2;                                      // odd positions
    auto got = download(x);
    for (size_t i = 0; i < n; ++i) CHECK_EQUAL(got[i], int(i % 2 + 1));
    vex::Reductor<int, vex::SUM> sum(q);
    CHECK_EQUAL(sum(flat[range(0, 2, n)](x)), int(n / 2));            // a view inside a reduction

    for (size_t j = 0; j < m; ++j) grid[_][j](x) = 10 * j;            // whole columns
    got = download(x);
    for (size_t i = 0; i < n; ++i) CHECK_EQUAL(got[i], int(10 * (i % m)));
    grid[6](x) += 7;                                                  // compound assignment to one row
    grid[range(0, 5)][range(0, 5)](x) *= -1;                          // ... and to a block
    got = download(x);
    for (size_t i = 0; i < m; ++i) for (size_t j = 0; j < m; ++j) {
        int want = int(10 * j) + (i == 6 ? 7 : 0);
        if (i < 5 && j < 5) want = -want;
        CHECK_EQUAL(got[i * m + j], want);
    }

    vex::vector<size_t> diag(q, m);                                   // permutation view as an lvalue
    diag = vex::element_index() * (m + 1);
    vex::permutation(diag)(x) = 42;
    got = download(x);
    for (size_t i = 0; i < m; ++i) CHECK_EQUAL(got[i * m + i], 42);
}

TEST_CASE(reductions_along_dimensions) {                              // vector_view.cpp: slice_reductor_*
    using vex::extents; using vex::_;
    auto q = one_queue();
    {   // everything to one number
        vex::vector<int> x(q, 60), y(q, 1);
        x = 3;
        y = vex::reduce<vex::SUM>(extents[60], x, 0);
        CHECK_EQUAL(int(y[0]), 180);
        y = vex::reduce<vex::SUM>(extents[5][12], x, extents[0][1]);
        CHECK_EQUAL(int(y[0]), 180);
    }
    const size_t A = 6, B = 10, Cd = 14;
    std::vector<int> h(A * B * Cd);
    for (size_t i = 0; i < h.size(); ++i) h[i] = int((i * 7919) % 23) - 11;
    vex::vector<int> x(q, h);
    vex::slicer<3> cube(extents[A][B][Cd]);
    auto at = [&](size_t a, size_t b, size_t c) { return h[(a * B + b) * Cd + c]; };
    {   // one dimension at a time: the same expression type, three different dimensions (one cached kernel)
        vex::vector<int> y0(q, B * Cd), y1(q, A * Cd), y2(q, A * B);
        y0 = vex::reduce<vex::SUM>(cube[_][_][_](x), 0);
        y1 = vex::reduce<vex::SUM>(cube[_][_][_](x), 1);
        y2 = vex::reduce<vex::SUM>(cube[_][_][_](x), 2);
        auto g0 = download(y0), g1 = download(y1), g2 = download(y2);
        for (size_t b = 0; b < B; ++b) for (size_t c = 0; c < Cd; ++c) { int s = 0; for (size_t a = 0; a < A; ++a) s += at(a, b, c); CHECK_EQUAL(g0[b * Cd + c], s); }
        for (size_t a = 0; a < A; ++a) for (size_t c = 0; c < Cd; ++c) { int s = 0; for (size_t b = 0; b < B; ++b) s += at(a, b, c); CHECK_EQUAL(g1[a * Cd + c], s); }
        for (size_t a = 0; a < A; ++a) for (size_t b = 0; b < B; ++b) { int s = 0; for (size_t c = 0; c < Cd; ++c) s += at(a, b, c); CHECK_EQUAL(g2[a * B + b], s); }
    }
    {   // two dimensions at once, of an expression with a temporary; MIN / MAX
        auto t = vex::make_temp<1>(x);
        vex::vector<int> y(q, B);
        std::array<size_t, 2> dims = {{0, 2}};
        y = vex::reduce<vex::SUM>(cube[_][_][_], t * t, dims);
        auto g = download(y);
        for (size_t b = 0; b < B; ++b) { int s = 0; for (size_t a = 0; a < A; ++a) for (size_t c = 0; c < Cd; ++c) s += at(a, b, c) * at(a, b, c); CHECK_EQUAL(g[b], s); }
        y = vex::reduce<vex::MAX>(cube[_][_][_], x, dims);
        g = download(y);
        for (size_t b = 0; b < B; ++b) { int s = -100; for (size_t a = 0; a < A; ++a) for (size_t c = 0; c < Cd; ++c) s = std::max(s, at(a, b, c)); CHECK_EQUAL(g[b], s); }
        y = vex::reduce<vex::MIN>(cube[_][_][_], x, dims);
        g = download(y);
        for (size_t b = 0; b < B; ++b) { int s = 100; for (size_t a = 0; a < A; ++a) for (size_t c = 0; c < Cd; ++c) s = std::min(s, at(a, b, c)); CHECK_EQUAL(g[b], s); }
    }
}

TEST_CASE(nested_reductions_and_reshape) {                            // vector_view.cpp: nested_reduce, reshape
    using vex::extents; using vex::_;
    auto q = one_queue();
    const size_t n = 18;
    std::vector<double> h = random_vector<double>(n * n * n);
    vex::vector<double> x(q, h), y(q, n);
    vex::slicer<2> s2(extents[n][n]);
    vex::slicer<3> s3(extents[n][n][n]);
    y = vex::reduce<vex::MAX>(s2[_], vex::reduce<vex::SUM>(s3[_], cos(x), 2), 1);     // max over j of sum over i
    auto got = download(y);
    for (size_t k = 0; k < n; ++k) {
        double best = -1e300;
        for (size_t j = 0; j < n; ++j) { double s = 0; for (size_t i = 0; i < n; ++i) s += cos(h[(k * n + j) * n + i]); best = std::max(best, s); }
        CHECK_CLOSE(got[k], best, 1e-8);
    }

    // reshape = permutation of dimensions: a 3 x 5 x 2 array read as 2 x 3 x 5.  src_dims[k] names the RESULT dimension
    // that source dimension k becomes (the source is shaped dst_dims[src_dims], vector_view.hpp:1092-1097 of the reference)
    const size_t d0 = 3, d1 = 5, d2 = 2;
    vex::vector<int> a(q, d0 * d1 * d2);
    a = vex::element_index();
    vex::vector<int> b = vex::reshape(a, vex::make_array<size_t>(d2, d0, d1), vex::make_array<size_t>(1, 2, 0));
    auto gb = download(b);
    for (size_t c = 0; c < d2; ++c) for (size_t i = 0; i < d0; ++i) for (size_t j = 0; j < d1; ++j)
        CHECK_EQUAL(gb[(c * d0 + i) * d1 + j], int((i * d1 + j) * d2 + c));
    vex::vector<int> t = vex::reshape(a, vex::extents[d1 * d2][d0], vex::extents[1][0]);   // plain 2-D transpose, extents form
    auto gt = download(t);
    for (size_t r = 0; r < d1 * d2; ++r) for (size_t c = 0; c < d0; ++c) CHECK_EQUAL(gt[r * d0 + c], int(c * d1 * d2 + r));
}
