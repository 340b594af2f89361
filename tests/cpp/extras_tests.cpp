// GPU: the components around the hot path that the reference's remaining test sources drive --
// vex::tensordot (tests/tensordot.cpp), vex::multi_array (tests/multi_array.cpp), vex::mba (tests/mba.cpp),
// vex::svm_vector (tests/svm.cpp), the symbolic kernel generator (tests/generator.cpp), vex::FFT (tests/fft.cpp),
// vex::reshape with broadcast and 3-cycles.  Every result is compared with a host model written here.
#include "vex_test.hpp"
#include <vexcl/tensordot.hpp>
#include <vexcl/multi_array.hpp>
#include <vexcl/mba.hpp>
#include <vexcl/svm_vector.hpp>
#include <vexcl/generator.hpp>
#include <vexcl/fft.hpp>
#include <vexcl/constant_address_space.hpp>
#include <complex>

namespace {
std::vector<vex::command_queue> one_queue() { return std::vector<vex::command_queue>(1, ctx.queue(0)); }
template <class T> std::vector<T> download(const vex::vector<T> &v) { std::vector<T> h(v.size()); vex::copy(v, h); return h; }
}

TEST_CASE(tensordot_matrix_products_and_double_contraction) {         // tensordot.cpp: mat_mat, mat_vec, vec_mat
    using vex::_; using vex::extents;
    auto q = one_queue();
    const size_t N = 24, M = 17, K = 9;
    std::vector<double> a = random_vector<double>(N * K), b = random_vector<double>(K * M);
    vex::vector<double> A(q, a), B(q, b), C(q, N * M);
    vex::slicer<2> sa(extents[N][K]), sb(extents[K][M]);
    C = vex::tensordot(sa[_](A), sb[_](B), vex::axes_pairs(1, 0));    // C = A B
    auto c = download(C);
    for (size_t i = 0; i < N; ++i) for (size_t j = 0; j < M; ++j) {
        double s = 0; for (size_t k = 0; k < K; ++k) s += a[i * K + k] * b[k * M + j];
        CHECK_CLOSE(c[i * M + j] + 100, s + 100, 1e-10);
    }
    // contraction over the FIRST axes: C2 = A^T A2, operands are expressions
    std::vector<double> a2 = random_vector<double>(N * M);
    vex::vector<double> A2(q, a2), C2(q, K * M);
    vex::slicer<2> sa2(extents[N][M]);
    C2 = 2 * vex::tensordot(sa[_](A * A), sa2[_](A2 + 1), vex::axes_pairs(0, 0));
    auto c2 = download(C2);
    for (size_t i = 0; i < K; ++i) for (size_t j = 0; j < M; ++j) {
        double s = 0; for (size_t k = 0; k < N; ++k) s += a[k * K + i] * a[k * K + i] * (a2[k * M + j] + 1);
        CHECK_CLOSE(c2[i * M + j] + 100, 2 * s + 100, 1e-10);
    }
    // both axes contracted: a scalar (one output element), Frobenius product with a transposed operand
    vex::vector<double> S(q, 1);
    vex::slicer<2> st(extents[K][N]);
    std::vector<double> t = random_vector<double>(K * N);
    vex::vector<double> Tt(q, t);
    S = vex::tensordot(sa[_](A), st[_](Tt), vex::axes_pairs(0, 1, 1, 0));
    double s = 0; for (size_t i = 0; i < N; ++i) for (size_t k = 0; k < K; ++k) s += a[i * K + k] * t[k * N + i];
    CHECK_CLOSE(download(S)[0] + 100, s + 100, 1e-10);
    // matrix * vector through a 1-D slicer
    std::vector<double> x = random_vector<double>(K);
    vex::vector<double> X(q, x), Y(q, N);
    vex::slicer<1> sx(extents[K]);
    Y = vex::tensordot(sa[_](A), sx[_](X), vex::axes_pairs(1, 0));
    auto y = download(Y);
    for (size_t i = 0; i < N; ++i) { double r = 0; for (size_t k = 0; k < K; ++k) r += a[i * K + k] * x[k]; CHECK_CLOSE(y[i] + 100, r + 100, 1e-10); }
}

TEST_CASE(reshape_broadcast_and_three_cycle) {                        // vector_view.hpp:1003-1124 semantics
    using vex::extents;
    auto q = one_queue();
    const size_t d0 = 3, d1 = 4, d2 = 5;
    // source shaped dst_dims[src_dims] = [d2][d0][d1], row-major; src_dims = (2, 0, 1)
    std::vector<int> h(d0 * d1 * d2); std::iota(h.begin(), h.end(), 0);
    vex::vector<int> X(q, h), Y(q, d0 * d1 * d2);
    Y = vex::reshape(X, extents[d0][d1][d2], extents[2][0][1]);
    auto y = download(Y);
    for (size_t i = 0; i < d0; ++i) for (size_t j = 0; j < d1; ++j) for (size_t k = 0; k < d2; ++k)
        CHECK_EQUAL(y[(i * d1 + j) * d2 + k], h[(k * d0 + i) * d1 + j]);
    // fewer source dimensions: the vector is repeated along the dimensions it does not name
    std::vector<int> v(d1); std::iota(v.begin(), v.end(), 100);
    vex::vector<int> V(q, v), Z(q, d0 * d1);
    Z = vex::reshape(V, extents[d0][d1], extents[1]);
    auto z = download(Z);
    for (size_t i = 0; i < d0; ++i) for (size_t j = 0; j < d1; ++j) CHECK_EQUAL(z[i * d1 + j], v[j]);
}

TEST_CASE(multi_array_views_and_reduction) {                          // multi_array.cpp: create, slicing, reducing
    using vex::extents; using vex::indices; using vex::range; using vex::_;
    auto q = one_queue();
    vex::multi_array<double, 3> x(q, extents[6][7][8]), y(q, extents[6][7][8]);
    CHECK_EQUAL(x.size<0>(), 6u); CHECK_EQUAL(x.size<2>(), 8u);
    auto view = x(indices[2][range(1, 5)][_]);
    CHECK_EQUAL(view.size<0>(), 4u); CHECK_EQUAL(view.size<1>(), 8u);
    x.vec() = vex::element_index();
    y.vec() = 0;
    for (size_t j = 0; j < 7; ++j) y(indices[_][j][_]).vec() = 10 * x(indices[_][6 - j][_]).vec();   // slice = slice copies elements
    auto hy = download(y.vec());
    for (size_t i = 0; i < 6; ++i) for (size_t j = 0; j < 7; ++j) for (size_t k = 0; k < 8; ++k)
        CHECK_EQUAL(hy[(i * 7 + j) * 8 + k], 10.0 * ((i * 7 + (6 - j)) * 8 + k));
    vex::vector<double> s(q, 6 * 8);
    s = vex::reduce<vex::SUM>(x, 1);
    auto hs = download(s);
    for (size_t i = 0; i < 6; ++i) for (size_t k = 0; k < 8; ++k) {
        double r = 0; for (size_t j = 0; j < 7; ++j) r += (i * 7 + j) * 8 + k;
        CHECK_EQUAL(hs[i * 8 + k], r);
    }
}

TEST_CASE(mba_interpolates_the_data_points) {                         // mba.cpp; host model: the data itself
    // a smooth function sampled on scattered points; with enough levels the surface passes through the data
    const size_t np = 200;
    std::vector<std::array<double, 2>> p(np); std::vector<double> v(np);
    std::mt19937 rng(3); std::uniform_real_distribution<double> U(0.0, 1.0);
    for (size_t i = 0; i < np; ++i) { p[i] = {{U(rng), U(rng)}}; v[i] = std::sin(3 * p[i][0]) * std::cos(2 * p[i][1]); }
    std::array<double, 2> lo = {{-0.01, -0.01}}, hi = {{1.01, 1.01}};
    std::array<size_t, 2> grid = {{3, 3}};
    vex::mba<2> surf(ctx, lo, hi, p, v, grid, 12, 1e-12);
    std::vector<double> px(np), py(np);
    for (size_t i = 0; i < np; ++i) { px[i] = p[i][0]; py[i] = p[i][1]; }
    vex::vector<double> X(ctx, px), Y(ctx, py), Z(ctx, np);
    Z = surf(X, Y);
    auto z = download(Z);
    double worst = 0; for (size_t i = 0; i < np; ++i) worst = std::max(worst, std::fabs(z[i] - v[i]));
    CHECK_SMALL(worst, 1e-4);
    Z = 2 * surf(X * 1.0, Y + 0.0) + 1;                               // coordinates are expressions, the value is a term
    auto z2 = download(Z);
    for (size_t i = 0; i < np; i += 13) CHECK_CLOSE(z2[i] + 10, 2 * z[i] + 1 + 10, 1e-10);
}

TEST_CASE(mba_device_fit_agrees_with_the_host_fit) {
    // the same cloud fitted in HBM (default) and by the host loop (VEXCL_MBA_HOST_FIT): lattices and values agree to rounding
    const size_t np = 5000;
    std::vector<std::array<double, 2>> p(np); std::vector<double> v(np);
    std::mt19937 rng(9); std::uniform_real_distribution<double> U(0.0, 1.0);
    for (size_t i = 0; i < np; ++i) { p[i] = {{U(rng), U(rng)}}; v[i] = std::exp(-3 * p[i][0]) * std::sin(5 * p[i][1]) + 0.1 * p[i][0]; }
    p[0] = {{2.0, 2.0}};                                              // a point outside the domain is ignored by the fit
    std::array<double, 2> lo = {{-0.01, -0.01}}, hi = {{1.01, 1.01}};
    std::array<size_t, 2> grid = {{2, 3}};
    vex::mba<2> dev_fit(ctx, lo, hi, p, v, grid, 6, 1e-10);
    setenv("VEXCL_MBA_HOST_FIT", "1", 1);
    vex::mba<2> host_fit(ctx, lo, hi, p, v, grid, 6, 1e-10);
    unsetenv("VEXCL_MBA_HOST_FIT");
    for (size_t d = 0; d < 2; ++d) { CHECK_EQUAL(dev_fit.n[d], host_fit.n[d]); CHECK_EQUAL(dev_fit.stride[d], host_fit.stride[d]); CHECK_CLOSE(dev_fit.hinv[d], host_fit.hinv[d], 1e-12); }
    const size_t m = 4000;
    std::vector<double> px(m), py(m);
    for (size_t i = 0; i < m; ++i) { px[i] = U(rng); py[i] = U(rng); }
    vex::vector<double> X(ctx, px), Y(ctx, py), Zd(ctx, m), Zh(ctx, m);
    Zd = dev_fit(X, Y); Zh = host_fit(X, Y);
    auto zd = download(Zd), zh = download(Zh);
    double worst = 0; for (size_t i = 0; i < m; ++i) worst = std::max(worst, std::fabs(zd[i] - zh[i]));
    CHECK_SMALL(worst, 1e-10);
    // 1-D and 3-D, float: the device fit reproduces smooth data
    {
        std::vector<std::array<float, 1>> q1(300); std::vector<float> v1(300);
        for (size_t i = 0; i < 300; ++i) { q1[i] = {{(float)U(rng)}}; v1[i] = std::cos(4 * q1[i][0]); }
        std::array<float, 1> l1 = {{-0.01f}}, h1 = {{1.01f}}; std::array<size_t, 1> g1 = {{3}};
        vex::mba<1, float> s1(ctx, l1, h1, q1, v1, g1, 10, 1e-9f);
        std::vector<float> x1(300); for (size_t i = 0; i < 300; ++i) x1[i] = q1[i][0];
        vex::vector<float> X1(ctx, x1), Z1(ctx, 300);
        Z1 = s1(X1);
        auto z1 = download(Z1);
        float w1 = 0; for (size_t i = 0; i < 300; ++i) w1 = std::max(w1, std::fabs(z1[i] - v1[i]));
        CHECK_SMALL(w1, 2e-3);
    }
    {
        std::vector<std::array<double, 3>> q3(2000); std::vector<double> v3(2000);
        for (size_t i = 0; i < 2000; ++i) { q3[i] = {{U(rng), U(rng), U(rng)}}; v3[i] = q3[i][0] + 2 * q3[i][1] * q3[i][2]; }
        std::array<double, 3> l3 = {{-0.01, -0.01, -0.01}}, h3 = {{1.01, 1.01, 1.01}}; std::array<size_t, 3> g3 = {{2, 2, 2}};
        vex::mba<3> s3(ctx, l3, h3, q3, v3, g3, 7, 1e-12);
        setenv("VEXCL_MBA_HOST_FIT", "1", 1);
        vex::mba<3> h3fit(ctx, l3, h3, q3, v3, g3, 7, 1e-12);
        unsetenv("VEXCL_MBA_HOST_FIT");
        std::vector<double> a(500), b(500), c(500);
        for (size_t i = 0; i < 500; ++i) { a[i] = U(rng); b[i] = U(rng); c[i] = U(rng); }
        vex::vector<double> A(ctx, a), B(ctx, b), C(ctx, c), R1(ctx, 500), R2(ctx, 500);
        R1 = s3(A, B, C); R2 = h3fit(A, B, C);
        auto r1 = download(R1), r2 = download(R2);
        double w3 = 0; for (size_t i = 0; i < 500; ++i) w3 = std::max(w3, std::fabs(r1[i] - r2[i]));
        CHECK_SMALL(w3, 1e-10);
    }
}

TEST_CASE(svm_vector_is_shared_with_the_host) {                       // svm.cpp
    auto q = one_queue();
    const int n = 3000;
    vex::svm_vector<int> x(q[0], n);
    { auto p = x.map(vex::backend::MAP_WRITE); for (int i = 0; i < n; ++i) p[i] = 3 * i; }
    vex::vector<int> y(q, n);
    y = x + 1;
    auto hy = download(y);
    for (int i = 0; i < n; i += 7) CHECK_EQUAL(hy[i], 3 * i + 1);
    x -= y;                                                            // lvalue of a fused kernel
    { auto p = x.map(vex::backend::MAP_READ); for (int i = 0; i < n; i += 11) CHECK_EQUAL(p[i], -1); }
    VEX_FUNCTION(int, at, (size_t, i)(int *, p), return p[i] * 2;);
    y = at(vex::element_index(), vex::raw_pointer(x));
    hy = download(y);
    for (int i = 0; i < n; i += 5) CHECK_EQUAL(hy[i], -2);
}

namespace {
template <class S> S logistic(const S &x, double r) { return r * x * (1 - x); }
template <class S> void iterate3(S &x, double r) { S a = logistic(x, r); S b = logistic(a, r); x = logistic(b, r); }
}

TEST_CASE(symbolic_generator_records_generic_code) {                  // generator.cpp: kernel_generator, function_adapter
    const size_t n = 5000;
    std::ostringstream body;
    vex::generator::set_recorder(body);
    typedef vex::symbolic<double> sym;
    sym sx(sym::VectorParameter), sr(sym::ScalarParameter, sym::Const);
    {   // r is a kernel parameter here: the recorded code multiplies by the variable, not a literal
        sym a = sr * sx * (1 - sx);
        sx = sr * a * (1 - a);
    }
    auto kernel = vex::generator::build_kernel(ctx, "logistic2", body.str(), sx, sr);
    std::vector<double> h = random_vector<double>(n);
    vex::vector<double> X(ctx, h);
    for (int it = 0; it < 5; ++it) kernel(X, 3.7);
    auto got = download(X);
    for (size_t i = 0; i < n; i += 17) {
        double s = h[i];
        for (int it = 0; it < 10; ++it) s = logistic(s, 3.7);
        CHECK_CLOSE(got[i] + 1, s + 1, 1e-9);
    }
    // a generic functor becomes a device function usable in expressions; literals are recorded exactly
    auto step3 = vex::generator::make_function<double(double)>([](const sym &x) { sym y = x; iterate3(y, 3.625); return y; });
    vex::vector<double> Y(ctx, h), Z(ctx, n);
    Z = step3(Y) + 1;
    got = download(Z);
    for (size_t i = 0; i < n; i += 19) { double s = h[i]; iterate3(s, 3.625); CHECK_CLOSE(got[i], s + 1, 1e-10); }
}

TEST_CASE(fft_against_the_definition_and_round_trips) {              // fft.cpp: check_correctness, test_dimensions
    auto q = one_queue();
    // 1-D complex, awkward length (Bluestein), against the O(n^2) definition
    const size_t n = 211;
    std::vector<cl_double2> h(n);
    std::mt19937 rng(5); std::uniform_real_distribution<double> U(-1.0, 1.0);
    for (auto &c : h) { c.s[0] = U(rng); c.s[1] = U(rng); }
    vex::vector<cl_double2> in(q, h), out(q, n), back(q, n);
    vex::FFT<cl_double2> fft(q, n), ifft(q, n, vex::fft::inverse);
    out = fft(in);
    auto ho = download(out);
    const double pi = 3.14159265358979323846;
    for (size_t k = 0; k < n; k += 10) {
        std::complex<long double> s = 0;
        for (size_t j = 0; j < n; ++j) s += std::complex<long double>(h[j].s[0], h[j].s[1]) * std::polar<long double>(1.0L, -2.0L * pi * (long double)((j * k) % n) / n);
        CHECK_SMALL(ho[k].s[0] - (double)s.real(), 1e-10); CHECK_SMALL(ho[k].s[1] - (double)s.imag(), 1e-10);
    }
    back = ifft(out);
    auto hb = download(back);
    for (size_t j = 0; j < n; ++j) { CHECK_SMALL(hb[j].s[0] - h[j].s[0], 1e-12); CHECK_SMALL(hb[j].s[1] - h[j].s[1], 1e-12); }

    // real input, real output, batch of 2-D transforms with one four-step dimension; result used inside an expression
    const size_t B = 3, H = 6, W = 4100;
    std::vector<double> r = random_vector<double>(B * H * W);
    vex::vector<double> R(q, r), R2(q, B * H * W);
    vex::vector<cl_double2> F(q, B * H * W);
    vex::FFT<double, cl_double2> f2(q, {B, H, W}, {vex::fft::none, vex::fft::forward, vex::fft::forward});
    vex::FFT<cl_double2, double> i2(q, {B, H, W}, {vex::fft::none, vex::fft::inverse, vex::fft::inverse});
    F = f2(R);
    R2 = 0.5 * i2(F) + R;                                             // = 1.5 R
    auto h2 = download(R2);
    double worst = 0; for (size_t i = 0; i < h2.size(); ++i) worst = std::max(worst, std::fabs(h2[i] - 1.5 * r[i]));
    CHECK_SMALL(worst, 1e-12);
    auto hf = download(F);                                            // DC term of every batch = the sum of its elements
    for (size_t b = 0; b < B; ++b) {
        double s = 0; for (size_t i = 0; i < H * W; ++i) s += r[b * H * W + i];
        CHECK_CLOSE(hf[b * H * W].s[0], s, 1e-9); CHECK_SMALL(hf[b * H * W].s[1], 1e-9);
    }
    CHECK_EQUAL(vex::fft::planner().best_size(1025), 1029u);

    // real input, forward along an even last dimension, batch of rows: the half-length path (rows transformed as n/2 complex
    // numbers, unpacked inside the consumer's kernel) against the definition; vector operand and expression operand
    const size_t RB = 5, RN = 1000;
    std::vector<double> rr = random_vector<double>(RB * RN);
    vex::vector<double> RR(q, rr);
    vex::vector<cl_double2> RF(q, RB * RN), RG(q, RB * RN);
    vex::FFT<double, cl_double2> rf(q, {RB, RN}, {vex::fft::none, vex::fft::forward});
    RF = rf(RR);
    RG = rf(2 * RR + 1);
    auto hrf = download(RF), hrg = download(RG);
    for (size_t b = 0; b < RB; b += 2) for (size_t k = 0; k < RN; k += 37) {
        std::complex<long double> s1 = 0, s2 = 0;
        for (size_t j = 0; j < RN; ++j) {
            const auto w = std::polar<long double>(1.0L, -2.0L * pi * (long double)((j * k) % RN) / RN);
            s1 += (long double)rr[b * RN + j] * w; s2 += (long double)(2 * rr[b * RN + j] + 1) * w;
        }
        CHECK_SMALL(hrf[b * RN + k].s[0] - (double)s1.real(), 1e-10); CHECK_SMALL(hrf[b * RN + k].s[1] - (double)s1.imag(), 1e-10);
        CHECK_SMALL(hrg[b * RN + k].s[0] - (double)s2.real(), 1e-10); CHECK_SMALL(hrg[b * RN + k].s[1] - (double)s2.imag(), 1e-10);
    }
    {   // the same path in single precision, 2 batch dimensions
        const size_t FB = 3, FC = 2, FN = 64;
        std::vector<float> fr = random_vector<float>(FB * FC * FN);
        vex::vector<float> FR(q, fr);
        vex::vector<cl_float2> FF(q, FB * FC * FN);
        vex::FFT<float, cl_float2> ff(q, {FB, FC, FN}, {vex::fft::none, vex::fft::none, vex::fft::forward});
        FF = ff(FR);
        auto hff = download(FF);
        for (size_t row = 0; row < FB * FC; ++row) for (size_t k = 0; k < FN; k += 5) {
            std::complex<double> s1 = 0;
            for (size_t j = 0; j < FN; ++j) s1 += (double)fr[row * FN + j] * std::polar<double>(1.0, -2.0 * pi * (double)((j * k) % FN) / FN);
            CHECK_SMALL(hff[row * FN + k].s[0] - (float)s1.real(), 2e-4); CHECK_SMALL(hff[row * FN + k].s[1] - (float)s1.imag(), 2e-4);
        }
    }
    // real in, real out (the real part of the spectrum), and the odd-length fallback
    vex::vector<double> RE(q, RB * RN);
    vex::FFT<double, double> rre(q, {RB, RN}, {vex::fft::none, vex::fft::forward});
    RE = rre(RR);
    auto hre = download(RE);
    for (size_t i = 0; i < RB * RN; i += 41) CHECK_SMALL(hre[i] - hrf[i].s[0], 1e-12);
    const size_t ON = 999;
    vex::vector<double> OR_(q, std::vector<double>(rr.begin(), rr.begin() + ON));
    vex::vector<cl_double2> OF(q, ON);
    vex::FFT<double, cl_double2> of(q, ON);
    OF = of(OR_);
    auto hof = download(OF);
    for (size_t k = 0; k < ON; k += 53) {
        std::complex<long double> s1 = 0;
        for (size_t j = 0; j < ON; ++j) s1 += (long double)rr[j] * std::polar<long double>(1.0L, -2.0L * pi * (long double)((j * k) % ON) / ON);
        CHECK_SMALL(hof[k].s[0] - (double)s1.real(), 1e-10); CHECK_SMALL(hof[k].s[1] - (double)s1.imag(), 1e-10);
    }
}

TEST_CASE(constant_vectors_and_pointers) {                            // vector_arithmetics.cpp:300-316, vector_pointer.cpp:84-120
    auto q = one_queue();
    const size_t N = 1000;
    int table[] = {5, 7, 11, 13};
    vex::vector<int> x(q, 4, table, vex::backend::MEM_READ_ONLY), y(q, N);
    y = vex::permutation(vex::element_index() % 4)(vex::constant(x)) + 1;
    auto h = download(y);
    for (size_t i = 0; i < N; ++i) CHECK_EQUAL(h[i], table[i % 4] + 1);
    auto X = vex::constant_pointer(x);
    y = X[vex::element_index() % 4];
    h = download(y);
    for (size_t i = 0; i < N; ++i) CHECK_EQUAL(h[i], table[i % 4]);
    VEX_FUNCTION(int, lookup, (int, idx)(vex::constant_ptr<int>, t), return t[idx % 4];);
    y = lookup(vex::element_index(), vex::constant_pointer(x));
    h = download(y);
    for (size_t i = 0; i < N; ++i) CHECK_EQUAL(h[i], table[i % 4]);
}

TEST_CASE(device_filters_of_the_other_backends) {                     // backend/opencl/filter.hpp, backend/cuda/filter.hpp
    const auto dev = ctx.queue(0).device();
    CHECK(vex::Filter::CC(9, 0)(dev));                                // gfx9xx
    CHECK(!vex::Filter::CC(99, 0)(dev));
    CHECK(vex::Filter::Extension("cl_khr_fp64")(dev));
    CHECK(!vex::Filter::GLSharing(dev));
    CHECK(vex::Filter::CLVersion(2, 0)(dev));
    vex::Context one(vex::Filter::CC(9, 0) && vex::Filter::CLVersion(1, 2) && vex::Filter::Count(1));
    CHECK_EQUAL(one.size(), 1u);
}

TEST_CASE(profiler_device_sections_wait_for_the_device) {             // profiler.hpp:249-269: toc of a tic_cl section finishes the queues
    const size_t n = size_t(1) << 26;
    vex::vector<double> x(ctx, n);
    x = 1;
    vex::profiler<> prof(ctx);
    prof.tic_cl("twenty passes");
    for (int i = 0; i < 20; ++i) x = sin(x) + 1;
    const double t = prof.toc("twenty passes");
    CHECK(t > 1e-3);                                                  // 20 x 1 GiB cannot cross the HBM pins in less (8 TB/s: 2.7 ms)
    prof.tic_cpu("host section");
    CHECK(prof.toc("host section") < 1e-3);
}
