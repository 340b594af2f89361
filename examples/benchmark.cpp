// Port of the reference harness examples/benchmark.cpp (SAXPY :84-148, vector
// arithmetic :153-216, reductor :220-278, SpMV :353-477, sort :669-757,
// scan :761-846, stencil :282-349, CCSR :481-606) onto the MI355X implementation.  Same problem sizes, same
// work formulas, same printed fields; options are plain "--name value" pairs
// (no Boost.program_options).  The RNG section is out of scope.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <numeric>
#include <vector>
#include <vexcl/vexcl.hpp>

static struct {
    bool bm_saxpy = true, bm_vector = true, bm_reductor = true, bm_stencil = true, bm_spmv = true, bm_spmv_ccsr = true, bm_sort = true, bm_scan = true, bm_cpu = true;
    size_t spmv_n = 128;
    size_t spmv_m = 1024;
} options;

template <typename real>
std::pair<double, double> benchmark_saxpy(const vex::Context &ctx, vex::profiler<> &prof) {
    const size_t N = 1024 * 1024, M = 1024;
    std::vector<real> A(N, 0), B = std::vector<real>(N);
    for (size_t i = 0; i < N; ++i) B[i] = real(i % 17) / 17;
    std::vector<real> alphavec(1, real(0.5));
    real alpha = alphavec[0];
    vex::vector<real> a(ctx, A), b(ctx, B);
    auto ta = vex::tag<1>(a);
    ta = alpha * ta + b;
    ta = static_cast<real>(0);
    prof.tic_cpu("OpenCL");
    for (size_t i = 0; i < M; i++) ta = alpha * ta + b;
    ctx.finish();
    double t = prof.toc("OpenCL");
    double gflops = (2.0 * N * M) / t / 1e9, bwidth = (3.0 * N * M * sizeof(real)) / t / 1e9;
    std::cout << "Vector SAXPY (" << vex::type_name<real>() << ")\n  OpenCL\n    GFLOPS:    " << gflops
              << "\n    Bandwidth: " << bwidth << std::endl;
    if (options.bm_cpu) {
        prof.tic_cpu("C++");
        for (size_t i = 0; i < M; i++) for (size_t j = 0; j < N; j++) A[j] = alpha * A[j] + B[j];
        double tc = prof.toc("C++");
        std::cout << "  C++\n    GFLOPS:    " << (2.0 * N * M) / tc / 1e9 << "\n    Bandwidth: "
                  << (3.0 * N * M * sizeof(real)) / tc / 1e9 << std::endl;
        vex::copy(A, b);
        vex::Reductor<real, vex::SUM> sum(ctx);
        a -= b;
        std::cout << "  res = " << sum(a * a) << std::endl << std::endl;
    }
    return std::make_pair(gflops, bwidth);
}

template <typename real>
std::pair<double, double> benchmark_vector(const vex::Context &ctx, vex::profiler<> &prof) {
    const size_t N = 1024 * 1024, M = 1024;
    std::vector<real> A(N, 0), B(N), C(N), D(N);
    for (size_t i = 0; i < N; ++i) { B[i] = real(i % 13) / 13; C[i] = real(i % 7) / 7; D[i] = real(i % 5) / 5; }
    vex::vector<real> a(ctx, A), b(ctx, B), c(ctx, C), d(ctx, D);
    a += b + c * d;
    a = 0;
    prof.tic_cpu("OpenCL");
    for (size_t i = 0; i < M; i++) a += b + c * d;
    ctx.finish();
    double t = prof.toc("OpenCL");
    double gflops = (3.0 * N * M) / t / 1e9, bwidth = (5.0 * N * M * sizeof(real)) / t / 1e9;
    std::cout << "Vector arithmetic (" << vex::type_name<real>() << ")\n  OpenCL\n    GFLOPS:    " << gflops
              << "\n    Bandwidth: " << bwidth << std::endl;
    if (options.bm_cpu) {
        prof.tic_cpu("C++");
        for (size_t i = 0; i < M; i++) for (size_t j = 0; j < N; j++) A[j] += B[j] + C[j] * D[j];
        double tc = prof.toc("C++");
        std::cout << "  C++\n    GFLOPS:    " << (3.0 * N * M) / tc / 1e9 << "\n    Bandwidth: "
                  << (5.0 * N * M * sizeof(real)) / tc / 1e9 << std::endl;
        vex::copy(A, b);
        vex::Reductor<real, vex::MAX> max(ctx);
        std::cout << "  res = " << max(fabs(a - b)) << std::endl << std::endl;
    }
    return std::make_pair(gflops, bwidth);
}

template <typename real>
std::pair<double, double> benchmark_reductor(const vex::Context &ctx, vex::profiler<> &prof) {
    const size_t N = 16 * 1024 * 1024, M = 64;
    std::vector<real> A(N), B(N);
    for (size_t i = 0; i < N; ++i) { A[i] = real(i % 11) / 11; B[i] = real(i % 3) / 3; }
    vex::vector<real> a(ctx, A), b(ctx, B);
    vex::Reductor<real, vex::SUM> sum(ctx);
    double sum_cl = sum(a * b);
    sum_cl = 0;
    prof.tic_cpu("OpenCL");
    for (size_t i = 0; i < M; i++) sum_cl += sum(a * b);
    ctx.finish();
    double t = prof.toc("OpenCL");
    double gflops = 2.0 * N * M / t / 1e9, bwidth = 2.0 * N * M * sizeof(real) / t / 1e9;
    std::cout << "Reduction (" << vex::type_name<real>() << ")\n  OpenCL\n    GFLOPS:    " << gflops
              << "\n    Bandwidth: " << bwidth << std::endl;
    if (options.bm_cpu) {
        double sum_cpp = 0;
        prof.tic_cpu("C++");
        for (size_t i = 0; i < M; i++) sum_cpp += std::inner_product(A.begin(), A.end(), B.begin(), static_cast<real>(0));
        double tc = prof.toc("C++");
        std::cout << "  C++\n    GFLOPS:    " << 2.0 * N * M / tc / 1e9 << "\n    Bandwidth: "
                  << 2.0 * N * M * sizeof(real) / tc / 1e9 << std::endl;
        std::cout << "  res = " << std::fabs(sum_cl - sum_cpp) / std::fabs(sum_cpp) << std::endl << std::endl;
    }
    return std::make_pair(gflops, bwidth);
}

// examples/benchmark.cpp:282-349
template <typename real>
std::pair<double, double> benchmark_stencil(const vex::Context &ctx, vex::profiler<> &prof) {
    const long N = 1024 * 1024, M = 1024;
    std::vector<real> A(N), B(N);
    for (long i = 0; i < N; ++i) A[i] = real(i % 29) / 29;
    std::vector<real> S(21, static_cast<real>(1) / 21);
    long center = S.size() / 2;
    vex::stencil<real> s(ctx, S, center);
    vex::vector<real> a(ctx, A), b(ctx, N);
    b = a * s;
    prof.tic_cpu("OpenCL");
    for (long i = 0; i < M; i++) b = a * s;
    ctx.finish();
    double t = prof.toc("OpenCL");
    double gflops = 2.0 * S.size() * N * M / t / 1e9, bwidth = 2.0 * S.size() * N * M * sizeof(real) / t / 1e9;
    std::cout << "Stencil convolution (" << vex::type_name<real>() << ")\n  OpenCL\n    GFLOPS:    " << gflops
              << "\n    Bandwidth: " << bwidth << std::endl;
    if (options.bm_cpu) {
        const long Mc = 16;
        prof.tic_cpu("C++");
        for (long j = 0; j < Mc; j++)
            for (long i = 0; i < N; i++) {
                real sum = 0;
                for (long k = 0; k < (long)S.size(); k++) sum += S[k] * A[std::min<long>(N - 1, std::max<long>(0, i + k - center))];
                B[i] = sum;
            }
        double tc = prof.toc("C++");
        std::cout << "  C++ (" << Mc << " passes)\n    GFLOPS:    " << 2.0 * S.size() * N * Mc / tc / 1e9 << std::endl;
        vex::Reductor<real, vex::MAX> max(ctx);
        vex::copy(B, a);
        std::cout << "  res = " << max(fabs(a - b)) << std::endl << std::endl;
    }
    return std::make_pair(gflops, bwidth);
}

template <typename real>
std::pair<double, double> benchmark_spmv(const vex::Context &ctx, vex::profiler<> &prof) {
    const size_t n = options.spmv_n, N = n * n * n, M = options.spmv_m;
    const real h2i = (n - 1) * (n - 1);
    std::vector<size_t> row; std::vector<unsigned> col; std::vector<real> val;
    std::vector<real> X(N, static_cast<real>(1e-2)), Y(N, 0);
    row.reserve(N + 1); col.reserve(6 * (n - 2) * (n - 2) * (n - 2) + N); val.reserve(6 * (n - 2) * (n - 2) * (n - 2) + N);
    row.push_back(0);
    for (size_t k = 0, idx = 0; k < n; k++) for (size_t j = 0; j < n; j++) for (size_t i = 0; i < n; i++, idx++) {
        if (i == 0 || i == n - 1 || j == 0 || j == n - 1 || k == 0 || k == n - 1) {
            col.push_back(idx); val.push_back(1); row.push_back(row.back() + 1);
        } else {
            col.push_back(idx - n * n); val.push_back(-h2i);
            col.push_back(idx - n);     val.push_back(-h2i);
            col.push_back(idx - 1);     val.push_back(-h2i);
            col.push_back(idx);         val.push_back(6 * h2i);
            col.push_back(idx + 1);     val.push_back(-h2i);
            col.push_back(idx + n);     val.push_back(-h2i);
            col.push_back(idx + n * n); val.push_back(-h2i);
            row.push_back(row.back() + 7);
        }
    }
    size_t nnz = row.back();
    vex::SpMat<real, unsigned> A(ctx, N, N, row.data(), col.data(), val.data());
    vex::vector<real> x(ctx, X), y(ctx, Y);
    y += A * x;
    y = 0;
    prof.tic_cpu("OpenCL");
    for (size_t i = 0; i < M; i++) y += A * x;
    ctx.finish();
    double t = prof.toc("OpenCL");
    double gflops = M / t / 1e9 * (2.0 * nnz + N);
    double bwidth = M / t / 1e9 * (nnz * (2 * sizeof(real) + sizeof(size_t)) + 4 * N * sizeof(real));
    // roofline denominator: int32 indices, x counted once, y += form
    double alg = M / t / 1e9 * (nnz * (sizeof(real) + 4.0) + (N + 1) * 4.0 + 3.0 * N * sizeof(real));
    std::cout << "SpMV (" << vex::type_name<real>() << ")\n  OpenCL\n    GFLOPS:    " << gflops
              << "\n    Bandwidth: " << bwidth << "\n    Algorithmic GB/s: " << alg << std::endl;
    if (options.bm_cpu) {
        const size_t Mc = std::max<size_t>(1, M / 64);
        prof.tic_cpu("C++");
        for (size_t k = 0; k < Mc; k++)
            for (size_t i = 0; i < N; i++) {
                real s = 0;
                for (size_t j = row[i]; j < row[i + 1]; j++) s += val[j] * X[col[j]];
                Y[i] += s;
            }
        double tc = prof.toc("C++");
        std::cout << "  C++ (" << Mc << " products)\n    GFLOPS:    " << Mc / tc / 1e9 * (2.0 * nnz + N)
                  << "\n    Bandwidth: " << Mc / tc / 1e9 * (nnz * (2 * sizeof(real) + sizeof(size_t)) + 4 * N * sizeof(real)) << std::endl;
        for (auto &v : Y) v *= real(M) / real(Mc);
        vex::copy(Y, x);
        y -= x;
        vex::Reductor<real, vex::SUM> sum(ctx);
        std::cout << "  res = " << sum(y * y) << std::endl << std::endl;
    }
    return std::make_pair(gflops, bwidth);
}

// examples/benchmark.cpp:481-606: the same operator in compressed-stencil form
template <typename real>
std::pair<double, double> benchmark_spmv_ccsr(const vex::Context &ctx, vex::profiler<> &prof) {
    const size_t n = options.spmv_n, N = n * n * n, M = options.spmv_m;
    const real h2i = (n - 1) * (n - 1);
    std::vector<size_t> idx; idx.reserve(N);
    std::vector<size_t> row = {0, 1, 8};
    std::vector<int> col = {0, -static_cast<int>(n * n), -static_cast<int>(n), -1, 0, 1, static_cast<int>(n), static_cast<int>(n * n)};
    std::vector<real> val = {1, -h2i, -h2i, -h2i, h2i * 6, -h2i, -h2i, -h2i};
    std::vector<real> X(N, static_cast<real>(1e-2)), Y(N, 0);
    for (size_t k = 0; k < n; k++) for (size_t j = 0; j < n; j++) for (size_t i = 0; i < n; i++)
        idx.push_back((i == 0 || i == n - 1 || j == 0 || j == n - 1 || k == 0 || k == n - 1) ? 0 : 1);
    size_t nnz = 6 * (n - 2) * (n - 2) * (n - 2) + N;
    vex::SpMatCCSR<real, int> A(ctx.queue(0), N, 2, idx.data(), row.data(), col.data(), val.data());
    std::vector<vex::command_queue> q1(1, ctx.queue(0));
    vex::vector<real> x(q1, X), y(q1, Y);
    y += A * x;
    y = 0;
    prof.tic_cpu("OpenCL");
    for (size_t i = 0; i < M; i++) y += A * x;
    ctx.finish();
    double t = prof.toc("OpenCL");
    double gflops = M / t / 1e9 * (2.0 * nnz + N);
    double bwidth = M / t / 1e9 * (2 * 8 * sizeof(real) + N * (3 * sizeof(real) + sizeof(size_t)) + 3 * sizeof(size_t));
    double alg = M / t / 1e9 * (N * (3.0 * sizeof(real) + 4.0));
    std::cout << "SpMV (CCSR) (" << vex::type_name<real>() << ")\n  OpenCL\n    GFLOPS:    " << gflops
              << "\n    Bandwidth: " << bwidth << "\n    Algorithmic GB/s: " << alg << std::endl;
    if (options.bm_cpu) {
        const size_t Mc = std::max<size_t>(1, M / 64);
        prof.tic_cpu("C++");
        for (size_t k = 0; k < Mc; k++)
            for (size_t i = 0; i < N; i++) {
                real s = 0;
                for (size_t j = row[idx[i]]; j < row[idx[i] + 1]; j++) s += val[j] * X[i + col[j]];
                Y[i] += s;
            }
        double tc = prof.toc("C++");
        std::cout << "  C++ (" << Mc << " products)\n    GFLOPS:    " << Mc / tc / 1e9 * (2.0 * nnz + N) << std::endl;
        for (auto &v : Y) v *= real(M) / real(Mc);
        vex::copy(Y, x);
        y -= x;
        vex::Reductor<real, vex::SUM> sum(q1);
        std::cout << "  res = " << sum(y * y) << std::endl << std::endl;
    }
    return std::make_pair(gflops, bwidth);
}

template <typename real>
double benchmark_sort(const vex::Context &ctx, vex::profiler<> &prof) {
    typedef typename std::conditional<std::is_same<float, real>::value, cl_uint, cl_ulong>::type key_type;
    const size_t N = 16 * 1024 * 1024, M = 16;
    std::vector<key_type> x0(N), x1(N);
    unsigned long long s = 88172645463325252ull;
    for (auto &v : x0) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = static_cast<key_type>(s); }
    vex::vector<key_type> X0(ctx, x0), X1(ctx, N);
    X1 = X0; vex::sort(X1);
    double tot = 0;
    for (size_t i = 0; i < M; i++) {
        X1 = X0; ctx.finish();
        prof.tic_cpu("OpenCL");
        vex::sort(X1); ctx.finish();
        tot += prof.toc("OpenCL");
    }
    double rate = N * M / tot;
    std::cout << "Sort (" << vex::type_name<key_type>() << ")\n    VexCL:         " << rate << " keys/sec\n";
    if (options.bm_cpu) {
        x1 = x0;
        prof.tic_cpu("STL");
        std::sort(x1.begin(), x1.end());
        double tc = prof.toc("STL");
        std::cout << "    STL:           " << N / tc << " keys/sec\n";
        std::vector<key_type> got(N); vex::copy(X1, got);
        std::cout << "    match:         " << (got == x1 ? "yes" : "NO") << "\n";
    }
    std::cout << std::endl;
    return rate;
}

template <typename real>
double benchmark_scan(const vex::Context &ctx, vex::profiler<> &prof) {
    typedef typename std::conditional<std::is_same<float, real>::value, cl_uint, cl_ulong>::type key_type;
    const size_t N = 16 * 1024 * 1024, M = 16;
    std::vector<key_type> x0(N), x1(N);
    for (size_t i = 0; i < N; ++i) x0[i] = static_cast<key_type>(i % 1000);
    vex::vector<key_type> X0(ctx, x0), X1(ctx, N);
    vex::exclusive_scan(X0, X1);
    ctx.finish();
    prof.tic_cpu("OpenCL");
    for (size_t i = 0; i < M; i++) vex::exclusive_scan(X0, X1);
    ctx.finish();
    double t = prof.toc("OpenCL");
    double rate = N * M / t;
    std::cout << "Scan (" << vex::type_name<key_type>() << ")\n    VexCL:         " << rate << " keys/sec\n";
    if (options.bm_cpu) {
        prof.tic_cpu("STL");
        x1[0] = 0; std::partial_sum(x0.begin(), x0.end() - 1, x1.begin() + 1);
        double tc = prof.toc("STL");
        std::cout << "    STL:           " << N / tc << " keys/sec\n";
        std::vector<key_type> got(N); vex::copy(X1, got);
        std::cout << "    match:         " << (got == x1 ? "yes" : "NO") << "\n";
    }
    std::cout << std::endl;
    return rate;
}

template <typename real>
void run_tests(const vex::Context &ctx, vex::profiler<> &prof) {
    std::cout << "----------------------------------------------------------\nProfiling \"" << vex::type_name<real>()
              << "\" performance\n----------------------------------------------------------" << std::endl;
    std::ofstream log("profile_" + vex::type_name<real>() + ".dat", std::ios::app);
    log << ctx.size() << " ";
    prof.tic_cpu(vex::type_name<real>());
    if (options.bm_saxpy) { auto r = benchmark_saxpy<real>(ctx, prof); log << r.first << " " << r.second << " "; }
    if (options.bm_vector) { auto r = benchmark_vector<real>(ctx, prof); log << r.first << " " << r.second << " "; }
    if (options.bm_reductor) { auto r = benchmark_reductor<real>(ctx, prof); log << r.first << " " << r.second << " "; }
    if (options.bm_stencil) { auto r = benchmark_stencil<real>(ctx, prof); log << r.first << " " << r.second << " "; }
    if (options.bm_spmv) { auto r = benchmark_spmv<real>(ctx, prof); log << r.first << " " << r.second << " "; }
    if (options.bm_spmv_ccsr) { auto r = benchmark_spmv_ccsr<real>(ctx, prof); log << r.first << " " << r.second << " "; }
    if (options.bm_sort) log << benchmark_sort<real>(ctx, prof) << " ";
    if (options.bm_scan) log << benchmark_scan<real>(ctx, prof) << " ";
    prof.toc(vex::type_name<real>());
    log << std::endl;
}

int main(int argc, char *argv[]) {
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i]; int v = std::atoi(argv[i + 1]);
        if (k == "--bm_saxpy") options.bm_saxpy = v; else if (k == "--bm_vector") options.bm_vector = v;
        else if (k == "--bm_reductor") options.bm_reductor = v; else if (k == "--bm_spmv") options.bm_spmv = v;
        else if (k == "--bm_stn") options.bm_stencil = v; else if (k == "--bm_spmv_ccsr") options.bm_spmv_ccsr = v; else if (k == "--bm_sort") options.bm_sort = v; else if (k == "--bm_scan") options.bm_scan = v;
        else if (k == "--bm_cpu") options.bm_cpu = v; else if (k == "--spmv_n") options.spmv_n = v;
        else if (k == "--spmv_m") options.spmv_m = v;
    }
    try {
        vex::Context ctx(vex::Filter::Env && vex::Filter::DoublePrecision);
        if (!ctx) { std::cerr << "No compute devices" << std::endl; return 1; }
        std::cout << ctx << std::endl;
        vex::profiler<> prof(ctx);
        run_tests<float>(ctx, prof);
        run_tests<double>(ctx, prof);
        std::cout << prof << std::endl;
    } catch (const vex::error &e) {
        std::cerr << e << std::endl;
        return 1;
    }
    return 0;
}
