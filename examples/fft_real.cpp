// Real-input transforms through vex::FFT: out = fft(in) for 65 536 rows of 1024 real fp64 values, timed with the
// half-length path (default) -- rows transformed as 512 complex numbers, unpacking fused into the assignment kernel --
// and, for comparison, the same data as a complex transform (what a real operand costs without that path).
#include <cstdlib>
#include <iostream>
#include <vexcl/vexcl.hpp>

int main() {
    vex::Context ctx(vex::Filter::Env && vex::Filter::Count(1));
    const size_t rows = 65536, n = 1024;
    vex::vector<double> x(ctx, rows * n);
    vex::vector<cl_double2> xc(ctx, rows * n), y(ctx, rows * n);
    x = vex::Random<double>()(vex::element_index(), 42);
    VEX_FUNCTION(cl_double2, widen, (double, v), double2 r = {v, 0}; return r;);
    xc = widen(x);
    vex::FFT<double, cl_double2> real_fft(ctx, {rows, n}, {vex::fft::none, vex::fft::forward});
    vex::FFT<cl_double2> cplx_fft(ctx, {rows, n}, {vex::fft::none, vex::fft::forward});
    setenv("VEXCL_FFT_NO_HALF", "1", 1);
    vex::FFT<double, cl_double2> wide_fft(ctx, {rows, n}, {vex::fft::none, vex::fft::forward});   // real operand widened to complex first
    unsetenv("VEXCL_FFT_NO_HALF");
    vex::profiler<> prof(ctx);
    y = real_fft(x); y = cplx_fft(xc);
    const int reps = 20;
    prof.tic_cl("y = fft(real x)");
    for (int i = 0; i < reps; ++i) y = real_fft(x);
    const double t_real = prof.toc("y = fft(real x)") / reps;
    prof.tic_cl("y = fft(complex x)");
    for (int i = 0; i < reps; ++i) y = cplx_fft(xc);
    const double t_cplx = prof.toc("y = fft(complex x)") / reps;
    y = wide_fft(x);
    prof.tic_cl("y = fft(real x), full length");
    for (int i = 0; i < reps; ++i) y = wide_fft(x);
    const double t_wide = prof.toc("y = fft(real x), full length") / reps;
    std::cout << "real input, widened to complex and transformed at full length: " << t_wide * 1e3 << " ms\n";
    std::cout << "real input   : " << t_real * 1e3 << " ms per transform + assignment\n"
              << "complex input: " << t_cplx * 1e3 << " ms per transform + assignment" << std::endl;
}
