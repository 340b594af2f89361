// BASELINE.json's headline through the C++ header API: `y = A * x` with A a
// vex::SpMat<double, int, int> holding the 3-D Poisson matrix of examples/benchmark.cpp:364-415
// on a 512^3 grid, and the same 7-point pattern with variable coefficients.  The reference harness
// (examples/benchmark.cpp:353-477) assembles the matrix on the host and uploads it (12 GB of host
// arrays at this size); here the CSR arrays are generated in HBM and handed to the device-array
// constructor of vex::SpMat, which converts them on the device (vexhip_spmat_create).
// Prints one JSON object per matrix: ms per product (HIP events on the compute queue, M products
// then one synchronisation -- benchmark.cpp:426-433), GFLOP/s = 2 nnz / t, and the bytes the chosen
// storage streams per product.   Usage: spmv_headline [grid = 512] [products = 100]
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vexcl/vexcl.hpp>

static const char *storage_name(int f) {
    switch (f) {
        case VEXHIP_SPMAT_SELL8V: return "sell8v (1-byte diagonal codes + 1-byte value codes)";
        case VEXHIP_SPMAT_SELL8:  return "sell8 (1-byte diagonal codes, fp64 values)";
        case VEXHIP_SPMAT_SELL:   return "sell32 (32-bit columns, fp64 values)";
        case VEXHIP_SPMAT_CSR:    return "csr";
    }
    return "?";
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 512;
    const int M = argc > 2 ? std::atoi(argv[2]) : 100;
    vex::Context ctx(vex::Filter::Env && vex::Filter::Count(1));
    if (!ctx) { std::cerr << "no device" << std::endl; return 1; }
    const vex::backend::command_queue &q = ctx.queue(0);
    const int dev = q.device_ordinal();
    const size_t N = (size_t)(n * n * n), nnz = (size_t)vexhip_poisson3d_nnz(n);
    void *e0 = nullptr, *e1 = nullptr;
    vex::backend::check(vexhip_event_create(dev, 1, &e0));
    vex::backend::check(vexhip_event_create(dev, 1, &e1));

    for (int variable = 0; variable < 2; ++variable) {
        vex::SpMat<double, int, int> A;
        {
            vex::backend::device_vector<int> ptr(q, N + 1), col(q, nnz);
            vex::backend::device_vector<double> val(q, nnz);
            if (variable) vex::backend::check(vexhip_diffusion3d_strip_f64_i32(dev, q.raw(), n, 0, (int64_t)N, 7, ptr.raw(), col.raw(), val.raw()));
            else vex::backend::check(vexhip_poisson3d_csr_f64_i32(dev, q.raw(), n, ptr.raw(), col.raw(), val.raw()));
            A = vex::SpMat<double, int, int>(ctx.queue(), N, N, nnz, ptr, col, val);
        }       // the CSR arrays are released here unless the matrix kept them (plain CSR storage)
        vex::vector<double> x(ctx, N), y(ctx, N);
        // the same x as bench.py (counter hash, seed 42): sum(y) of the Poisson row must equal bench.py's checksum
        vex::backend::check(vexhip_fill_hash(dev, q.raw(), VEXHIP_F64, 42, x(0).raw(), (int64_t)N));
        // warm-up: the set-up above leaves the device idle for milliseconds at a time, and after such a gap the first ~20
        // products run up to 12 % slow (tools/r02_ramp.py); the timed products follow the warm-up without a host sync
        for (int i = 0; i < 40; ++i) y = A * x;
        vex::backend::check(vexhip_event_record(dev, e0, q.raw()));
        for (int i = 0; i < M; ++i) y = A * x;
        vex::backend::check(vexhip_event_record(dev, e1, q.raw()));
        vex::backend::check(vexhip_event_sync(dev, e1));
        float total = 0;
        vex::backend::check(vexhip_event_elapsed_ms(dev, e0, e1, &total));
        const double ms = total / M;
        vex::Reductor<double, vex::SUM_Kahan> sum(ctx);
        const double checksum = sum(y);
        const vexhip_spmat_info &info = A.storage_info();
        const double moved = (double)info.matrix_bytes + 16.0 * N;   // stored matrix + x once + y once
        std::printf("{\"row\": \"vex::SpMat<double,int,int> y = A*x, %s 7-point %lld^3\", \"front_end\": \"C++ vexcl/spmat.hpp\", "
                    "\"storage\": \"%s\", \"rows\": %zu, \"nnz\": %zu, \"ms\": %.5f, \"gflops\": %.1f, "
                    "\"bytes_streamed\": %.0f, \"streamed_gbps\": %.1f, \"streamed_frac_of_8TBps\": %.4f, "
                    "\"csr_algorithmic_gbps\": %.1f, \"sum_y\": %.17g}\n",
                variable ? "variable-coefficient" : "Poisson", (long long)n, storage_name(info.format), N, nnz, ms, 2.0 * nnz / ms / 1e6,
                moved, moved / ms / 1e6, moved / ms / 1e6 / 8000.0, (12.0 * nnz + 4.0 * (N + 1) + 16.0 * N) / ms / 1e6, checksum);
        std::fflush(stdout);
    }
    return 0;
}
