// BASELINE.json's headline through the C++ header API: `y = A * x` with A a
// vex::SpMat<double, int, int> holding the 3-D Poisson matrix of examples/benchmark.cpp:364-415
// on a 512^3 grid, and the same 7-point pattern with variable coefficients.  The reference harness
// (examples/benchmark.cpp:353-477) assembles the matrix on the host and uploads it (12 GB of host
// arrays at this size); here the CSR arrays are generated in HBM and handed to the device-array
// constructors of vex::SpMat, which convert them on the device (vexhip_spmat_create).
// Prints one JSON object per matrix: ms per product (M products, then one synchronisation --
// benchmark.cpp:426-433), GFLOP/s = 2 nnz / t, and the bytes the chosen storage streams per product.
//
//   spmv_headline [grid = 512] [products = 100]                 one device (Filter::Count(1))
//   spmv_headline [grid] [products] --devices D|all             ONE vex::Context driving D GPUs (north_star's multi-GPU shape:
//       VexCL's own multi-device partitioning, vector.hpp:131-167, spmat.hpp:74-106): every device generates ITS row strip in
//       its own HBM, vex::SpMat is built from the per-device strips (all devices at once), the product runs the ghost
//       exchange of vexcl/exchange.hpp (RCCL over xGMI between distinct GPUs).  Reported: wall time per product over the
//       whole context (host clock around M products + ctx.finish()), the slowest device's event time, the per-device step
//       phases (local part / wait for ghosts / remote part), and sum(y) -- the same for every D up to rounding.
//       Round 6: where every device's remote columns are its two neighbouring planes (this workload), the product of a device is
//       ONE launch that reads the neighbours' boundary planes of x in place ("step" in the output; VEXCL_HALO=off keeps the exchange);
//       --check compares y bit for bit with the product of the same matrix on one device.
//   VEXCL_LOGICAL_DEVICES=k (tests) makes one GPU appear k times, as in the reference's test fixture.
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <vexcl/vexcl.hpp>

static const char *storage_name(const vexhip_spmat_info &info) {
    if (info.format == VEXHIP_SPMAT_SELL8V && info.grid.usable && !info.sell && !info.code_pool)
        return "by grid line (a class per line, 7 x nx value codes per class)";
    switch (info.format) {
        case VEXHIP_SPMAT_SELL8V: return "sell8v (1-byte diagonal codes + 1-byte value codes)";
        case VEXHIP_SPMAT_SELL8: return "sell8 (1-byte diagonal codes, fp64 values)";
        case VEXHIP_SPMAT_SELL: return "sell32 (32-bit columns, fp64 values)";
        case VEXHIP_SPMAT_CSR: return "csr";
    }
    return "?";
}

static int multi_device(int64_t n, int M, int want, bool check_bits) {
    std::shared_ptr<vex::Context> context = want > 0 ? std::make_shared<vex::Context>(vex::Filter::Env && vex::Filter::Count(want))
                                                     : std::make_shared<vex::Context>(vex::Filter::Env);
    if (!*context) { std::cerr << "no device" << std::endl; return 1; }
    if (const char *e = std::getenv("VEXCL_LOGICAL_DEVICES")) {
        // one GPU appears k times (the reference's test fixture duplicates its queue the same way, tests/context_setup.hpp:24-38)
        const int k = std::atoi(e);
        std::vector<vex::backend::context> c = context->context();
        std::vector<vex::backend::command_queue> qq = context->queue();
        while ((int)qq.size() < k) {
            vex::Context more(vex::Filter::Env && vex::Filter::Count(1));
            c.push_back(more.context(0)); qq.push_back(more.queue(0));
        }
        context = std::make_shared<vex::Context>(c, qq);
    }
    vex::Context &ctx = *context;
    const std::vector<vex::backend::command_queue> &q = ctx.queue();
    const unsigned D = (unsigned)q.size();
    const size_t N = (size_t)(n * n * n), nnz = (size_t)vexhip_poisson3d_nnz(n);
    const std::vector<size_t> part = vex::partition(N, q);
    const unsigned long long GOLD = 0x9E3779B97F4A7C15ull;

    for (int variable = 0; variable < 2; ++variable) {
        const auto t_setup0 = std::chrono::steady_clock::now();
        vex::SpMat<double, int, int> A;
        {
            std::vector<vex::backend::device_vector<int>> ptr(D), col(D);
            std::vector<vex::backend::device_vector<double>> val(D);
            std::vector<size_t> strip_nnz(D);
            for (unsigned d = 0; d < D; ++d) {       // every device generates its strip in its own HBM (global column ids)
                const int dev = q[d].device_ordinal();
                const int64_t r0 = (int64_t)part[d], r1 = (int64_t)part[d + 1];
                strip_nnz[d] = (size_t)vexhip_poisson3d_strip_nnz(n, r0, r1);
                ptr[d] = vex::backend::device_vector<int>(q[d], (size_t)(r1 - r0) + 1);
                col[d] = vex::backend::device_vector<int>(q[d], std::max<size_t>(1, strip_nnz[d]));
                val[d] = vex::backend::device_vector<double>(q[d], std::max<size_t>(1, strip_nnz[d]));
                if (variable) vex::backend::check(vexhip_diffusion3d_strip_f64_i32(dev, q[d].raw(), n, r0, r1, 7, ptr[d].raw(), col[d].raw(), val[d].raw()));
                else vex::backend::check(vexhip_poisson3d_strip_f64_i32(dev, q[d].raw(), n, r0, r1, ptr[d].raw(), col[d].raw(), val[d].raw()));
            }
            A = vex::SpMat<double, int, int>(q, N, N, ptr, col, val, strip_nnz);
        }
        ctx.finish();
        const double setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_setup0).count();
        vex::vector<double> x(ctx, N), y(ctx, N);
        // x is a function of the GLOBAL index (bench.py's hash): the job computes the same product for every D
        for (unsigned d = 0; d < D; ++d)
            vex::backend::check(vexhip_fill_hash(q[d].device_ordinal(), q[d].raw(), VEXHIP_F64, 42ull + (unsigned long long)part[d] * GOLD,
                        x(d).raw(), (int64_t)(part[d + 1] - part[d])));
        for (int i = 0; i < 40; ++i) y = A * x;
        ctx.finish();
        std::vector<void *> e0(D, nullptr), e1(D, nullptr);
        for (unsigned d = 0; d < D; ++d) {
            vex::backend::check(vexhip_event_create(q[d].device_ordinal(), 1, &e0[d]));
            vex::backend::check(vexhip_event_create(q[d].device_ordinal(), 1, &e1[d]));
            vex::backend::check(vexhip_event_record(q[d].device_ordinal(), e0[d], q[d].raw()));
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < M; ++i) y = A * x;
        const auto t_issued = std::chrono::steady_clock::now();
        for (unsigned d = 0; d < D; ++d) vex::backend::check(vexhip_event_record(q[d].device_ordinal(), e1[d], q[d].raw()));
        ctx.finish();
        const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / M;
        const double host_issue_us = std::chrono::duration<double, std::micro>(t_issued - t0).count() / M;
        double slowest = 0;
        std::vector<float> dev_ms(D);
        for (unsigned d = 0; d < D; ++d) {
            vex::backend::check(vexhip_event_elapsed_ms(q[d].device_ordinal(), e0[d], e1[d], &dev_ms[d]));
            dev_ms[d] /= M; slowest = std::max<double>(slowest, dev_ms[d]);
            vexhip_event_destroy(q[d].device_ordinal(), e0[d]); vexhip_event_destroy(q[d].device_ordinal(), e1[d]);
        }
        // phases of one product per device: median of 7
        std::vector<std::array<std::array<float, 4>, 7>> reps(D);
        for (int r = 0; r < 7; ++r) {
            std::vector<std::array<float, 4>> ms;
            A.apply_timed(x, y, ms);
            for (unsigned d = 0; d < D; ++d) reps[d][r] = ms[d];
        }
        vex::Reductor<double, vex::SUM_Kahan> sum(ctx);
        y = A * x;
        const double checksum = sum(y);
        // --check: the same matrix on ONE device (device 0 of the context), the same x: how many elements of y differ in their BITS
        // (the one-launch step keeps a row's entries in column order: 0; the split step adds the remote entries last: not 0)
        long long differing = -1;
        if (check_bits) {
            std::vector<double> ym(N), y1(N);
            vex::copy(y, ym);
            std::vector<vex::backend::context> c1(1, ctx.context(0));
            std::vector<vex::backend::command_queue> q1(1, q[0]);
            vex::Context one(c1, q1);
            const int dev = q[0].device_ordinal();
            vex::backend::device_vector<int> ptr(q[0], N + 1), col(q[0], nnz);
            vex::backend::device_vector<double> val(q[0], nnz);
            if (variable) vex::backend::check(vexhip_diffusion3d_strip_f64_i32(dev, q[0].raw(), n, 0, (int64_t)N, 7, ptr.raw(), col.raw(), val.raw()));
            else vex::backend::check(vexhip_poisson3d_csr_f64_i32(dev, q[0].raw(), n, ptr.raw(), col.raw(), val.raw()));
            vex::SpMat<double, int, int> A1(one.queue(), N, N, nnz, ptr, col, val);
            vex::vector<double> x1(one, N), yy(one, N);
            vex::backend::check(vexhip_fill_hash(dev, q[0].raw(), VEXHIP_F64, 42ull, x1(0).raw(), (int64_t)N));
            yy = A1 * x1;
            vex::copy(yy, y1);
            differing = 0;
            for (size_t i = 0; i < N; ++i) differing += std::memcmp(&ym[i], &y1[i], 8) != 0;
        }
        std::printf("{\"row\": \"vex::SpMat<double,int,int> y = A*x, %s 7-point %lld^3, ONE vex::Context x%u\", \"front_end\": \"C++ vexcl/spmat.hpp\", "
                    "\"step\": \"%s\", \"one_launch_step_declined\": \"%s\", \"elements_differing_from_one_device_product\": %lld, "
                    "\"devices\": %u, \"rows\": %zu, \"nnz\": %zu, \"setup_ms\": %.2f, \"ms\": %.5f, \"gflops\": %.1f, \"slowest_device_event_ms\": %.5f, "
                    "\"host_issue_us_per_product\": %.2f, \"csr_algorithmic_gbps\": %.1f, \"sum_y\": %.17g, \"per_device\": [",
                variable ? "variable-coefficient" : "Poisson", (long long)n, D, A.step_kind(), A.halo_declined().c_str(), differing,
                D, N, nnz, setup_ms, wall_ms, 2.0 * nnz / wall_ms / 1e6, slowest,
                host_issue_us, (12.0 * nnz + 4.0 * (N + 1) + 16.0 * N) / wall_ms / 1e6, checksum);
        for (unsigned d = 0; d < D; ++d) {
            std::array<float, 4> med;
            for (int k = 0; k < 4; ++k) { float v[7]; for (int r = 0; r < 7; ++r) v[r] = reps[d][r][k]; std::sort(v, v + 7); med[k] = v[3]; }
            const vexhip_spmat_info &info = A.storage_info(d);
            std::printf("%s{\"device\": %d, \"rows\": %zu, \"storage\": \"%s\", \"plane_product\": %d, \"event_ms\": %.5f, "
                        "\"step_ms\": {\"total\": %.5f, \"local\": %.5f, \"wait_for_ghosts\": %.5f, \"remote\": %.5f}}",
                    d ? ", " : "", q[d].device_ordinal(), part[d + 1] - part[d], storage_name(info), (int)info.plane.usable, dev_ms[d],
                    med[0], med[1], med[2], med[3]);
        }
        std::printf("]}\n");
        std::fflush(stdout);
    }
    return 0;
}

int main(int argc, char **argv) {
    int devices = -1;                       // -1: the single-device run
    bool check_bits = false;
    std::vector<char *> pos;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) { ++i; devices = !std::strcmp(argv[i], "all") ? 0 : std::atoi(argv[i]); }
        else if (!std::strcmp(argv[i], "--check")) check_bits = true;
        else pos.push_back(argv[i]);
    }
    const int64_t n = pos.size() > 0 ? std::atoll(pos[0]) : 512;
    const int M = pos.size() > 1 ? std::atoi(pos[1]) : 100;
    if (devices >= 0) return multi_device(n, M, devices, check_bits);

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
                    "\"storage\": \"%s\", \"kernel\": \"%s\", \"selection\": \"%s\", \"plane_product\": %d, \"rows\": %zu, \"nnz\": %zu, \"ms\": %.5f, \"gflops\": %.1f, "
                    "\"bytes_streamed\": %.0f, \"streamed_gbps\": %.1f, \"streamed_frac_of_8TBps\": %.4f, "
                    "\"csr_algorithmic_gbps\": %.1f, \"sum_y\": %.17g}\n",
                variable ? "variable-coefficient" : "Poisson", (long long)n, storage_name(info), info.product, info.reason, (int)info.plane.usable, N, nnz, ms, 2.0 * nnz / ms / 1e6,
                moved, moved / ms / 1e6, moved / ms / 1e6 / 8000.0, (12.0 * nnz + 4.0 * (N + 1) + 16.0 * N) / ms / 1e6, checksum);
        std::fflush(stdout);
    }
    return 0;
}
