// Secondary rows of BASELINE.md section 4 through the vex:: API, timed with
// HIP events on the compute queue:  C2 elementwise a = b*c + sin(d) (n = 1e8),
// reduce sum(a*b) (n = 2^24 and 1e8), inclusive scan and sort of 1e9 uint32
// keys.  Prints one JSON object per row: algorithmic GB/s and fraction of 8 TB/s.
// Usage: roofline [n_big = 1e9] [sections = "escpk"]   (e elementwise + reduce, s stencil, c SpMatCCSR, i inline SpMV terminals at 512^3,
// p scan + sort, k by-key primitives); bench.py runs section "e" for its elementwise / reduce rows, so that
// those rows come from the kernels the expression engine itself generates.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vexcl/vexcl.hpp>

struct timer {
    const vex::backend::command_queue &q; void *e0 = nullptr, *e1 = nullptr;
    explicit timer(const vex::backend::command_queue &q) : q(q) {
        vex::backend::check(vexhip_event_create(q.device_ordinal(), 1, &e0));
        vex::backend::check(vexhip_event_create(q.device_ordinal(), 1, &e1));
    }
    void start() { vex::backend::check(vexhip_event_record(q.device_ordinal(), e0, q.raw())); }
    double stop_ms() {
        vex::backend::check(vexhip_event_record(q.device_ordinal(), e1, q.raw()));
        vex::backend::check(vexhip_event_sync(q.device_ordinal(), e1));
        float ms = 0; vex::backend::check(vexhip_event_elapsed_ms(q.device_ordinal(), e0, e1, &ms));
        return ms;
    }
};

// JIT + ~30 ms of the same launches ahead of the timed ones: after >= 5 ms without work the first ~16 ms of launches run up to
// 12 % slow on this part (tools/r02_ramp.py), and every section starts behind host-side set-up
template <class F> static void warm(timer &t, F f) {
    f(); t.start(); f(); const double ms = t.stop_ms();
    const int k = (int)std::min(200.0, 30.0 / std::max(ms, 0.05));
    for (int i = 0; i < k; ++i) f();
}

static void report(const char *row, double n, double bytes_per_elem, double ms, const char *extra = "") {
    double gbps = n * bytes_per_elem / ms / 1e6;
    std::printf("{\"row\": \"%s\", \"n\": %.0f, \"ms\": %.4f, \"alg_gbps\": %.1f, \"frac_of_8TBps\": %.4f, \"elems_per_s\": %.4g%s}\n",
            row, n, ms, gbps, gbps / 8000.0, n / ms * 1e3, extra);
    std::fflush(stdout);
}

int main(int argc, char **argv) {
    size_t big = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1000000000ull;
    const std::string sections = argc > 2 ? argv[2] : "escpk";
    auto on = [&](char c) { return sections.find(c) != std::string::npos; };
    vex::Context ctx(vex::Filter::Env && vex::Filter::Count(1));
    if (!ctx) { std::cerr << "no device" << std::endl; return 1; }
    std::cout << ctx << std::endl;
    const vex::backend::command_queue &q = ctx.queue(0);
    timer t(q);
    const int reps = 20;
    if (on('e')) {   // C2
        const size_t n = 100000000;
        vex::vector<double> a(ctx, n), b(ctx, n), c(ctx, n), d(ctx, n);
        b = 0.5 + 1e-9 * vex::element_index(); c = 1.5; d = 1e-8 * vex::element_index();
        warm(t, [&] { a = b * c + sin(d); });
        t.start(); for (int i = 0; i < reps; ++i) a = b * c + sin(d); double ms = t.stop_ms() / reps;
        report("elementwise a=b*c+sin(d) f64", n, 32, ms);
        warm(t, [&] { a = b * c + d; });                                 // JIT outside the timed region
        t.start(); for (int i = 0; i < reps; ++i) a = b * c + d; ms = t.stop_ms() / reps;
        report("elementwise a=b*c+d f64", n, 32, ms);
        auto ta = vex::tag<1>(a);
        warm(t, [&] { ta = 0.5 * ta + b; });
        t.start(); for (int i = 0; i < reps; ++i) ta = 0.5 * ta + b; ms = t.stop_ms() / reps;
        report("saxpy a=alpha*a+b f64", n, 24, ms);
        vex::Reductor<double, vex::SUM> sum(ctx);
        double s = 0; warm(t, [&] { s += sum(a * b); });
        t.start(); for (int i = 0; i < reps; ++i) s += sum(a * b); ms = t.stop_ms() / reps;
        report("reduce sum(a*b) f64 n=1e8", n, 16, ms);
        (void)s;
    }
    if (on('e')) {
        const size_t n = 1 << 24;
        vex::vector<double> a(ctx, n), b(ctx, n); a = 1.0; b = 0.5;
        vex::Reductor<double, vex::SUM> sum(ctx);
        double s = sum(a * b);
        t.start(); for (int i = 0; i < 64; ++i) s += sum(a * b); double ms = t.stop_ms() / 64;
        report("reduce sum(a*b) f64 n=2^24 (incl. host readback)", n, 16, ms);
        (void)s;
    }
    if (on('s')) {   // next row: 21-point stencil convolution, LDS-staged
        const size_t n = 100000000;
        std::vector<double> S(21, 1.0 / 21);
        vex::stencil<double> s(ctx, S, 10);
        vex::vector<double> a(ctx, n), b(ctx, n);
        a = 1e-8 * vex::element_index();
        warm(t, [&] { b = a * s; });
        t.start(); for (int i = 0; i < reps; ++i) b = a * s; double ms = t.stop_ms() / reps;
        report("stencil b = a * s (21 points) f64", (double)n, 16, ms);
    }
    if (on('c')) {   // next row: the 512^3 Poisson operator as SpMatCCSR -- no (col, val) stream at all
        const size_t n = 512, N = n * n * n;
        const double h2i = (n - 1.0) * (n - 1.0);
        std::vector<size_t> idx(N), row = {0, 1, 8};
        std::vector<int> col = {0, -(int)(n * n), -(int)n, -1, 0, 1, (int)n, (int)(n * n)};
        std::vector<double> val = {1, -h2i, -h2i, -h2i, 6 * h2i, -h2i, -h2i, -h2i};
        for (size_t k = 0, p = 0; k < n; k++) for (size_t j = 0; j < n; j++) for (size_t i = 0; i < n; i++, p++)
            idx[p] = (i == 0 || i == n - 1 || j == 0 || j == n - 1 || k == 0 || k == n - 1) ? 0 : 1;
        vex::SpMatCCSR<double, int> A(q, N, 2, idx.data(), row.data(), col.data(), val.data());
        std::vector<vex::command_queue> q1(1, q);
        vex::vector<double> x(q1, N), y(q1, N);
        x = 1e-2 + 1e-9 * vex::element_index();
        if (const char *rpl = std::getenv("VEXHIP_CCSR_ROWS_PER_LANE")) vexhip_spmv_ccsr_set_rows_per_lane(std::atoi(rpl));   // A/B
        warm(t, [&] { y = A * x; });
        t.start(); for (int i = 0; i < reps; ++i) y = A * x; double ms = t.stop_ms() / reps;
        // the class hands the operator to the library's matrix object (value codes + slice dictionary: x and y once from HBM)
        // unless VEXCL_CCSR_KERNEL is set (its own kernel: 4 B of positions + x + y per row)
        const bool own = std::getenv("VEXCL_CCSR_KERNEL") != nullptr;
        report(own ? "SpMatCCSR y=A*x f64 512^3 (own kernel: 4 B idx + x + y per row)" : "SpMatCCSR y=A*x f64 512^3 (through vexhip_spmat: x + y per row)",
               (double)N, own ? 20 : 16, ms, ", \"equiv_csr_gflops\": 0");
        std::printf("{\"row\": \"SpMatCCSR vs CSR-algorithmic\", \"gflops\": %.1f, \"csr_equiv_gbps\": %.1f}\n",
                2.0 * 930123728.0 / ms / 1e6, 13845839300.0 / ms / 1e6);
    }
    if (on('i')) {   // SpMV fused INTO an expression kernel (spmat/inline_spmv.hpp:142-198, sparse/ell.hpp:207-267) at 512^3
        const long long n = 512;
        const size_t N = (size_t)(n * n * n), nnz = (size_t)vexhip_poisson3d_nnz(n);
        const int dev = q.device_ordinal();
        std::vector<vex::command_queue> q1(1, q);
        for (int variable = 0; variable < 2; ++variable) {
            vex::SpMat<double, int, int> A;
            {
                vex::backend::device_vector<int> ptr(q, N + 1), col(q, nnz);
                vex::backend::device_vector<double> val(q, nnz);
                if (variable) vex::backend::check(vexhip_diffusion3d_strip_f64_i32(dev, q.raw(), n, 0, (int64_t)N, 7, ptr.raw(), col.raw(), val.raw()));
                else vex::backend::check(vexhip_poisson3d_csr_f64_i32(dev, q.raw(), n, ptr.raw(), col.raw(), val.raw()));
                A = vex::SpMat<double, int, int>(q1, N, N, nnz, ptr, col, val);
            }
            vex::vector<double> x(q1, N), y(q1, N);
            vex::backend::check(vexhip_fill_hash(dev, q.raw(), VEXHIP_F64, 42, x(0).raw(), (int64_t)N));
            const vexhip_spmat_info &info = A.storage_info();
            const double moved = (double)info.matrix_bytes + 16.0 * N;            // stored matrix + x once + y once: what a product kernel moves
            char extra[512];
            // the product as its own kernel, then the same product as a terminal of a generated kernel (one row per lane, entries
            // looked up in the stored form -- class tables, code blocks or columns -- by the generated device function)
            warm(t, [&] { y = A * x; });
            t.start(); for (int i = 0; i < reps; ++i) y = A * x; double ms = t.stop_ms() / reps;
            std::snprintf(extra, sizeof extra, ", \"kernel\": \"%s\", \"gflops\": %.1f", info.plane.usable ? "sell8_plane_kernel" : "sell8_pair_kernel", 2.0 * nnz / ms / 1e6);
            report(variable ? "SpMat y = A*x, variable-coefficient 512^3 (library product)" : "SpMat y = A*x, Poisson 512^3 (library product)", 1.0, moved, ms, extra);
            // round 6: one vector + one product term is handed to the product whole (SpMat::apply_axpby: the plane product adds the vector in
            // its own pass; x itself costs no byte more than y = A*x); a residual reads b as well; any other expression around make_inline
            // runs the library product into a vector the matrix keeps, then the fused kernel
            const bool one_pass = vexhip_spmat_axpby_fused(A.storage_handle(), x(0).raw(), x(0).raw(), y(0).raw()) != 0;
            const std::string pks = std::string(info.product) + (one_pass ? ", the vector added in the same pass" : ": takes no addend -- the general route (y = z, y -= A*x; make_inline: the product into a kept vector + vexcl_vector_kernel)");
            const char *pk = pks.c_str();
            warm(t, [&] { y = x + 2 * vex::make_inline(A * x); });
            t.start(); for (int i = 0; i < reps; ++i) y = x + 2 * vex::make_inline(A * x); ms = t.stop_ms() / reps;
            std::snprintf(extra, sizeof extra, ", \"kernel\": \"%s\", \"gflops\": %.1f", pk, 2.0 * nnz / ms / 1e6);
            report(variable ? "y = x + 2 * make_inline(A*x), variable-coefficient 512^3 (one vector + one product)"
                            : "y = x + 2 * make_inline(A*x), Poisson 512^3 (one vector + one product)", 1.0, moved, ms, extra);
            {
                vex::vector<double> b(q1, N);
                vex::backend::check(vexhip_fill_hash(dev, q.raw(), VEXHIP_F64, 43, b(0).raw(), (int64_t)N));
                warm(t, [&] { y = b - A * x; });
                t.start(); for (int i = 0; i < reps; ++i) y = b - A * x; ms = t.stop_ms() / reps;
                std::snprintf(extra, sizeof extra, ", \"kernel\": \"%s\", \"gflops\": %.1f", pk, 2.0 * nnz / ms / 1e6);
                report(variable ? "r = b - A*x, variable-coefficient 512^3 (a residual: one vector + one product)"
                                : "r = b - A*x, Poisson 512^3 (a residual: one vector + one product)", 1.0, moved + 8.0 * N, ms, extra);
            }
            warm(t, [&] { y = x * vex::make_inline(A * x); });
            t.start(); for (int i = 0; i < reps; ++i) y = x * vex::make_inline(A * x); ms = t.stop_ms() / reps;
            std::snprintf(extra, sizeof extra, ", \"kernel\": \"library product into a kept vector + vexcl_vector_kernel (hiprtc)\", \"gflops\": %.1f", 2.0 * nnz / ms / 1e6);
            report(variable ? "y = x * make_inline(A*x), variable-coefficient 512^3 (any other expression around the product)"
                            : "y = x * make_inline(A*x), Poisson 512^3 (any other expression around the product)", 1.0, moved, ms, extra);
        }
    }
    if (on('i')) {   // sparse::ell (sparse/ell.hpp:207-267): its product IS an inline terminal of the expression kernel; built from host arrays, 256^3
        const size_t g = 256, N = g * g * g;
        std::vector<int> row(1, 0), col; std::vector<double> val;
        col.reserve(7 * N); val.reserve(7 * N); row.reserve(N + 1);
        const double h2i = (g - 1.0) * (g - 1.0);
        for (size_t k = 0, p = 0; k < g; ++k) for (size_t j = 0; j < g; ++j) for (size_t i = 0; i < g; ++i, ++p) {
            if (i == 0 || i == g - 1 || j == 0 || j == g - 1 || k == 0 || k == g - 1) { col.push_back((int)p); val.push_back(1.0); }
            else {
                const long off[7] = {-(long)(g * g), -(long)g, -1, 0, 1, (long)g, (long)(g * g)};
                for (int e = 0; e < 7; ++e) { col.push_back((int)((long)p + off[e])); val.push_back(e == 3 ? 6 * h2i : -h2i); }
            }
            row.push_back((int)col.size());
        }
        std::vector<vex::command_queue> q1(1, q);
        vex::sparse::ell<double> E(q1, N, N, row, col, val);
        vex::vector<double> x(q1, N), y(q1, N);
        x = 1e-2 + 1e-9 * vex::element_index();
        warm(t, [&] { y = E * x; });
        t.start(); for (int i = 0; i < reps; ++i) y = E * x; const double ms = t.stop_ms() / reps;
        char extra[256];
        std::snprintf(extra, sizeof extra, ", \"kernel\": \"vexcl_vector_kernel (hiprtc; ELL columns + values read by the generated device function)\", \"gflops\": %.1f", 2.0 * col.size() / ms / 1e6);
        report("y = E * x, sparse::ell<double> Poisson 256^3 (inline terminal; ELL width 7: 12 B per slot + x + y)", (double)N, 7 * 12 + 16, ms, extra);
    }
    if (on('p')) {   // C5 scan
        const size_t n = big;
        vex::vector<cl_uint> x(ctx, n), y(ctx, n);
        vex::backend::check(vexhip_fill_hash(q.device_ordinal(), q.raw(), VEXHIP_U32, 42, x(0).raw(), (int64_t)n));
        warm(t, [&] { vex::inclusive_scan(x, y); });
        t.start(); for (int i = 0; i < 5; ++i) vex::inclusive_scan(x, y); double ms = t.stop_ms() / 5;
        report("inclusive_scan u32", n, 8, ms);
        // C5 sort
        vex::sort(y); q.finish();
        double tot = 0;
        for (int i = 0; i < 3; ++i) {
            y = x; q.finish();
            t.start(); vex::sort(y); tot += t.stop_ms();
        }
        report("sort u32 keys", n, 8, tot / 3, ", \"note\": \"8-bit LSD radix: 4 passes x (hist read + scatter read + write)\"");
        vex::Reductor<size_t, vex::SUM> bad(ctx);
        vex::vector<cl_uint> z(ctx, n);
        z = y;
        // sortedness check on the device: count inversions between neighbours
        size_t inv = bad(vex::if_else(vex::permutation(vex::element_index(0, n - 1) + 1)(z) < vex::permutation(vex::element_index(0, n - 1))(z), 1, 0));
        std::printf("{\"row\": \"sort check\", \"inversions\": %zu}\n", inv);
    }
    if (on('k')) {   // by-key primitives: runs of ~64 equal keys, 1e8 (int key, double value) pairs
        const size_t n = 100000000;
        std::vector<vex::command_queue> q1(1, q);
        vex::vector<int> keys(q1, n), okeys;
        vex::vector<double> vals(q1, n), out(q1, n), ovals;
        keys = vex::element_index() / 64;
        vals = 1e-3 * (vex::element_index() % 1000);
        vex::inclusive_scan_by_key(keys, vals, out); q.finish();
        t.start(); for (int i = 0; i < 5; ++i) vex::inclusive_scan_by_key(keys, vals, out); double ms = t.stop_ms() / 5;
        report("inclusive_scan_by_key (int, f64) n=1e8: single pass, (4+8) B read + 8 B written", (double)n, 20, ms);
        setenv("VEXCL_SCAN_BY_KEY", "tree", 1);
        vex::inclusive_scan_by_key(keys, vals, out); q.finish();
        t.start(); for (int i = 0; i < 5; ++i) vex::inclusive_scan_by_key(keys, vals, out); ms = t.stop_ms() / 5;
        report("inclusive_scan_by_key (int, f64) n=1e8, VEXCL_SCAN_BY_KEY=tree: three phases, 2 x (4+8) B read + 8 B written", (double)n, 20, ms);
        unsetenv("VEXCL_SCAN_BY_KEY");
        int runs = vex::reduce_by_key(keys, vals, okeys, ovals); q.finish();
        t.start(); for (int i = 0; i < 5; ++i) runs = vex::reduce_by_key(keys, vals, okeys, ovals); ms = t.stop_ms() / 5;
        report("reduce_by_key (int, f64) n=1e8: one pass into the outputs of the previous call + 4-byte run count read back (first call: keys-only count first)", (double)n, 12, ms);
        std::printf("{\"row\": \"reduce_by_key runs\", \"runs\": %d}\n", runs);
        setenv("VEXCL_SCAN_BY_KEY", "tree", 1);
        runs = vex::reduce_by_key(keys, vals, okeys, ovals); q.finish();
        t.start(); for (int i = 0; i < 5; ++i) runs = vex::reduce_by_key(keys, vals, okeys, ovals); ms = t.stop_ms() / 5;
        report("reduce_by_key (int, f64) n=1e8, VEXCL_SCAN_BY_KEY=tree: three phases", (double)n, 12, ms);
        unsetenv("VEXCL_SCAN_BY_KEY");
        // runs whose carries cross many tiles of 16 Ki elements: the value of the look-back is folded serially from the nearest tile
        // with a run head (scan_by_key.hpp sbk_look_back: reproducible bits) -- what that costs where it has work to do
        for (size_t len : {size_t(1) << 20, n}) {
            keys = vex::element_index() / len;
            vex::inclusive_scan_by_key(keys, vals, out); q.finish();
            t.start(); for (int i = 0; i < 5; ++i) vex::inclusive_scan_by_key(keys, vals, out); ms = t.stop_ms() / 5;
            report(len == n ? "inclusive_scan_by_key (int, f64) n=1e8, ONE run (every carry crosses every tile in front of it)"
                            : "inclusive_scan_by_key (int, f64) n=1e8, runs of 2^20 elements (a carry crosses 64 tiles)", (double)n, 20, ms);
        }
    }
    return 0;
}
