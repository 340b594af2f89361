#ifndef VEXCL_FFT_HPP
#define VEXCL_FFT_HPP
// vex::FFT<Tin, Tout>: fast Fourier transform of vector expressions (reference: vexcl/fft.hpp:40-148,
// vexcl/fft/plan.hpp:46-412).
//
//   vex::FFT<cl_double2> fft(ctx, n);                  out = fft(in);          // 1-D, any n
//   vex::FFT<cl_double2> ifft(ctx, {h, w}, vex::fft::inverse);                 // n-D, row-major like FFTW
//   vex::FFT<cl_double2> batch(ctx, {b, n}, {vex::fft::none, vex::fft::forward});
//   vex::FFT<double, cl_double2> r2c(ctx, n);          // real input is extended with a zero imaginary part,
//   vex::FFT<cl_double2, double> c2r(ctx, n, inverse); // real output drops the imaginary part
//   y += fft(x * x) * 5;                               // the result is an ordinary vector expression
//
// The transform itself is the native plan of libvexhip (vexhip_fft_*: LDS row kernel, four-step for long rows,
// Bluestein for awkward lengths); this header evaluates the operand expression into the plan's complex input
// buffer with one fused kernel (r2c conversion included; a complex VECTOR operand is used where it is), runs the plan on the
// queue, and hands back `scale * c2r(out)` / `scl(out, scale)` as an expression over the output buffer, exactly
// the reference's shape (plan.hpp:336-357), so the 1/n of inverse transforms and any further arithmetic fuse
// into the consumer's kernel.  Single-device, as in the reference (plan.hpp:226-229).
#include <cstdlib>
#include <iostream>
#include <memory>
#include <numeric>
#include <queue>
#include <sstream>
#include <vector>
#include "vector.hpp"
#include "function.hpp"
#include "profiler.hpp"
#include "element_index.hpp"
#include "vector_pointer.hpp"

namespace vex {
namespace fft {

/// What happens along one dimension.
enum direction {
    forward = VEXHIP_FFT_FORWARD,   ///< forward transform
    inverse = VEXHIP_FFT_INVERSE,   ///< inverse transform (scaled by 1/n)
    none    = VEXHIP_FFT_NONE       ///< batch dimension
};

/// base^exponent, a factor of a transform length (fft/kernels.hpp:49-57).
struct pow {
    size_t base, exponent, value;
    pow(size_t b, size_t e) : base(b), exponent(e), value(1) { for (size_t i = 0; i < e; ++i) value *= b; }
};
inline std::ostream &operator<<(std::ostream &o, const pow &p) {
    o << p.base;
    if (p.exponent != 1) o << '^' << p.exponent;
    return o;
}

/// Successive primes, one per call (plan.hpp:56-78).
struct prime_generator {
    std::vector<size_t> found;
    size_t x;
    prime_generator() : x(2) {}
    size_t operator()() {
        for (;; ++x) {
            bool is_prime = true;
            for (size_t p : found) { if (p * p > x) break; if (x % p == 0) { is_prime = false; break; } }
            if (is_prime) { found.push_back(x); return x++; }
        }
    }
};

/// Prime factorization.
inline std::vector<pow> prime_factors(size_t n) {
    std::vector<pow> fs;
    if (n != 0) {
        prime_generator next;
        while (n != 1) {
            const size_t prime = next();
            size_t e = 0;
            while (n % prime == 0) { n /= prime; ++e; }
            if (e) fs.push_back(pow(prime, e));
        }
    }
    return fs;
}

/// Knows which lengths run fastest (plan.hpp:115-185).  The native plan handles every length; lengths of the
/// form 2^a 3^b 5^c 7^d stay inside the LDS row kernel.
struct planner {
    const size_t max_size;
    planner(size_t s = 25) : max_size(s) {}
    /// The size the data should be padded to.
    size_t best_size(size_t n) const { return vexhip_fft_best_size(n); }
};

template <class Tv, class Planner = planner>
struct plan {
    typedef typename cl_scalar_of<Tv>::type Ts;
    static_assert(std::is_same<Ts, cl_float>::value || std::is_same<Ts, cl_double>::value, "Only float and double data supported.");
    typedef typename cl_vector_of<Ts, 2>::type T2;

    VEX_FUNCTION_S(T2, r2c, (Ts, v), type_name<T2>() + " r = {v, 0}; return r;");
    VEX_FUNCTION_S(Ts, c2r, (T2, v), "return v.x;");
    VEX_FUNCTION_S(T2, scl, (T2, v)(Ts, s), "v.x *= s; v.y *= s; return v;");

    /// Value idx of the result for REAL input, read from the plan's output.  mode 0: the output holds the full
    /// transform, the value is z[idx] * scale.  mode 1 (forward transform of rows of even length n = 2 h): the rows were
    /// transformed as h complex numbers z = x[2k] + i x[2k+1]; X[k] = E[k] + W_n^k O[k] with E, O recovered from z[k] and
    /// conj(z[h-k]), and X[n-k] = conj(X[k]) -- half the data through the transform, the unpacking fused into whatever
    /// kernel consumes the result.
    VEX_FUNCTION_S(T2, rpost, (size_t, idx)(T2 *, z)(size_t, h)(int, mode)(Ts, scale),
        type_name<T2>() + " r;\n"
        "if (mode == 0) { r = z[idx]; r.x *= scale; r.y *= scale; return r; }\n"
        "const ulong n = 2 * h, row = idx / n, k = idx - row * n;\n"
        "const bool upper = k > h; const ulong kk = upper ? n - k : k;\n"
        + type_name<T2>() + " a = z[row * h + (kk == h ? 0 : kk)], b = z[row * h + (kk == 0 || kk == h ? 0 : h - kk)];\n"
        + type_name<Ts>() + " er = (a.x + b.x) / 2, ei = (a.y - b.y) / 2, pr = (a.x - b.x) / 2, pi = (a.y + b.y) / 2, sn, cs;\n"
        "const " + type_name<Ts>() + " ang = (" + type_name<Ts>() + ")(-2) * (" + type_name<Ts>() + ")kk / (" + type_name<Ts>() + ")n;\n"
        + std::string(std::is_same<Ts, cl_float>::value ? "sn = sinpif(ang); cs = cospif(ang);" : "sn = sinpi(ang); cs = cospi(ang);") + "\n"   // (sincospi's out-parameters cost scratch memory)
        "r.x = er + cs * pi + sn * pr; r.y = ei - cs * pr + sn * pi;\n"
        "if (upper) r.y = -r.y;\n"
        "r.x *= scale; r.y *= scale; return r;");

    std::vector<backend::command_queue> queues;
    Planner planner_;
    Ts scale;
    bool half;                               // real input through a half-length complex transform (see rpost)
    vex::vector<Ts> rbuf;                    // staging for real EXPRESSION operands of the half-length path
    const std::vector<size_t> sizes;
    std::vector<direction> dirs;
    std::vector<vex::vector<T2>> bufs;       // [0] input of the transform, [1] its output
    size_t input, output;
    profiler<> *profile;

    /// sizes: {n} in 1-D, {h, w} in 2-D (row-major: x + y * w, like FFTW), ...
    plan(const std::vector<backend::command_queue> &queues_, const std::vector<size_t> &sizes_,
         const std::vector<direction> &dirs_, const Planner &planner = Planner())
        : queues(queues_), planner_(planner), sizes(sizes_), dirs(dirs_), input(0), output(1), profile(nullptr)
    {
        precondition(!sizes.empty() && sizes.size() == dirs.size(), "FFT: one direction per dimension is required");
        precondition(queues.size() == 1, "FFT is only supported for single-device contexts.");
        const size_t total = std::accumulate(sizes.begin(), sizes.end(), size_t(1), std::multiplies<size_t>());
        size_t inv_n = 1;
        std::vector<int> d(sizes.size());
        for (size_t i = 0; i < sizes.size(); ++i) { d[i] = (int)dirs[i]; if (dirs[i] == inverse) inv_n *= sizes[i]; }
        scale = (Ts)1 / inv_n;
        // real input, forward along the (even) last dimension, every other dimension a batch: half-length transform
        half = cl_vector_length<Tv>::value == 1 && dirs.back() == forward && sizes.back() >= 2 && sizes.back() % 2 == 0
            && !std::getenv("VEXCL_FFT_NO_HALF");           // (measurement switch: full-length transform of the widened input)
        for (size_t i = 0; i + 1 < sizes.size(); ++i) if (dirs[i] != none) half = false;
        std::vector<size_t> native = sizes;
        if (half) native.back() /= 2;
        bufs.push_back(vex::vector<T2>());                   // input staging: allocated by the first operand that needs it
        bufs.push_back(vex::vector<T2>(queues, half ? total / 2 : total));
        void *p = nullptr;
        backend::check(vexhip_fft_plan_create(queues[0].device_ordinal(),
                std::is_same<Ts, cl_float>::value ? VEXHIP_F32 : VEXHIP_F64, (int)native.size(), native.data(), d.data(), &p));
        handle.reset(p, [](void *h) { vexhip_fft_plan_destroy(h); });
    }

    /// Evaluates `in` into the complex input buffer (real input: zero imaginary part) and runs the transform.
    template <class Expr>
    void transform(const Expr &in) {
        if (half) { transform_half(in); return; }
        if (profile) { profile->tic_cl(desc()); profile->tic_cl("in"); }
        vector<T2> &in_c = bufs[input];
        if (in_c.size() != bufs[output].size()) in_c = vex::vector<T2>(queues, bufs[output].size());
        assign_input(in_c, in, std::integral_constant<bool, cl_vector_length<Tv>::value == 1>());
        if (profile) { profile->toc("in"); profile->tic_cl("transform"); }
        if (in_c.size())
            backend::check(vexhip_fft_exec(handle.get(), queues[0].raw(), in_c(0).raw(), bufs[output](0).raw()));
        if (profile) { profile->toc("transform"); profile->toc(""); }
    }

    /// Real operand of the half-length path: a real vector IS the packed complex input; an expression is evaluated into
    /// a real staging buffer first.
    void transform_half(const vector<Ts> &in) { run_half(in); }
    template <class Expr> void transform_half(const Expr &in) {
        const size_t total = 2 * bufs[output].size();
        if (rbuf.size() != total) rbuf = vex::vector<Ts>(queues, total);
        rbuf = in;
        run_half(rbuf);
    }
    void run_half(const vector<Ts> &in) {
        precondition(in.nparts() == 1 && in.size() == 2 * bufs[output].size(), "FFT: the operand does not match the plan");
        if (profile) { profile->tic_cl(desc()); profile->tic_cl("transform"); }
        if (in.size())
            backend::check(vexhip_fft_exec(handle.get(), queues[0].raw(), in(0).raw(), bufs[output](0).raw()));
        if (profile) { profile->toc("transform"); profile->toc(""); }
    }

    /// A complex VECTOR operand is transformed where it is: no copy into the plan's input buffer.
    void transform(const vector<T2> &in) {
        precondition(!half, "FFT: a real-input plan was given a complex operand");
        precondition(in.nparts() == 1 && in.size() == bufs[output].size(), "FFT: the operand does not match the plan");
        if (profile) { profile->tic_cl(desc()); profile->tic_cl("transform"); }
        if (in.size())
            backend::check(vexhip_fft_exec(handle.get(), queues[0].raw(), in(0).raw(), bufs[output](0).raw()));
        if (profile) { profile->toc("transform"); profile->toc(""); }
    }

    static const bool real_input = cl_vector_length<Tv>::value == 1;

    template <typename Tout, class Expr>
    auto apply(const Expr &expr) -> typename std::enable_if<!real_input && cl_vector_length<Tout>::value == 1,
            decltype(std::declval<Ts>() * c2r(std::declval<vex::vector<T2> &>()))>::type
    {
        transform(expr);
        return scale * c2r(bufs[output]);
    }
    template <typename Tout, class Expr>
    auto apply(const Expr &expr) -> typename std::enable_if<!real_input && cl_vector_length<Tout>::value == 2,
            decltype(scl(std::declval<vex::vector<T2> &>(), std::declval<Ts>()))>::type
    {
        transform(expr);
        return scl(bufs[output], scale);
    }
    // real input: the result is read through rpost (which also covers the full-length case, mode 0)
    template <typename Tout, class Expr>
    auto apply(const Expr &expr) -> typename std::enable_if<real_input && cl_vector_length<Tout>::value == 2,
            decltype(rpost(vex::element_index(0, 0), vex::raw_pointer(std::declval<vex::vector<T2> &>()), size_t(), int(), std::declval<Ts>()))>::type
    {
        transform(expr);
        return rpost(vex::element_index(0, result_size()), vex::raw_pointer(bufs[output]), sizes.back() / 2, half ? 1 : 0, scale);
    }
    template <typename Tout, class Expr>
    auto apply(const Expr &expr) -> typename std::enable_if<real_input && cl_vector_length<Tout>::value == 1,
            decltype(c2r(rpost(vex::element_index(0, 0), vex::raw_pointer(std::declval<vex::vector<T2> &>()), size_t(), int(), std::declval<Ts>())))>::type
    {
        transform(expr);
        return c2r(rpost(vex::element_index(0, result_size()), vex::raw_pointer(bufs[output]), sizes.back() / 2, half ? 1 : 0, scale));
    }
    size_t result_size() const { return half ? 2 * bufs[output].size() : bufs[output].size(); }

    std::string desc() const {
        std::ostringstream o;
        o << "FFT(";
        for (auto n = sizes.begin(); n != sizes.end(); ++n) {
            if (n != sizes.begin()) o << " x ";
            o << *n;
            auto fs = prime_factors(*n);
            if (fs.size() > 1) {
                o << '=';
                for (auto f = fs.begin(); f != fs.end(); ++f) { if (f != fs.begin()) o << '*'; o << *f; }
            }
        }
        o << ")";
        return o.str();
    }

    /// Launches the plan consists of: (LDS row passes, transposes, other kernels).
    std::tuple<int, int, int> steps() const {
        int r = 0, t = 0, x = 0;
        backend::check(vexhip_fft_plan_steps(handle.get(), &r, &t, &x));
        return std::make_tuple(r, t, x);
    }

    private:
        std::shared_ptr<void> handle;

        template <class Expr> void assign_input(vector<T2> &in_c, const Expr &in, std::true_type) { in_c = r2c(in); }
        template <class Expr> void assign_input(vector<T2> &in_c, const Expr &in, std::false_type) { in_c = in; }
};

template <class T, class P>
inline std::ostream &operator<<(std::ostream &o, const plan<T, P> &p) {
    int r, t, x;
    std::tie(r, t, x) = p.steps();
    return o << p.desc() << "{\n  " << r << " LDS row pass(es), " << t << " transpose(s), " << x << " other launch(es)\n}";
}

} // namespace fft

/// Fast Fourier Transform.  Works on complex values internally; real input is extended with a zero imaginary
/// part, real output drops the imaginary part.
template <typename Tin, typename Tout = Tin, class Planner = fft::planner>
struct FFT {
    typedef typename cl_scalar_of<Tin>::type value_type;
    fft::plan<Tin, Planner> plan;

    /// 1-D
    FFT(const std::vector<backend::command_queue> &queues, size_t length, fft::direction dir = fft::forward, const Planner &planner = Planner())
        : plan(queues, std::vector<size_t>(1, length), std::vector<fft::direction>(1, dir), planner) {}
    FFT(size_t length, fft::direction dir = fft::forward, const Planner &planner = Planner())
        : plan(current_context().queue(), std::vector<size_t>(1, length), std::vector<fft::direction>(1, dir), planner) {}

    /// n-D, the same direction along every dimension
    FFT(const std::vector<backend::command_queue> &queues, const std::vector<size_t> &lengths, fft::direction dir = fft::forward, const Planner &planner = Planner())
        : plan(queues, lengths, std::vector<fft::direction>(lengths.size(), dir), planner) {}
    FFT(const std::vector<size_t> &lengths, fft::direction dir = fft::forward, const Planner &planner = Planner())
        : plan(current_context().queue(), lengths, std::vector<fft::direction>(lengths.size(), dir), planner) {}

    /// n-D, a direction per dimension (`none`: batch)
    FFT(const std::vector<backend::command_queue> &queues, const std::vector<size_t> &lengths, const std::vector<fft::direction> &dirs, const Planner &planner = Planner())
        : plan(queues, lengths, dirs, planner) {}
    FFT(const std::vector<size_t> &lengths, const std::vector<fft::direction> &dirs, const Planner &planner = Planner())
        : plan(current_context().queue(), lengths, dirs, planner) {}

    FFT(const std::vector<backend::command_queue> &queues, const std::initializer_list<size_t> &lengths, fft::direction dir = fft::forward, const Planner &planner = Planner())
        : plan(queues, lengths, std::vector<fft::direction>(lengths.size(), dir), planner) {}
    FFT(const std::initializer_list<size_t> &lengths, fft::direction dir = fft::forward, const Planner &planner = Planner())
        : plan(current_context().queue(), lengths, std::vector<fft::direction>(lengths.size(), dir), planner) {}
    FFT(const std::vector<backend::command_queue> &queues, const std::initializer_list<size_t> &lengths, const std::initializer_list<fft::direction> &dirs, const Planner &planner = Planner())
        : plan(queues, lengths, dirs, planner) {}
    FFT(const std::initializer_list<size_t> &lengths, const std::initializer_list<fft::direction> &dirs, const Planner &planner = Planner())
        : plan(current_context().queue(), lengths, dirs, planner) {}

    /// Performs the transform; the result is a vector expression.
    template <class Expr>
    auto operator()(const Expr &x) -> decltype(plan.template apply<Tout>(x)) { return plan.template apply<Tout>(x); }
};

} // namespace vex
#endif
