#ifndef VEXCL_STENCIL_HPP
#define VEXCL_STENCIL_HPP
// vex::stencil<T>: 1-D convolution of a partitioned vector with a small stencil
// (reference: vexcl/stencil.hpp:150-500).   y = x * s;  y += 42 * (x * s);
//     (x * s)[i] = sum_j s[j] * x[clamp(i + j - center, 0, n - 1)]
// The product is an additive term like A * x (stencil.hpp:472-485).  Kernel:
// libvexhip `vexhip_stencil_conv_*` (LDS-staged).  Between devices the halos
// (center values from the left neighbour, width-center-1 from the right) move
// with device-to-device copies ordered by events; the reference stages them
// through a host buffer with two finish() rounds (stencil.hpp:90-150).
#include <initializer_list>
#include <vector>
#include "operations.hpp"
#include "vector.hpp"

namespace vex {
template <class T, size_t N> class multivector;   // multivector.hpp

namespace detail {
/// What stencil<T> and StencilOperator share: the reach to either side and the per-device buffer
/// holding the neighbours' values (lhalo from the left, rhalo from the right).
template <typename T>
class stencil_halo {
    protected:
        stencil_halo(const std::vector<backend::command_queue> &queue, int lhalo, int rhalo)
            : queue(queue), lhalo(lhalo), rhalo(rhalo)
        {
            precondition(lhalo >= 0 && rhalo >= 0, "stencil: center must lie inside a non-empty stencil");
            for (const auto &q : this->queue) dbuf.push_back(backend::device_vector<T>(q, (size_t)(lhalo + rhalo + 1)));
        }
        std::vector<backend::command_queue> queue;
        int lhalo, rhalo;
        std::vector<backend::device_vector<T>> dbuf;

        void exchange_halos(const vex::vector<T> &x) const;
};
} // namespace detail

namespace detail {
/// Fills dbuf[d] = { x[clamp(start_d - lhalo + k)] for k < lhalo } ++ { x[clamp(end_d + k)] for k < rhalo }
/// with device-to-device copies of the runs owned by the other devices.
template <typename T>
void stencil_halo<T>::exchange_halos(const vex::vector<T> &x) const {
    const unsigned nd = static_cast<unsigned>(queue.size());
    if (nd <= 1 || lhalo + rhalo == 0) return;
    const std::vector<size_t> &part = x.partition();
    const long long N = (long long)x.size();
    std::vector<backend::event> ready(nd);                 // producers' pending writes to x
    for (unsigned o = 0; o < nd; ++o) if (x.part_size(o)) ready[o] = backend::enqueue_marker(queue[o]);
    std::vector<std::vector<unsigned>> read_from(nd);      // read_from[d]: owners whose segment device d copies from
    for (unsigned d = 0; d < nd; ++d) {
        if (!x.part_size(d)) continue;
        const long long start = (long long)part[d], end = (long long)part[d + 1];
        for (int k = 0; k < lhalo + rhalo; ++k) {
            long long g = k < lhalo ? start - lhalo + k : end + (k - lhalo);
            if ((k < lhalo && start == 0) || (k >= lhalo && end == N)) continue;   // edge of the whole vector: kernel clamps
            g = std::min(N - 1, std::max(0ll, g));
            unsigned o = static_cast<unsigned>(column_owner((size_t)g, part));
            // extend to a run of consecutive in-range positions on the same owner
            int run = 1;
            while (k + run < lhalo + rhalo && (k < lhalo) == (k + run < lhalo)) {
                long long g2 = (k + run) < lhalo ? start - lhalo + k + run : end + (k + run - lhalo);
                if (g2 != g + run || g2 >= (long long)part[o + 1] || g2 >= N) break;
                ++run;
            }
            backend::enqueue_barrier(queue[d], backend::wait_list(1, ready[o]));
            backend::check(vexhip_memcpy_peer(queue[d].device_ordinal(), dbuf[d].raw() + k,
                        queue[o].device_ordinal(), x(o).raw() + (g - (long long)part[o]), (size_t)run * sizeof(T), queue[d].raw()));
            if (o != d) read_from[d].push_back(o);
            k += run - 1;
        }
    }
    // the copies read x(o) on the CONSUMER's queue: a later write to x on queue[o] (`y = x * s; x = ...;` with no host
    // synchronisation) must wait for them -- the reference stages the halos through the host behind finish()
    for (unsigned d = 0; d < nd; ++d) {
        if (read_from[d].empty()) continue;
        backend::event copied = backend::enqueue_marker(queue[d]);
        for (unsigned o : read_from[d]) backend::enqueue_barrier(queue[o], backend::wait_list(1, copied));
    }
}
} // namespace detail

template <typename T>
class stencil : private detail::stencil_halo<T> {
        typedef detail::stencil_halo<T> halo;
        using halo::queue; using halo::lhalo; using halo::rhalo; using halo::dbuf; using halo::exchange_halos;
    public:
        typedef T value_type;
        typedef T scalar_type;
        static_assert(std::is_same<T, double>::value || std::is_same<T, float>::value, "stencil value type must be float or double");

        stencil(const std::vector<backend::command_queue> &queue, const std::vector<T> &st, unsigned center)
            : halo(queue, (int)center, (int)st.size() - (int)center - 1) { init(st.begin(), st.end()); }
        template <class Iterator>
        stencil(const std::vector<backend::command_queue> &queue, Iterator begin, Iterator end, unsigned center)
            : halo(queue, (int)center, (int)(end - begin) - (int)center - 1) { init(begin, end); }
        stencil(const std::vector<backend::command_queue> &queue, std::initializer_list<T> list, unsigned center)
            : halo(queue, (int)center, (int)list.size() - (int)center - 1) { init(list.begin(), list.end()); }

        /// y = alpha * conv(x)   or   y += alpha * conv(x)   (stencil.hpp:428-457)
        /// Component by component for multivectors (the left-hand sides arrive as a tuple of vectors).
        template <size_t N, class Target>
        void apply(const multivector<T, N> &x, Target &y, T alpha = 1, bool append = false) const {
            detail::tuple_for_each(y.v, [&](auto &yk, size_t k) { this->apply(x(k), yk, alpha, append); });
        }
        void apply(const vex::vector<T> &x, vex::vector<T> &y, T alpha = 1, bool append = false) const {
            precondition(x.size() == y.size() && x.nparts() == queue.size() && y.nparts() == queue.size(),
                    "stencil: incompatible vectors");
            exchange_halos(x);
            const unsigned nd = static_cast<unsigned>(queue.size());
            for (unsigned d = 0; d < nd; ++d) {
                size_t psize = x.part_size(d);
                if (!psize) continue;
                // a device has a neighbour if anything precedes / follows its segment
                int has_left = x.part_start(d) > 0, has_right = x.part_start(d) + psize < x.size();
                backend::check(conv(queue[d].device_ordinal(), queue[d].raw(), (int64_t)psize, has_left, has_right, lhalo, rhalo,
                            s[d].raw(), x(d).raw(), dbuf[d].raw(), y(d).raw(), append ? T(1) : T(0), alpha));
            }
        }
    private:
        std::vector<backend::device_vector<T>> s;

        template <class It> void init(It begin, It end) {
            precondition(begin != end, "stencil: center must lie inside a non-empty stencil");
            std::vector<T> host(begin, end);
            for (const auto &q : queue)
                s.push_back(backend::device_vector<T>(q, host.size(), host.data(), backend::MEM_READ_ONLY));
        }

        static int conv(int dev, void *st, int64_t n, int hl, int hr, int lh, int rh, const double *s, const double *x,
                const double *xr, double *y, double beta, double alpha) { return vexhip_stencil_conv_f64(dev, st, n, hl, hr, lh, rh, s, x, xr, y, beta, alpha); }
        static int conv(int dev, void *st, int64_t n, int hl, int hr, int lh, int rh, const float *s, const float *x,
                const float *xr, float *y, float beta, float alpha) { return vexhip_stencil_conv_f32(dev, st, n, hl, hr, lh, rh, s, x, xr, y, beta, alpha); }

};

/// X * s and s * X with X a multivector: every component is convolved (stencil.hpp:487-500;
/// tests/stencil.cpp:124).
template <typename T, size_t N>
detail::additive_operator<stencil<T>, multivector<T, N>> operator*(const stencil<T> &s, const multivector<T, N> &x) {
    return detail::additive_operator<stencil<T>, multivector<T, N>>(s, x);
}
template <typename T, size_t N>
detail::additive_operator<stencil<T>, multivector<T, N>> operator*(const multivector<T, N> &x, const stencil<T> &s) {
    return detail::additive_operator<stencil<T>, multivector<T, N>>(s, x);
}

/// x * s and s * x: the convolution as an additive term (stencil.hpp:472-485).
template <typename T>
detail::additive_operator<stencil<T>, vector<T>> operator*(const stencil<T> &s, const vector<T> &x) {
    return detail::additive_operator<stencil<T>, vector<T>>(s, x);
}
template <typename T>
detail::additive_operator<stencil<T>, vector<T>> operator*(const vector<T> &x, const stencil<T> &s) {
    return detail::additive_operator<stencil<T>, vector<T>>(s, x);
}

/// User-defined stencil operator (stencil.hpp:500-676 of the reference; tests/stencil.cpp:184-216):
///     VEX_STENCIL_OPERATOR(oscillate, double, 3, 1, "return sin(X[1] - X[0]) + sin(X[0] - X[-1]);", ctx);
///     y = oscillate(x);     y = 41 * oscillate(x) + oscillate(x);
/// The body sees `X[k]`, k in [-center, width - center), the values around the current element, the
/// ends of the whole vector repeated outwards as for stencil<T>.  One generated kernel per operator
/// type: every lane fills its window of `width` values in registers (neighbouring lanes read the same
/// cache lines, so x is fetched from HBM once) and calls the body as a device function.
template <typename T, unsigned width, unsigned center, class Impl>
class StencilOperator : private detail::stencil_halo<T> {
        typedef detail::stencil_halo<T> halo;
        using halo::queue; using halo::lhalo; using halo::rhalo; using halo::dbuf; using halo::exchange_halos;
        static_assert(center < width, "stencil operator: center must lie inside the window");
    public:
        typedef T value_type;
        typedef T scalar_type;

        StencilOperator(const std::vector<backend::command_queue> &queue) : halo(queue, (int)center, (int)width - (int)center - 1) {}

        detail::additive_operator<StencilOperator, vector<T>> operator()(const vector<T> &x) const {
            return detail::additive_operator<StencilOperator, vector<T>>(*this, x);
        }

        void apply(const vex::vector<T> &x, vex::vector<T> &y, T alpha = 1, bool append = false) const {
            precondition(x.size() == y.size() && x.nparts() == queue.size() && y.nparts() == queue.size(),
                    "stencil operator: incompatible vectors");
            static detail::kernel_cache cache;
            exchange_halos(x);
            for (unsigned d = 0; d < queue.size(); ++d) {
                const size_t psize = x.part_size(d);
                if (!psize) continue;
                auto kernel = cache.find(queue[d]);
                if (kernel == cache.end())
                    kernel = cache.insert(queue[d], backend::kernel(queue[d], source(queue[d]), "vexcl_stencil_operator"));
                backend::kernel &krn = kernel->second;
                const int has_left = x.part_start(d) > 0, has_right = x.part_start(d) + psize < x.size();
                krn.push_arg(psize); krn.push_arg(has_left); krn.push_arg(has_right);
                krn.push_arg(x(d)); krn.push_arg(dbuf[d]); krn.push_arg(y(d));
                krn.push_arg(append ? T(1) : T(0)); krn.push_arg(alpha);
                krn.config_streaming(queue[d], psize, 1);
                krn(queue[d]);
            }
        }
    private:
        static std::string source(const backend::command_queue &q) {
            const std::string V = type_name<T>();
            backend::source_generator src(q);
            src.template begin_function<T>("vexcl_stencil_body");
            src.begin_function_parameters();
            src.parameter("const " + V + " *", "X");
            src.end_function_parameters();
            src.new_line() << Impl::body();
            src.end_function();
            src.begin_kernel("vexcl_stencil_operator");
            src.begin_kernel_parameters();
            src.template parameter<size_t>("n");
            src.template parameter<int>("has_left");
            src.template parameter<int>("has_right");
            src.parameter("const " + V + " *", "x");
            src.parameter("const " + V + " *", "halo");
            src.parameter(V + " *", "y");
            src.template parameter<T>("beta");
            src.template parameter<T>("alpha");
            src.end_kernel_parameters();
            src.grid_stride_loop().open("{");
            src.new_line() << V << " window[" << width << "];";
            src.new_line() << "#pragma unroll";
            src.new_line() << "for(int k = 0; k < " << width << "; ++k)";
            src.open("{");
            src.new_line() << "const long pos = (long)idx + k - " << center << ";";
            src.new_line() << "window[k] = pos < 0 ? (has_left ? halo[" << center << " + pos] : x[0])";
            src.new_line() << "          : pos >= (long)n ? (has_right ? halo[" << center << " + pos - (long)n] : x[n - 1]) : x[pos];";
            src.close("}");
            src.new_line() << "const " << V << " r = alpha * vexcl_stencil_body(window + " << center << ");";
            src.new_line() << "y[idx] = beta != 0 ? beta * y[idx] + r : r;";
            src.close("}");
            src.end_kernel();
            return src.str();
        }
};

#define VEX_STENCIL_OPERATOR_TYPE(name, type, width, center, body_str)                                          \
    struct name : vex::StencilOperator<type, width, center, name> {                                             \
        name(const std::vector<vex::backend::command_queue> &q) : vex::StencilOperator<type, width, center, name>(q) {} \
        static std::string body() { return body_str; }                                                          \
    }

#define VEX_STENCIL_OPERATOR(name, type, width, center, body, queue)                                            \
    VEX_STENCIL_OPERATOR_TYPE(stencil_operator_##name##_t, type, width, center, body) const name(queue)

} // namespace vex
#endif
