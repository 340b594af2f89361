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

template <typename T>
class stencil {
    public:
        typedef T value_type;
        typedef T scalar_type;
        static_assert(std::is_same<T, double>::value || std::is_same<T, float>::value, "stencil value type must be float or double");

        stencil(const std::vector<backend::command_queue> &queue, const std::vector<T> &st, unsigned center)
            : queue(queue), lhalo((int)center), rhalo((int)st.size() - (int)center - 1) { init(st.begin(), st.end()); }
        template <class Iterator>
        stencil(const std::vector<backend::command_queue> &queue, Iterator begin, Iterator end, unsigned center)
            : queue(queue), lhalo((int)center), rhalo((int)(end - begin) - (int)center - 1) { init(begin, end); }
        stencil(const std::vector<backend::command_queue> &queue, std::initializer_list<T> list, unsigned center)
            : queue(queue), lhalo((int)center), rhalo((int)list.size() - (int)center - 1) { init(list.begin(), list.end()); }

        /// y = alpha * conv(x)   or   y += alpha * conv(x)   (stencil.hpp:428-457)
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
        std::vector<backend::command_queue> queue;
        int lhalo, rhalo;
        std::vector<backend::device_vector<T>> s, dbuf;

        template <class It> void init(It begin, It end) {
            precondition(begin != end && lhalo >= 0 && rhalo >= 0, "stencil: center must lie inside a non-empty stencil");
            std::vector<T> host(begin, end);
            for (const auto &q : queue) {
                s.push_back(backend::device_vector<T>(q, host.size(), host.data(), backend::MEM_READ_ONLY));
                dbuf.push_back(backend::device_vector<T>(q, host.size()));      // lhalo + rhalo (+1) values
            }
        }

        static int conv(int dev, void *st, int64_t n, int hl, int hr, int lh, int rh, const double *s, const double *x,
                const double *xr, double *y, double beta, double alpha) { return vexhip_stencil_conv_f64(dev, st, n, hl, hr, lh, rh, s, x, xr, y, beta, alpha); }
        static int conv(int dev, void *st, int64_t n, int hl, int hr, int lh, int rh, const float *s, const float *x,
                const float *xr, float *y, float beta, float alpha) { return vexhip_stencil_conv_f32(dev, st, n, hl, hr, lh, rh, s, x, xr, y, beta, alpha); }

        /// Fills dbuf[d] = { x[clamp(start_d - lhalo + k)] for k < lhalo } ++ { x[clamp(end_d + k)] for k < rhalo }
        /// with device-to-device copies of the runs owned by the other devices.
        void exchange_halos(const vex::vector<T> &x) const {
            const unsigned nd = static_cast<unsigned>(queue.size());
            if (nd <= 1 || lhalo + rhalo == 0) return;
            const std::vector<size_t> &part = x.partition();
            const long long N = (long long)x.size();
            std::vector<backend::event> ready(nd);                 // producers' pending writes to x
            for (unsigned o = 0; o < nd; ++o) if (x.part_size(o)) ready[o] = backend::enqueue_marker(queue[o]);
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
                    k += run - 1;
                }
            }
        }
};

/// x * s and s * x: the convolution as an additive term (stencil.hpp:472-485).
template <typename T>
detail::additive_operator<stencil<T>, vector<T>> operator*(const stencil<T> &s, const vector<T> &x) {
    return detail::additive_operator<stencil<T>, vector<T>>(s, x);
}
template <typename T>
detail::additive_operator<stencil<T>, vector<T>> operator*(const vector<T> &x, const stencil<T> &s) {
    return detail::additive_operator<stencil<T>, vector<T>>(s, x);
}

} // namespace vex
#endif
