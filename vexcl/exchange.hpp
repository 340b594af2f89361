#ifndef VEXCL_EXCHANGE_HPP
#define VEXCL_EXCHANGE_HPP
// Ghost-value exchange between the devices of one context, shared by
// vex::sparse::distributed (reference: vexcl/sparse/distributed.hpp:347-428 --
// gather kernel, blocking D2H, host scatter, H2D).  Here: per consumer, each
// owner packs exactly the values that consumer needs (gather kernel) and the
// consumer pulls them with one peer copy per (owner, consumer) pair on a
// secondary queue: xGMI between GPUs, a device-to-device copy between logical
// devices of one GPU; no host hop, no finish().
#include <vector>
#include "backend.hpp"
#include "util.hpp"

namespace vex {
namespace detail {

template <class T>
class ghost_exchange {
    public:
        ghost_exchange() {}

        /// ghosts[d]: sorted global columns device d needs but does not own.
        template <class C>
        void setup(const std::vector<backend::command_queue> &q, const std::vector<size_t> &col_part,
                const std::vector<std::vector<C>> &ghosts)
        {
            queue = q;
            for (const auto &qq : q) squeue.push_back(backend::duplicate_queue(qq));
            const unsigned nd = static_cast<unsigned>(q.size());
            dev.resize(nd);
            std::vector<std::vector<int>> send_idx(nd);
            for (unsigned d = 0; d < nd; ++d) {
                const auto &g = ghosts[d];
                dev[d].nghost = g.size();
                size_t i = 0;
                while (i < g.size()) {
                    unsigned o = static_cast<unsigned>(column_owner(static_cast<size_t>(g[i]), col_part));
                    size_t j = i;
                    while (j < g.size() && static_cast<size_t>(g[j]) < col_part[o + 1]) ++j;
                    pair_t p; p.owner = o; p.consumer = d; p.send_off = send_idx[o].size(); p.recv_off = i; p.count = j - i;
                    for (size_t k = i; k < j; ++k) send_idx[o].push_back(static_cast<int>(static_cast<size_t>(g[k]) - col_part[o]));
                    pairs.push_back(p);
                    i = j;
                }
            }
            for (unsigned d = 0; d < nd; ++d) {
                dev[d].nsend = send_idx[d].size();
                if (dev[d].nsend) {
                    dev[d].send_idx = backend::device_vector<int>(q[d], send_idx[d].size(), send_idx[d].data());
                    dev[d].send_buf = backend::device_vector<T>(q[d], send_idx[d].size());
                }
                // never empty: kernels always receive a valid pointer
                dev[d].ghost_buf = backend::device_vector<T>(q[d], std::max<size_t>(1, dev[d].nghost));
            }
        }

        bool active() const { return !pairs.empty(); }
        size_t ghosts(unsigned d) const { return dev[d].nghost; }
        const backend::device_vector<T> &ghost_buffer(unsigned d) const { return dev[d].ghost_buf; }

        /// Packs and ships the ghosts of x; the primary queues wait for their arrival.
        template <class Parts>
        void run(const Parts &x) const {
            const unsigned nd = static_cast<unsigned>(queue.size());
            for (unsigned o = 0; o < nd; ++o)
                if (dev[o].nsend && !copies_done.empty()) backend::enqueue_barrier(queue[o], copies_done);
            std::vector<backend::event> packed(nd);
            for (unsigned o = 0; o < nd; ++o) {
                if (!dev[o].nsend) continue;
                backend::check(gather(queue[o].device_ordinal(), queue[o].raw(), (int64_t)dev[o].nsend,
                            dev[o].send_idx.raw(), x(o).raw(), dev[o].send_buf.raw()));
                packed[o] = backend::enqueue_marker(queue[o]);
            }
            copies_done.assign(nd, backend::event());
            for (const auto &p : pairs) {
                const backend::command_queue &sq = squeue[p.consumer];
                backend::enqueue_barrier(sq, backend::wait_list(1, packed[p.owner]));
                backend::check(vexhip_memcpy_peer(sq.device_ordinal(), dev[p.consumer].ghost_buf.raw() + p.recv_off,
                            queue[p.owner].device_ordinal(), dev[p.owner].send_buf.raw() + p.send_off,
                            p.count * sizeof(T), sq.raw()));
            }
            for (unsigned d = 0; d < nd; ++d)
                if (dev[d].nghost) {
                    copies_done[d] = backend::enqueue_marker(squeue[d]);
                    backend::enqueue_barrier(queue[d], backend::wait_list(1, copies_done[d]));
                }
        }
    private:
        struct pair_t { unsigned owner, consumer; size_t send_off, recv_off, count; };
        struct dev_t {
            backend::device_vector<int> send_idx; backend::device_vector<T> send_buf, ghost_buf;
            size_t nsend = 0, nghost = 0;
        };
        std::vector<backend::command_queue> queue, squeue;
        std::vector<pair_t> pairs;
        std::vector<dev_t> dev;
        mutable std::vector<backend::event> copies_done;

        static int gather(int d, void *s, int64_t n, const int *idx, const double *src, double *dst) { return vexhip_gather_f64_i32(d, s, n, idx, src, dst); }
        static int gather(int d, void *s, int64_t n, const int *idx, const float *src, float *dst) { return vexhip_gather_f32_i32(d, s, n, idx, src, dst); }
};

} // namespace detail
} // namespace vex
#endif
