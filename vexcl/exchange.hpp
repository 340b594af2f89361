#ifndef VEXCL_EXCHANGE_HPP
#define VEXCL_EXCHANGE_HPP
// Ghost-value exchange between the devices of one context, shared by vex::SpMat and
// vex::sparse::distributed (reference: vexcl/spmat.hpp:125-183 and sparse/distributed.hpp:347-428 --
// gather kernel, blocking D2H, host scatter, H2D, four finish() fences).  Here: every owner packs,
// per consumer, exactly the values that consumer needs (gather kernel on its primary queue), then
// ONE call ships everything on the secondary queues: vexhip_halo_exchange (include/vexhip.h) --
// grouped ncclSend / ncclRecv over xGMI when the context's devices are distinct GPUs, event-ordered
// device-to-device copies when logical devices share a GPU (the reference's test fixture) or when
// VEXCL_EXCHANGE=peer asks for it.  No host hop, no finish(); the local product overlaps the exchange.
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include "backend.hpp"
#include "util.hpp"

namespace vex {
namespace detail {

/// ONE communicator per (device list, transport), shared by every matrix, reductor and scan that works on those devices
/// (round 3: each SpMat / sparse::distributed / Reductor used to call ncclCommInitAll for itself -- seconds of set-up per
/// level of an AMG hierarchy, and several RCCL communicators driven from different streams on the same GPUs, which NCCL
/// documents as a deadlock hazard; with a single communicator its operations serialise in issue order).  The cache
/// holds weak references: the communicator goes away with its last user.
inline std::shared_ptr<vexhip_comm> make_comm(const std::vector<backend::command_queue> &q) {
    std::vector<int> key;
    for (const auto &qq : q) key.push_back(qq.device_ordinal());
    int transport = VEXHIP_COMM_AUTO;
    if (const char *e = std::getenv("VEXCL_EXCHANGE")) {
        if (!std::strcmp(e, "peer")) transport = VEXHIP_COMM_PEER;
        else if (!std::strcmp(e, "rccl")) transport = VEXHIP_COMM_RCCL;
    }
    const int ndev = static_cast<int>(key.size());
    key.push_back(transport);
    static std::mutex mx;
    static std::map<std::vector<int>, std::weak_ptr<vexhip_comm>> cache;
    std::lock_guard<std::mutex> lock(mx);
    auto it = cache.find(key);
    if (it != cache.end()) if (auto live = it->second.lock()) return live;
    vexhip_comm *c = nullptr;
    backend::check(vexhip_comm_init(ndev, key.data(), transport, &c));
    std::shared_ptr<vexhip_comm> made(c, [](vexhip_comm *p) { vexhip_comm_destroy(p); });
    cache[key] = made;
    return made;
}

template <class T>
class ghost_exchange {
    public:
        ghost_exchange() {}

        /// ghosts[d]: sorted global columns device d needs but does not own.
        template <class C>
        void setup(const std::vector<backend::command_queue> &q, const std::vector<size_t> &col_part,
                const std::vector<std::vector<C>> &ghosts)
        {
            static_assert(sizeof(T) % 4 == 0, "ghost values travel as 32-bit words");
            queue = q;
            for (const auto &qq : q) squeue.push_back(backend::duplicate_queue(qq));
            const unsigned nd = static_cast<unsigned>(q.size());
            dev.resize(nd);
            send_counts.assign(size_t(nd) * nd, 0); recv_counts.assign(size_t(nd) * nd, 0);
            std::vector<std::vector<int>> send_idx(nd);
            // consumers in rank order, so that owner o's send buffer is grouped by consumer in rank order; every
            // consumer's sorted ghost list splits into one contiguous run per owner (owners hold contiguous columns),
            // so what it receives, concatenated in owner order, IS its ghost vector (spmat.hpp:291-378)
            for (unsigned d = 0; d < nd; ++d) {
                const auto &g = ghosts[d];
                dev[d].nghost = g.size();
                size_t i = 0;
                while (i < g.size()) {
                    unsigned o = static_cast<unsigned>(column_owner(static_cast<size_t>(g[i]), col_part));
                    size_t j = i;
                    while (j < g.size() && static_cast<size_t>(g[j]) < col_part[o + 1]) ++j;
                    for (size_t k = i; k < j; ++k) send_idx[o].push_back(static_cast<int>(static_cast<size_t>(g[k]) - col_part[o]));
                    send_counts[size_t(o) * nd + d] += static_cast<int64_t>((j - i) * words);
                    recv_counts[size_t(d) * nd + o] += static_cast<int64_t>((j - i) * words);
                    any = true;
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
            if (any) {
                comm = make_comm(q);
                ev = std::make_shared<events>();
                ev->devs.resize(nd); ev->packed.assign(nd, nullptr); ev->shipped.assign(nd, nullptr);
                for (unsigned d = 0; d < nd; ++d) {
                    ev->devs[d] = q[d].device_ordinal();
                    backend::check(vexhip_event_create(ev->devs[d], 0, &ev->packed[d]));
                    backend::check(vexhip_event_create(ev->devs[d], 0, &ev->shipped[d]));
                }
            }
        }

        bool active() const { return any; }
        size_t ghosts(unsigned d) const { return dev[d].nghost; }
        const backend::device_vector<T> &ghost_buffer(unsigned d) const { return dev[d].ghost_buf; }

        /// Packs the ghosts of x and ships them on the secondary queues; returns without waiting.
        template <class Parts>
        void start(const Parts &x) const {
            const unsigned nd = static_cast<unsigned>(queue.size());
            // (1) the previous exchange must be done with the send buffers before they are packed again
            for (unsigned o = 0; o < nd; ++o)
                if (dev[o].nsend && ev->live) wait(queue[o], ev->shipped[o]);
            for (unsigned o = 0; o < nd; ++o) {
                if (!dev[o].nsend) continue;
                backend::check(gather(queue[o].device_ordinal(), queue[o].raw(), (int64_t)dev[o].nsend,
                            dev[o].send_idx.raw(), x(o).raw(), dev[o].send_buf.raw()));
            }
            // (2) a marker on every primary queue: behind it lie this product's pack AND the previous product's kernel
            // that still reads the ghost buffer (the reference fences with finish() at the top of every apply,
            // spmat.hpp:125-128); the secondary queue waits for it before anything lands in that buffer
            std::vector<const void *> sbuf(nd); std::vector<void *> rbuf(nd), str(nd);
            for (unsigned d = 0; d < nd; ++d) {
                record(queue[d], ev->packed[d]);
                wait(squeue[d], ev->packed[d]);
                sbuf[d] = dev[d].nsend ? dev[d].send_buf.raw() : nullptr;
                rbuf[d] = dev[d].ghost_buf.raw();
                str[d] = squeue[d].raw();
            }
            backend::check(vexhip_halo_exchange(comm.get(), VEXHIP_U32, sbuf.data(), send_counts.data(), rbuf.data(), recv_counts.data(), str.data()));
            for (unsigned d = 0; d < nd; ++d) record(squeue[d], ev->shipped[d]);
            ev->live = true;
        }

        /// The primary queue of device d waits for its ghosts.
        void finish(unsigned d) const { if (ev && ev->live) wait(queue[d], ev->shipped[d]); }

        /// start + finish on every device (sparse::distributed: the fused kernel needs the ghosts at once).
        template <class Parts>
        void run(const Parts &x) const {
            start(x);
            for (unsigned d = 0; d < queue.size(); ++d) finish(d);
        }
    private:
        static constexpr size_t words = sizeof(T) / 4;
        struct dev_t {
            backend::device_vector<int> send_idx; backend::device_vector<T> send_buf, ghost_buf;
            size_t nsend = 0, nghost = 0;
        };
        std::vector<backend::command_queue> queue, squeue;
        std::vector<dev_t> dev;
        std::vector<int64_t> send_counts, recv_counts;      // [device][peer], in 32-bit words
        std::shared_ptr<vexhip_comm> comm;
        bool any = false;
        // events re-recorded every product (created once: an event per marker would cost a create/destroy per step)
        struct events {
            std::vector<int> devs; std::vector<void *> packed, shipped; bool live = false;
            ~events() { for (size_t d = 0; d < devs.size(); ++d) { if (packed[d]) vexhip_event_destroy(devs[d], packed[d]); if (shipped[d]) vexhip_event_destroy(devs[d], shipped[d]); } }
        };
        std::shared_ptr<events> ev;
        static void record(const backend::command_queue &q, void *e) { backend::check(vexhip_event_record(q.device_ordinal(), e, q.raw())); }
        static void wait(const backend::command_queue &q, void *e) { backend::check(vexhip_stream_wait_event(q.device_ordinal(), q.raw(), e)); }

        static int gather(int d, void *s, int64_t n, const int *idx, const double *src, double *dst) { return vexhip_gather_f64_i32(d, s, n, idx, src, dst); }
        static int gather(int d, void *s, int64_t n, const int *idx, const float *src, float *dst) { return vexhip_gather_f32_i32(d, s, n, idx, src, dst); }
};

} // namespace detail
} // namespace vex
#endif
