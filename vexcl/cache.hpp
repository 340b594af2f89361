#ifndef VEXCL_CACHE_HPP
#define VEXCL_CACHE_HPP
// Kernel / object caches keyed by context (reference: vexcl/cache.hpp:49-183).
// Every cache registers itself so that purge_caches() / ~Context can drop the
// compiled kernels that belong to a context.
#include <map>
#include <mutex>
#include <set>
#include <functional>
#include "backend.hpp"

namespace vex {
namespace detail {

struct cache_register {
    static std::mutex &mx() { static std::mutex m; return m; }
    static std::set<struct cache_base *> &all() { static std::set<cache_base *> s; return s; }
};

struct cache_base {
    cache_base() { std::lock_guard<std::mutex> l(cache_register::mx()); cache_register::all().insert(this); }
    virtual ~cache_base() { std::lock_guard<std::mutex> l(cache_register::mx()); cache_register::all().erase(this); }
    virtual void purge(backend::context_id) = 0;
    virtual void clear() = 0;
};

/// Objects cached per context (kernel_cache) -- cache.hpp:118-162.
template <class Object>
class object_cache : public cache_base {
    public:
        typedef std::map<backend::context_id, Object> store_type;
        typedef typename store_type::iterator iterator;

        iterator end() { return store.end(); }
        iterator find(const backend::command_queue &q) {
            std::lock_guard<std::mutex> l(mx);
            return store.find(backend::get_context_id(q));
        }
        template <class Obj>
        iterator insert(const backend::command_queue &q, Obj &&o) {
            std::lock_guard<std::mutex> l(mx);
            return store.insert(std::make_pair(backend::get_context_id(q), std::forward<Obj>(o))).first;
        }
        void purge(backend::context_id id) override { std::lock_guard<std::mutex> l(mx); store.erase(id); }
        void clear() override { std::lock_guard<std::mutex> l(mx); store.clear(); }
    private:
        std::mutex mx;
        store_type store;
};

typedef object_cache<backend::kernel> kernel_cache;

/// Per-context scratch buffers for the primitives (sort ping-pong arrays, scan and
/// sort workspaces): grown on demand, reused across calls (hipMalloc / hipFree
/// cost more than a 16 Mi-key sort), dropped with the context like the kernels.
class scratch_pool : public cache_base {
    public:
        static scratch_pool &instance() { static scratch_pool p; return p; }
        /// At least `bytes` bytes, private to (queue, slot) until the next request for that slot.
        backend::device_vector<char> get(const backend::command_queue &q, unsigned slot, size_t bytes) {
            std::lock_guard<std::mutex> l(mx);
            // one set of buffers per QUEUE (two queues of a context may run primitives concurrently)
            auto &b = store[std::make_pair(backend::get_context_id(q), std::make_pair(q.id(), slot))];
            if (b.size() < bytes) b = backend::device_vector<char>(q, bytes + bytes / 8);
            return b;
        }
        void purge(backend::context_id id) override {
            std::lock_guard<std::mutex> l(mx);
            for (auto it = store.begin(); it != store.end();) if (it->first.first == id) it = store.erase(it); else ++it;
        }
        void clear() override { std::lock_guard<std::mutex> l(mx); store.clear(); }
    private:
        std::mutex mx;
        std::map<std::pair<backend::context_id, std::pair<size_t, unsigned>>, backend::device_vector<char>> store;
};

} // namespace detail

/// Drops everything cached for the context of the queue (cache.hpp:170-183).
inline void purge_caches(const backend::command_queue &q) {
    std::lock_guard<std::mutex> l(detail::cache_register::mx());
    for (auto c : detail::cache_register::all()) c->purge(backend::get_context_id(q));
}
inline void purge_caches(const std::vector<backend::command_queue> &queues) {
    for (const auto &q : queues) purge_caches(q);
}
inline void purge_kernel_caches(const std::vector<backend::command_queue> &q) { purge_caches(q); }

} // namespace vex
#endif
