#ifndef VEXCL_SORT_HPP
#define VEXCL_SORT_HPP
// vex::sort / vex::sort_by_key (reference: vexcl/sort.hpp:2158-2182, drivers
// :1716-1870, multi-device merge :1924-2116).  Per device: libvexhip's stable
// LSD radix sort; several devices: partitions are sorted on their GPUs, then
// merged on the host with a stable K-way merge and written back -- the
// reference's own multi-device strategy (sort.hpp:2081-2087).  Comparators:
// vex::less<T> (default) and vex::greater<T>; tuple keys and user functors are
// out of scope.
#include <algorithm>
#include <numeric>
#include "vector.hpp"
#include "scan.hpp"

namespace vex {

template <class T> struct less { bool operator()(const T &a, const T &b) const { return a < b; } static const bool descending = false; };
template <class T> struct greater { bool operator()(const T &a, const T &b) const { return a > b; } static const bool descending = true; };
template <class T> struct less_equal { bool operator()(const T &a, const T &b) const { return a <= b; } };
template <class T> struct greater_equal { bool operator()(const T &a, const T &b) const { return a >= b; } };

namespace detail {
    template <class K, class V>
    void sort_partition(const backend::command_queue &q, backend::device_vector<K> &keys, size_t n,
            backend::device_vector<V> *vals, bool descending)
    {
        if (n < 2) return;
        int dev = q.device_ordinal();
        // ping-pong arrays and workspace come from the context's scratch pool; the sort is
        // enqueued like any other operation (no finish): later work on this queue is ordered after it
        auto &pool = scratch_pool::instance();
        backend::device_vector<char> ktmp = pool.get(q, 0, n * sizeof(K));
        backend::device_vector<char> tmp = pool.get(q, 2, vexhip_sort_tmp_bytes(prim_dtype<K>::value, (int64_t)n));
        if (vals) {
            static_assert(sizeof(V) == 4 || sizeof(V) == 8, "sort_by_key values must be 4 or 8 bytes wide");
            backend::device_vector<char> vtmp = pool.get(q, 1, n * sizeof(V));
            backend::check(vexhip_sort(dev, q.raw(), prim_dtype<K>::value, descending ? 1 : 0, keys.raw(), ktmp.raw(),
                        (int)sizeof(V), vals->raw(), vtmp.raw(), (int64_t)n, tmp.raw()));
        } else {
            backend::check(vexhip_sort(dev, q.raw(), prim_dtype<K>::value, descending ? 1 : 0, keys.raw(), ktmp.raw(),
                        0, nullptr, nullptr, (int64_t)n, tmp.raw()));
        }
    }

    // stable merge of the sorted partitions on the host (sort.hpp:1924-1985)
    template <class K, class Comp>
    std::vector<size_t> merge_order(const std::vector<K> &keys, const std::vector<size_t> &part, Comp comp) {
        std::vector<size_t> order(keys.size());
        std::iota(order.begin(), order.end(), size_t(0));
        std::vector<size_t> bounds = part;            // k sorted runs <=> k + 1 boundaries
        while (bounds.size() > 2) {
            std::vector<size_t> next(1, bounds[0]);
            for (size_t i = 0; i + 2 < bounds.size(); i += 2) {
                std::inplace_merge(order.begin() + bounds[i], order.begin() + bounds[i + 1], order.begin() + bounds[i + 2],
                        [&](size_t a, size_t b) { return comp(keys[a], keys[b]); });
                next.push_back(bounds[i + 2]);
            }
            if (bounds.size() % 2 == 0) next.push_back(bounds.back());   // odd run count: last run carried over
            bounds.swap(next);
        }
        return order;
    }
}

/// Sorts the vector in place (sort.hpp:2158-2167).
template <class K, class Comp>
void sort(vector<K> &keys, Comp comp) {
    const auto &queue = keys.queue_list();
    for (unsigned d = 0; d < queue.size(); ++d)
        detail::sort_partition<K, int>(queue[d], keys(d), keys.part_size(d), nullptr, Comp::descending);
    if (queue.size() > 1) {
        std::vector<K> h(keys.size());
        copy(keys, h);
        auto order = detail::merge_order(h, keys.partition(), comp);
        std::vector<K> s(h.size());
        for (size_t i = 0; i < s.size(); ++i) s[i] = h[order[i]];
        copy(s, keys);
    }
}
template <class K> void sort(vector<K> &keys) { sort(keys, less<K>()); }

/// Sorts keys and carries the values along, stable (sort.hpp:2170-2182).
template <class K, class V, class Comp>
void sort_by_key(vector<K> &keys, vector<V> &vals, Comp comp) {
    precondition(keys.size() == vals.size() && keys.nparts() == vals.nparts(), "sort_by_key: incompatible vectors");
    const auto &queue = keys.queue_list();
    for (unsigned d = 0; d < queue.size(); ++d)
        detail::sort_partition<K, V>(queue[d], keys(d), keys.part_size(d), &vals(d), Comp::descending);
    if (queue.size() > 1) {
        std::vector<K> hk(keys.size()); std::vector<V> hv(vals.size());
        copy(keys, hk); copy(vals, hv);
        auto order = detail::merge_order(hk, keys.partition(), comp);
        std::vector<K> sk(hk.size()); std::vector<V> sv(hv.size());
        for (size_t i = 0; i < sk.size(); ++i) { sk[i] = hk[order[i]]; sv[i] = hv[order[i]]; }
        copy(sk, keys); copy(sv, vals);
    }
}
template <class K, class V> void sort_by_key(vector<K> &keys, vector<V> &vals) { sort_by_key(keys, vals, less<K>()); }

} // namespace vex
#endif
