#ifndef VEXCL_REDUCE_BY_KEY_HPP
#define VEXCL_REDUCE_BY_KEY_HPP
// vex::reduce_by_key (reference: vexcl/reduce_by_key.hpp:556-581 API, :65-552
// the five-kernel pipeline: offset calculation, scan of offsets, two scan-by-key
// kernels, final scatter; tests/reduce_by_key.cpp).
//
// MI355X design: ONE segmented scan (scan_by_key.hpp) whose triples also count the
// run heads, so the output slot of every run is known inside the scan itself: it
// stores -- at every run head -- the head's key and the finished sum of the run
// before it; the last element stores the last run.  The outputs must be sized first:
// round 3 counts the run heads with a keys-only kernel (4 of the 12 bytes per (int,
// double) element), reads the count back (4 bytes) and then runs the single-pass
// look-back scan -- keys twice, values once, nothing the size of the input written
// (the reference writes an offsets array and a scanned-values array of the input's
// size, reduce_by_key.hpp:470-500).  Value types the look-back does not carry (and
// VEXCL_SCAN_BY_KEY=tree) take the three phases with the count read after phase 2.
#include "scan_by_key.hpp"

namespace vex {
namespace detail {
namespace rbk {

// (an output that already has the right size on the right queue is kept: a solver that reduces the same keys every
// iteration does not allocate per call)
template <class Vec>
void size_output(Vec &v, const std::vector<backend::command_queue> &q, size_t n) {
    if (v.size() != n || v.nparts() != q.size() || (n && v.queue_list()[0].id() != q[0].id())) v.resize(q, n);
}
template <class OTuple, size_t... I>
void resize_outputs(const OTuple &okeys, const std::vector<backend::command_queue> &q, size_t n, std::index_sequence<I...>) {
    int dummy[] = {0, (size_output(std::get<I>(okeys), q, n), 0)...};
    (void)dummy;
}

template <class T> struct out_seq;
template <class K> struct out_seq<vector<K>> {
    typedef std::tuple<vector<K> &> tuple_type;
    static tuple_type get(vector<K> &k) { return tuple_type(k); }
};
template <class... K> struct out_seq<std::tuple<K...>> {
    typedef std::tuple<K...> tuple_type;
    static const tuple_type &get(const std::tuple<K...> &k) { return k; }
};

template <class IKTuple, class OKTuple, class V, class Comp, class Oper>
int reduce_by_key_sink(const IKTuple &ikeys, const vector<V> &ivals, const OKTuple &okeys, vector<V> &ovals, Comp comp, Oper oper) {
    static_assert(std::tuple_size<IKTuple>::value == std::tuple_size<OKTuple>::value, "input and output keys differ in number");
    typedef std::make_index_sequence<std::tuple_size<OKTuple>::value> seq;
    const auto &queue = ivals.queue_list();
    if (!ivals.size()) {
        resize_outputs(okeys, queue, 0, seq());
        ovals.resize(queue, 0);
        return 0;
    }
    // outputs that already have one common size on this queue: the likely number of runs (see sbk::run)
    size_t guess = (ovals.nparts() == queue.size() && ovals.size() && ovals.queue_list()[0].id() == queue[0].id()) ? ovals.size() : 0;
    sbk::for_each_key(okeys, [&](auto &o) {
        if (o.size() != guess || o.nparts() != queue.size() || (guess && o.queue_list()[0].id() != queue[0].id())) guess = 0;
    }, seq());
    return sbk::run<sbk::REDUCE>(ikeys, ivals, comp, oper,
            [&](backend::kernel &k, int count) {
                resize_outputs(okeys, queue, count, seq());
                size_output(ovals, queue, count);
                sbk::for_each_key(okeys, [&](auto &o) { k.push_arg(o(0).raw()); }, seq());
                k.push_arg(ovals(0).raw());
            }, true, false, guess);
}

} // namespace rbk
} // namespace detail

/// Reduces every run of equal (by comp) consecutive keys with oper; okeys and ovals are
/// resized to the number of runs, which is returned (reduce_by_key.hpp:556-567).
template <typename IKeys, typename OKeys, typename V, class Comp, class Oper>
int reduce_by_key(const IKeys &ikeys, const vector<V> &ivals, OKeys &&okeys, vector<V> &ovals, Comp comp, Oper oper) {
    return detail::rbk::reduce_by_key_sink(detail::sbk::key_seq<IKeys>::get(ikeys), ivals,
            detail::rbk::out_seq<typename std::decay<OKeys>::type>::get(okeys), ovals, comp, oper);
}

/// Keys compared with ==, values added (reduce_by_key.hpp:570-581).
template <typename K, typename V>
int reduce_by_key(const vector<K> &ikeys, const vector<V> &ivals, vector<K> &okeys, vector<V> &ovals) {
    return reduce_by_key(ikeys, ivals, okeys, ovals, detail::sbk::equal_fn<K>(), detail::sbk::plus_fn<V>());
}

} // namespace vex
#endif
