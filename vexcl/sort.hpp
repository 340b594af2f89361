#ifndef VEXCL_SORT_HPP
#define VEXCL_SORT_HPP
// vex::sort / vex::sort_by_key (reference: vexcl/sort.hpp:2158-2182, drivers
// :1716-1870, multi-device merge :1924-2116).
//
// Two engines per device:
//  * keys of one arithmetic type ordered by vex::less / vex::greater: libvexhip's stable
//    LSD radix sort (include/vexhip.h `vexhip_sort`).  One 4- or 8-byte value vector rides
//    along; any other set of values is permuted afterwards by the sorted positions.
//  * anything else -- a user comparator (VEX_DUAL_FUNCTOR), several key vectors tied
//    together -- a stable merge sort generated for the key types and the comparator
//    (detail::msort below; the reference's merge sort is sort.hpp:180-1700).
// Several devices: partitions are sorted on their GPUs, then merged on the host with a
// stable K-way merge and written back -- the reference's own multi-device strategy
// (sort.hpp:2081-2087).
#include <algorithm>
#include <numeric>
#include <tuple>
#include "vector.hpp"
#include "scan.hpp"
#include "function.hpp"
#include "vector_view.hpp"
#include "element_index.hpp"

namespace vex {

/// Comparators (sort.hpp:2120-2156).  `device` is the same predicate as a device function.
template <class T> struct less { VEX_DUAL_FUNCTOR(bool, (T, a)(T, b), return a < b;) static const bool descending = false; };
template <class T> struct greater { VEX_DUAL_FUNCTOR(bool, (T, a)(T, b), return a > b;) static const bool descending = true; };
template <class T> struct less_equal { VEX_DUAL_FUNCTOR(bool, (T, a)(T, b), return a <= b;) };
template <class T> struct greater_equal { VEX_DUAL_FUNCTOR(bool, (T, a)(T, b), return a >= b;) };

namespace detail {
    template <class K, class V>
    void radix_sort_partition(const backend::command_queue &q, backend::device_vector<K> &keys, size_t n,
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


namespace detail {
namespace msort {
// ---- stable merge sort for arbitrary keys and comparators ----------------------------------------
// Items are (key tuple, source position).  256 lanes x VT items per tile:
//   vexcl_msort_block      every lane sorts its VT items in registers (odd-even transposition, stable),
//                          then log2(256) rounds of merge-path merging through LDS: sorted tiles.
//   vexcl_msort_partition  per pass over runs of width W: where the diagonal of every output tile cuts
//                          the two runs it merges (binary search in global memory, one lane per tile).
//   vexcl_msort_merge      one workgroup per output tile: the two input pieces are staged in LDS,
//                          every lane finds its own cut and merges VT items; coalesced store.
// Ties take the element of the left run, so the sort is stable.  The comparator is the user's
// device function, called as `device(a_k0, a_k1, ..., b_k0, b_k1, ...)`.
static const int NT = 256;

template <class T> struct vector_value;
template <class T> struct vector_value<vector<T> &> { typedef T type; };
template <class T> struct vector_value<vector<T>> { typedef T type; };
template <class T> struct vector_value<const vector<T> &> { typedef T type; };

template <class Tuple, size_t... I>
std::vector<std::string> value_types(std::index_sequence<I...>) {
    return {type_name<typename vector_value<typename std::tuple_element<I, Tuple>::type>::type>()...};
}
template <class Tuple, size_t... I>
size_t value_bytes(std::index_sequence<I...>) {
    size_t b = 0;
    int dummy[] = {0, (b += sizeof(typename vector_value<typename std::tuple_element<I, Tuple>::type>::type), 0)...};
    (void)dummy;
    return b;
}

inline int items_per_lane(size_t key_bytes) {
    for (int vt = 8; vt > 1; vt /= 2) if ((size_t)NT * vt * (key_bytes + 4) <= 48 * 1024) return vt;
    return 1;
}

template <class DeviceComp>
std::string source(const backend::command_queue &q, const std::vector<std::string> &K, int vt) {
    backend::source_generator src(q);
    { gen_context c(src, q); DeviceComp::preamble(c); }
    const size_t nk = K.size();
    auto list = [&](const std::function<std::string(size_t)> &f, const std::string &sep) {
        std::string r; for (size_t k = 0; k < nk; ++k) r += (k ? sep : "") + f(k); return r; };
    auto num = [](size_t k) { return std::to_string(k); };
    std::ostringstream s;
    s << "\n#define NT " << NT << "\n#define VT " << vt << "\n#define TILE (NT * VT)\n#define MS_PAD 0xffffffffu\n";
    s << "struct ms_t { " << list([&](size_t k) { return K[k] + " k" + num(k) + ";"; }, " ") << " uint i; };\n";
    s << "__device__ inline bool ms_less(const ms_t &a, const ms_t &b) {\n"
         "    if (b.i == MS_PAD) return a.i != MS_PAD;\n"
         "    if (a.i == MS_PAD) return false;\n"
         "    return " << DeviceComp::name() << "(" << list([&](size_t k) { return "a.k" + num(k); }, ", ") << ", "
      << list([&](size_t k) { return "b.k" + num(k); }, ", ") << ");\n}\n";
    const std::string cptrs = list([&](size_t k) { return "const " + K[k] + " *p" + num(k); }, ", ") + ", const uint *pi";
    const std::string mptrs = list([&](size_t k) { return K[k] + " *p" + num(k); }, ", ") + ", uint *pi";
    s << "__device__ inline ms_t ms_get(" << cptrs << ", ulong at) { ms_t v; "
      << list([&](size_t k) { return "v.k" + num(k) + " = p" + num(k) + "[at];"; }, " ") << " v.i = pi[at]; return v; }\n";
    s << "__device__ inline void ms_put(" << mptrs << ", ulong at, const ms_t &v) { "
      << list([&](size_t k) { return "p" + num(k) + "[at] = v.k" + num(k) + ";"; }, " ") << " pi[at] = v.i; }\n";
    s << "__device__ inline ms_t ms_pad() { ms_t v; " << list([&](size_t k) { return "v.k" + num(k) + " = " + K[k] + "();"; }, " ")
      << " v.i = MS_PAD; return v; }\n";
    const std::string sargs = list([&](size_t k) { return "s" + num(k); }, ", ") + ", si";
    const std::string sdecl = list([&](size_t k) { return "__shared__ " + K[k] + " s" + num(k) + "[TILE];"; }, " ") + " __shared__ uint si[TILE];";
    // merge path: how many of the first `diag` merged items come from run A = [a0, a0 + na)
    s << "__device__ inline int ms_path(" << cptrs << ", ulong a0, long na, ulong b0, long nb, long diag) {\n"
         "    long lo = diag > nb ? diag - nb : 0, hi = diag < na ? diag : na;\n"
         "    while (lo < hi) {\n"
         "        const long mid = (lo + hi) >> 1;\n"
         "        const ms_t a = ms_get(" << list([&](size_t k) { return "p" + num(k); }, ", ") << ", pi, a0 + mid);\n"
         "        const ms_t b = ms_get(" << list([&](size_t k) { return "p" + num(k); }, ", ") << ", pi, b0 + diag - 1 - mid);\n"
         "        if (!ms_less(b, a)) lo = mid + 1; else hi = mid;\n"
         "    }\n"
         "    return (int)lo;\n}\n";
    // serial merge of VT items out of LDS, starting at the cut `lo` of diagonal `diag`
    s << "#define MS_MERGE(r, a0, na, b0, nb, diag, lo) { \\\n"
         "    int ai = (a0) + (lo), bi = (b0) + (diag) - (lo); const int ae = (a0) + (na), be = (b0) + (nb); \\\n"
         "    _Pragma(\"unroll\") for (int j = 0; j < VT; ++j) { \\\n"
         "        const bool ha = ai < ae, hb = bi < be; \\\n"
         "        const ms_t a = ms_get(" << sargs << ", ai < TILE ? ai : TILE - 1), b = ms_get(" << sargs << ", bi < TILE ? bi : TILE - 1); \\\n"
         "        const bool ta = ha && (!hb || !ms_less(b, a)); \\\n"
         "        r[j] = ta ? a : (hb ? b : ms_pad()); ai += ta; bi += (!ta && hb); } }\n";
    const std::string in_k = list([&](size_t k) { return "const " + K[k] + " *ik" + num(k); }, ", ");
    const std::string out = list([&](size_t k) { return K[k] + " *ok" + num(k); }, ", ") + ", uint *oi";
    const std::string in_args = list([&](size_t k) { return "ik" + num(k); }, ", ");
    const std::string out_args = list([&](size_t k) { return "ok" + num(k); }, ", ") + ", oi";

    s << "extern \"C\" __global__ __launch_bounds__(NT) void vexcl_msort_block(ulong n, " << in_k << ", " << out << ") {\n"
         "    " << sdecl << "\n"
         "    const int t = threadIdx.x;\n"
         "    const ulong tile0 = (ulong)blockIdx.x * TILE;\n"
         "    const int count = (int)(n - tile0 < (ulong)TILE ? n - tile0 : (ulong)TILE);\n"
         "    ms_t r[VT];\n"
         "    _Pragma(\"unroll\") for (int j = 0; j < VT; ++j) {\n"
         "        const int p = t * VT + j;\n"
         "        r[j] = ms_pad();\n"
         "        if (p < count) { " << list([&](size_t k) { return "r[j].k" + num(k) + " = ik" + num(k) + "[tile0 + p];"; }, " ") << " r[j].i = (uint)(tile0 + p); }\n"
         "    }\n"
         "    _Pragma(\"unroll\") for (int pass = 0; pass < VT; ++pass)\n"
         "        _Pragma(\"unroll\") for (int j = pass & 1; j + 1 < VT; j += 2)\n"
         "            if (ms_less(r[j + 1], r[j])) { const ms_t x = r[j]; r[j] = r[j + 1]; r[j + 1] = x; }\n"
         "    _Pragma(\"unroll\") for (int j = 0; j < VT; ++j) ms_put(" << sargs << ", t * VT + j, r[j]);\n"
         "    for (int w = VT; w < TILE; w <<= 1) {\n"
         "        __syncthreads();\n"
         "        const int o = t * VT, base = o & ~(2 * w - 1), diag = o - base;\n"
         "        const int lo = ms_path(" << sargs << ", base, w, base + w, w, diag);\n"
         "        MS_MERGE(r, base, w, base + w, w, diag, lo)\n"
         "        __syncthreads();\n"
         "        _Pragma(\"unroll\") for (int j = 0; j < VT; ++j) ms_put(" << sargs << ", o + j, r[j]);\n"
         "    }\n"
         "    __syncthreads();\n"
         "    for (int p = t; p < count; p += NT) ms_put(" << out_args << ", tile0 + p, ms_get(" << sargs << ", p));\n"
         "}\n";

    s << "extern \"C\" __global__ void vexcl_msort_partition(ulong n, ulong W, uint ntiles, " << in_k << ", const uint *ii, uint *mp) {\n"
         "    const uint p = blockIdx.x * blockDim.x + threadIdx.x;\n"
         "    if (p > ntiles) return;\n"
         "    const ulong g = (ulong)p * TILE < n ? (ulong)p * TILE : n;\n"
         "    const ulong base = g / (2 * W) * (2 * W);\n"
         "    const ulong a_end = base + W < n ? base + W : n, b_end = base + 2 * W < n ? base + 2 * W : n;\n"
         "    mp[p] = ms_path(" << in_args << ", ii, base, (long)(a_end - base), a_end, (long)(b_end - a_end), (long)(g - base));\n"
         "}\n";

    s << "extern \"C\" __global__ __launch_bounds__(NT) void vexcl_msort_merge(ulong n, ulong W, " << in_k << ", const uint *ii, " << out << ", const uint *mp) {\n"
         "    " << sdecl << "\n"
         "    const int t = threadIdx.x;\n"
         "    const ulong g0 = (ulong)blockIdx.x * TILE;\n"
         "    const ulong base = g0 / (2 * W) * (2 * W);\n"
         "    const ulong a_end = base + W < n ? base + W : n, b_end = base + 2 * W < n ? base + 2 * W : n;\n"
         "    const long na = (long)(a_end - base), nb = (long)(b_end - a_end);\n"
         "    const long d0 = (long)(g0 - base), d1 = d0 + TILE < na + nb ? d0 + TILE : na + nb;\n"
         "    const long a0 = mp[blockIdx.x], a1 = d1 == na + nb ? na : (long)mp[blockIdx.x + 1];\n"
         "    const long b0 = d0 - a0, b1 = d1 - a1;\n"
         "    const int ca = (int)(a1 - a0), cb = (int)(b1 - b0);\n"
         "    for (int p = t; p < ca; p += NT) ms_put(" << sargs << ", p, ms_get(" << in_args << ", ii, base + a0 + p));\n"
         "    for (int p = t; p < cb; p += NT) ms_put(" << sargs << ", ca + p, ms_get(" << in_args << ", ii, a_end + b0 + p));\n"
         "    __syncthreads();\n"
         "    const int diag = t * VT < ca + cb ? t * VT : ca + cb;\n"
         "    const int lo = ms_path(" << sargs << ", 0, ca, ca, cb, diag);\n"
         "    ms_t r[VT];\n"
         "    MS_MERGE(r, 0, ca, ca, cb, diag, lo)\n"
         "    __syncthreads();\n"
         "    _Pragma(\"unroll\") for (int j = 0; j < VT; ++j) if (t * VT + j < ca + cb) ms_put(" << sargs << ", t * VT + j, r[j]);\n"
         "    __syncthreads();\n"
         "    for (int p = t; p < ca + cb; p += NT) ms_put(" << out_args << ", base + d0 + p, ms_get(" << sargs << ", p));\n"
         "}\n";
    return src.str() + s.str();
}

struct kernels { backend::kernel block, partition, merge; int vt; };

template <class Tuple, class F, size_t... I>
void for_each(const Tuple &t, F &&f, std::index_sequence<I...>) {
    int dummy[] = {0, (f(std::get<I>(t), std::integral_constant<size_t, I>()), 0)...};
    (void)dummy;
}

template <class KTuple, size_t... I>
auto for_key_buffers(const backend::command_queue &q, const KTuple &, size_t n, std::index_sequence<I...>) {
    return std::make_tuple(backend::device_vector<typename vector_value<typename std::tuple_element<I, KTuple>::type>::type>(q, n)...);
}

/// Sorts partition d of the tied key vectors; returns the sorted source positions (for the values).
template <class KTuple, class Comp>
backend::device_vector<unsigned> sort_partition(const backend::command_queue &q, unsigned d, const KTuple &keys, Comp) {
    typedef std::make_index_sequence<std::tuple_size<KTuple>::value> seq;
    typedef typename std::decay<decltype(std::declval<Comp>().device)>::type device_comp;
    const size_t n = std::get<0>(keys).part_size(d);
    precondition(n < 0xffffffffull, "sort: a partition may hold at most 2^32 - 2 elements");
    static object_cache<kernels> cache;
    auto it = cache.find(q);
    if (it == cache.end()) {
        kernels k;
        k.vt = items_per_lane(value_bytes<KTuple>(seq()));
        backend::program prog = backend::build_sources(q, source<device_comp>(q, value_types<KTuple>(seq()), k.vt));
        k.block = backend::kernel(q, prog, "vexcl_msort_block");
        k.partition = backend::kernel(q, prog, "vexcl_msort_partition");
        k.merge = backend::kernel(q, prog, "vexcl_msort_merge");
        it = cache.insert(q, std::move(k));
    }
    kernels &K = it->second;
    const size_t tile = (size_t)NT * K.vt, ntiles = (n + tile - 1) / tile;

    // two sets of (keys..., positions); the block sort fills set 0, passes alternate
    auto bufs = std::make_tuple(for_key_buffers(q, keys, n, seq()), for_key_buffers(q, keys, n, seq()));
    backend::device_vector<unsigned> pos[2] = {backend::device_vector<unsigned>(q, n), backend::device_vector<unsigned>(q, n)};
    backend::device_vector<unsigned> mp(q, ntiles + 1);

    K.block.push_arg(n);
    for_each(keys, [&](const auto &k, auto) { K.block.push_arg(k(d).raw()); }, seq());
    for_each(std::get<0>(bufs), [&](const auto &b, auto) { K.block.push_arg(b.raw()); }, seq());
    K.block.push_arg(pos[0].raw());
    K.block.config(ntiles, NT);
    K.block(q);

    int cur = 0;
    for (size_t W = tile; W < n; W *= 2, cur ^= 1) {
        auto push_set = [&](backend::kernel &krn, int which, bool with_pos) {
            if (which == 0) for_each(std::get<0>(bufs), [&](const auto &b, auto) { krn.push_arg(b.raw()); }, seq());
            else            for_each(std::get<1>(bufs), [&](const auto &b, auto) { krn.push_arg(b.raw()); }, seq());
            if (with_pos) krn.push_arg(pos[which].raw());
        };
        K.partition.push_arg(n); K.partition.push_arg(W); K.partition.push_arg((unsigned)ntiles);
        push_set(K.partition, cur, true);
        K.partition.push_arg(mp.raw());
        K.partition.config((ntiles + 1 + 255) / 256, 256);
        K.partition(q);

        K.merge.push_arg(n); K.merge.push_arg(W);
        push_set(K.merge, cur, true);
        push_set(K.merge, cur ^ 1, true);
        K.merge.push_arg(mp.raw());
        K.merge.config(ntiles, NT);
        K.merge(q);
    }
    // sorted keys back into the user's vectors
    auto write_back = [&](const auto &set) {
        for_each(keys, [&](auto &k, auto I) {
            typedef typename std::decay<decltype(k)>::type::value_type T;
            vector<T> dst(q, k(d), n), from(q, std::get<decltype(I)::value>(set), n);
            dst = from;
        }, seq());
    };
    if (cur == 0) write_back(std::get<0>(bufs)); else write_back(std::get<1>(bufs));
    return pos[cur];
}

} // namespace msort

/// vals(d) = vals(d)[pos] for every tied value vector.
template <class VTuple, size_t... I>
void permute_partition(const backend::command_queue &q, unsigned d, const VTuple &vals, const backend::device_vector<unsigned> &pos,
        size_t n, std::index_sequence<I...>)
{
    vector<unsigned> where(q, pos, n);
    msort::for_each(vals, [&](auto &v, auto) {
        typedef typename std::decay<decltype(v)>::type::value_type T;
        vector<T> part(q, v(d), n), tmp(std::vector<backend::command_queue>(1, q), n);
        tmp = permutation(where)(part);
        part = tmp;
    }, std::index_sequence<I...>());
}
} // namespace detail

namespace detail {

template <class K, class Comp> struct radix_sortable : std::false_type {};
template <class K> struct radix_sortable<K, less<K>> : std::is_arithmetic<K> {};
template <class K> struct radix_sortable<K, greater<K>> : std::is_arithmetic<K> {};

template <class T> struct is_vex_vector : std::false_type {};
template <class T> struct is_vex_vector<vector<T>> : std::true_type {};
template <class Keys, class Comp> struct takes_radix_keys : std::false_type {};
template <class K, class Comp> struct takes_radix_keys<vector<K>, Comp> : radix_sortable<K, Comp> {};

// keys / values given as one vector or as a tuple of vector references
template <class T> struct tied;
template <class T> struct tied<vector<T>> {
    typedef std::tuple<vector<T> &> type;
    static type get(vector<T> &v) { return type(v); }
};
template <class... T> struct tied<std::tuple<T...>> {
    typedef std::tuple<T...> type;
    static const type &get(const std::tuple<T...> &t) { return t; }
};

template <class Comp, class HK, size_t... I>
bool host_less(Comp &comp, const HK &hk, size_t a, size_t b, std::index_sequence<I...>) {
    return comp(std::get<I>(hk)[a]..., std::get<I>(hk)[b]...);
}

template <class Tuple, size_t... I>
auto host_copies(const Tuple &t, std::index_sequence<I...>) {
    return std::make_tuple(std::vector<typename msort::vector_value<typename std::tuple_element<I, Tuple>::type>::type>(std::get<I>(t).size())...);
}

/// The general driver: tied keys, tied values (possibly none), any comparator.
template <class KTuple, class VTuple, class Comp>
void sort_tied(const KTuple &keys, const VTuple &vals, Comp comp) {
    typedef std::make_index_sequence<std::tuple_size<KTuple>::value> kseq;
    typedef std::make_index_sequence<std::tuple_size<VTuple>::value> vseq;
    auto &first = std::get<0>(keys);
    const auto &queue = first.queue_list();
    const size_t n = first.size();
    msort::for_each(keys, [&](const auto &k, auto) { precondition(k.size() == n && k.nparts() == queue.size(), "sort: tied keys differ in size"); }, kseq());
    msort::for_each(vals, [&](const auto &v, auto) { precondition(v.size() == n && v.nparts() == queue.size(), "sort_by_key: keys and values differ in size"); }, vseq());

    for (unsigned d = 0; d < queue.size(); ++d) {
        const size_t psize = first.part_size(d);
        if (psize < 2) continue;
        typedef typename msort::vector_value<typename std::tuple_element<0, KTuple>::type>::type K0;
        backend::device_vector<unsigned> pos;
        if constexpr (std::tuple_size<KTuple>::value == 1 && radix_sortable<K0, Comp>::value) {
            // one arithmetic key, natural order: the radix sort, carrying the source positions
            pos = backend::device_vector<unsigned>(queue[d], psize);
            vector<unsigned> iota(queue[d], pos, psize);
            iota = element_index();
            radix_sort_partition<K0, unsigned>(queue[d], first(d), psize, &pos, Comp::descending);
        } else {
            pos = msort::sort_partition(queue[d], d, keys, comp);
        }
        if (std::tuple_size<VTuple>::value) permute_partition(queue[d], d, vals, pos, psize, vseq());
    }
    if (queue.size() > 1) {
        // stable merge of the sorted partitions on the host (sort.hpp:1924-2116 of the reference)
        auto hk = host_copies(keys, kseq()); auto hv = host_copies(vals, vseq());
        msort::for_each(keys, [&](const auto &k, auto I) { copy(k, std::get<decltype(I)::value>(hk)); }, kseq());
        msort::for_each(vals, [&](const auto &v, auto I) { copy(v, std::get<decltype(I)::value>(hv)); }, vseq());
        std::vector<size_t> order(n);
        std::iota(order.begin(), order.end(), size_t(0));
        std::vector<size_t> bounds = first.partition();
        while (bounds.size() > 2) {
            std::vector<size_t> next(1, bounds[0]);
            for (size_t i = 0; i + 2 < bounds.size(); i += 2) {
                std::inplace_merge(order.begin() + bounds[i], order.begin() + bounds[i + 1], order.begin() + bounds[i + 2],
                        [&](size_t a, size_t b) { return host_less(comp, hk, a, b, kseq()); });
                next.push_back(bounds[i + 2]);
            }
            if (bounds.size() % 2 == 0) next.push_back(bounds.back());
            bounds.swap(next);
        }
        auto put_back = [&](auto &dev, auto &host) {
            typename std::decay<decltype(host)>::type sorted(n);
            for (size_t i = 0; i < n; ++i) sorted[i] = host[order[i]];
            copy(sorted, dev);
        };
        msort::for_each(keys, [&](auto &k, auto I) { put_back(k, std::get<decltype(I)::value>(hk)); }, kseq());
        msort::for_each(vals, [&](auto &v, auto I) { put_back(v, std::get<decltype(I)::value>(hv)); }, vseq());
    }
}

} // namespace detail

/// Sorts the vector in place (sort.hpp:2158-2167).
template <class K, class Comp>
typename std::enable_if<detail::radix_sortable<K, Comp>::value>::type
sort(vector<K> &keys, Comp comp) {
    const auto &queue = keys.queue_list();
    for (unsigned d = 0; d < queue.size(); ++d)
        detail::radix_sort_partition<K, int>(queue[d], keys(d), keys.part_size(d), nullptr, Comp::descending);
    if (queue.size() > 1) {
        std::vector<K> h(keys.size());
        copy(keys, h);
        auto order = detail::merge_order(h, keys.partition(), comp);
        std::vector<K> s(h.size());
        for (size_t i = 0; i < s.size(); ++i) s[i] = h[order[i]];
        copy(s, keys);
    }
}
/// ... by a user comparator, or several key vectors tied together (std::tie / boost::fusion::vector_tie in
/// the reference): `comp(a_k0, a_k1, ..., b_k0, b_k1, ...)`.
template <class Keys, class Comp>
typename std::enable_if<!detail::takes_radix_keys<typename std::decay<Keys>::type, Comp>::value>::type
sort(Keys &&keys, Comp comp) {
    detail::sort_tied(detail::tied<typename std::decay<Keys>::type>::get(keys), std::tuple<>(), comp);
}
template <class K> void sort(vector<K> &keys) { sort(keys, less<K>()); }

/// Sorts keys and carries the values along, stable (sort.hpp:2170-2182).
template <class K, class V, class Comp>
typename std::enable_if<detail::radix_sortable<K, Comp>::value && (sizeof(V) == 4 || sizeof(V) == 8)>::type
sort_by_key(vector<K> &keys, vector<V> &vals, Comp comp) {
    precondition(keys.size() == vals.size() && keys.nparts() == vals.nparts(), "sort_by_key: incompatible vectors");
    const auto &queue = keys.queue_list();
    for (unsigned d = 0; d < queue.size(); ++d)
        detail::radix_sort_partition<K, V>(queue[d], keys(d), keys.part_size(d), &vals(d), Comp::descending);
    if (queue.size() > 1) {
        std::vector<K> hk(keys.size()); std::vector<V> hv(vals.size());
        copy(keys, hk); copy(vals, hv);
        auto order = detail::merge_order(hk, keys.partition(), comp);
        std::vector<K> sk(hk.size()); std::vector<V> sv(hv.size());
        for (size_t i = 0; i < sk.size(); ++i) { sk[i] = hk[order[i]]; sv[i] = hv[order[i]]; }
        copy(sk, keys); copy(sv, vals);
    }
}
namespace detail {
template <class Keys, class Vals, class Comp, class Enable = void> struct takes_radix_pairs : std::false_type {};
template <class K, class V, class Comp>
struct takes_radix_pairs<vector<K>, vector<V>, Comp, void>
    : std::integral_constant<bool, radix_sortable<K, Comp>::value && (sizeof(V) == 4 || sizeof(V) == 8)> {};
}
/// ... any combination of tied keys, tied values and comparator.
template <class Keys, class Vals, class Comp>
typename std::enable_if<!detail::takes_radix_pairs<typename std::decay<Keys>::type, typename std::decay<Vals>::type, Comp>::value>::type
sort_by_key(Keys &&keys, Vals &&vals, Comp comp) {
    detail::sort_tied(detail::tied<typename std::decay<Keys>::type>::get(keys), detail::tied<typename std::decay<Vals>::type>::get(vals), comp);
}
template <class K, class V> void sort_by_key(vector<K> &keys, vector<V> &vals) { sort_by_key(keys, vals, less<K>()); }

} // namespace vex
#endif
