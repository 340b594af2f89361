#ifndef VEXCL_SCAN_BY_KEY_HPP
#define VEXCL_SCAN_BY_KEY_HPP
// vex::inclusive_scan_by_key / vex::exclusive_scan_by_key (reference:
// vexcl/scan_by_key.hpp:713-757 API, :77-700 kernels; tests/scan_by_key.cpp) and
// the engine shared with vex::reduce_by_key (reduce_by_key.hpp).
//
// MI355X design.  A segmented scan over triples (c, f, v): c = heads seen,
// f = {bit 0: a head was seen, bit 1: non-empty}, v = running value of the LAST
// segment; combine(a, b) = (a.c + b.c, a.f | b.f, b has a head ? b.v : oper(a.v, b.v)).
// The operator is associative, so the scan is the usual three phases.  The tile
// geometry is fixed (it does not depend on the device or on n), so results are
// reproducible run to run; floating point sums are associated as a tree, not as
// the serial loop would, and differ from it by rounding only:
//   1. vexcl_sbk_reduce: one aggregate per tile of 4 waves x 2 rows x 256 elements;
//   2. vexcl_sbk_carry_local + vexcl_sbk_carry: the aggregates become carries-in
//      (groups of 1024 tiles scanned in parallel, then ONE workgroup over the groups);
//   3. vexcl_sbk_scan  : re-reads the tile, adds its carry, stores the result.
// Round 3: arithmetic value types of 4 or 8 bytes take ONE pass instead of the three phases -- vexcl_sbk_lookback, a
// decoupled look-back over the tile triples (the scheme of scan.hip: tiles handed out by an atomic ticket, every tile
// publishes its aggregate and later its inclusive prefix in 8-byte words that carry their state, predecessors are read
// 64 at a time by the tile's first wave and folded IN ORDER with wave shuffles): inputs are read once (12 + 8 instead of
// 24 + 8 bytes per (int, double) element).  reduce_by_key counts the run heads first (keys only) so that the outputs can
// be sized, then runs the same single pass.  Floating point (round 6): the carried value of a run that spans SEVERAL tiles
// is folded serially, oldest tile first, from the nearest tile that holds a run head (sbk_look_back below) -- the same
// association whichever predecessors happened to have published an inclusive prefix, so the bits are the same from run to run
// (until round 5 they were not: 60 of 100 million elements differed between two calls on runs that cross tiles,
// profiles/r06_bykey_reproducible.log).  VEXCL_SCAN_BY_KEY=tree keeps the three phases, with their own fixed association
// (vex::inclusive_scan / exclusive_scan with a user operator always use them).
// A lane owns FOUR CONSECUTIVE elements: it folds them serially (head flags come
// from its own previous element, the first one from the neighbour lane -- one
// extra load per row for lane 0), and the wave scans ONE aggregate per lane with
// wave-64 shuffles -- a quarter of the shuffle work of one element per lane (the
// first version, 1.30 ms per 1e8 (int, double) pairs, was bound by it), and no LDS
// ping-pong as in the reference's two-element-per-thread Hillis-Steele
// (scan_by_key.hpp:200-252).  A wave's 256-element row is contiguous in memory.  Keys may be a single vector or a std::tie of
// vectors (the reference takes boost::fusion::vector_tie); comparison and
// operator are VEX_FUNCTIONs pasted into the generated source.
#include <algorithm>
#include <cstdlib>
#include <sstream>
#include <tuple>
#include <type_traits>
#include "vector.hpp"
#include "function.hpp"

namespace vex {
namespace detail {
namespace sbk {

enum scan_mode { INCLUSIVE = 0, EXCLUSIVE = 1, REDUCE = 2 };

static const bool SBK_DPP_DEFAULT = true;        // round 5: the wave scan on DPP is the default (VEXCL_SBK_DPP=0: six shuffles per row) -- profiles/r05_sbk_sweep.log
static const int ITEMS = 4;         // consecutive elements per lane
static const int ROWS = 2;          // rows of 64 x ITEMS elements per wave
static const int WAVES = 4;         // waves per workgroup
static const int TILE = ROWS * ITEMS * WAVES * 64;

// ---- key sequences: one vector or a tuple of vector references -----------------
template <class T> struct key_seq;
template <class K> struct key_seq<vector<K>> {
    static const size_t size = 1;
    typedef std::tuple<const vector<K> &> tuple_type;
    static tuple_type get(const vector<K> &k) { return tuple_type(k); }
};
template <class... K> struct key_seq<std::tuple<K...>> {
    static const size_t size = sizeof...(K);
    typedef std::tuple<K...> tuple_type;
    static const tuple_type &get(const std::tuple<K...> &k) { return k; }
};
template <class V> struct key_value;
template <class K> struct key_value<vector<K>> { typedef K type; };

template <class Tuple, size_t... I>
std::vector<std::string> key_types(std::index_sequence<I...>) {
    return {type_name<typename key_value<typename std::decay<typename std::tuple_element<I, Tuple>::type>::type>::type>()...};
}

struct kernels {
    backend::kernel reduce, carry_local, carry, scan;
    backend::kernel lookback, count;       // single-pass form (valid when has_lookback)
    backend::kernel pipe;                  // pipelined single pass (valid when has_pipe: VEXCL_SBK_PIPELINE=1)
    bool has_lookback = false, has_pipe = false;
};

// geometry of the single-pass kernel: waves per workgroup, rows of 64 x ITEMS elements per wave (tuning: VEXCL_SBK_WAVES / _ROWS)
inline int lb_env(const char *name, int def) { const char *e = std::getenv(name); return e ? std::atoi(e) : def; }
static const int LB_WAVES = lb_env("VEXCL_SBK_WAVES", 16);
static const int LB_ROWS = lb_env("VEXCL_SBK_ROWS", 4);
template <class V> struct lookback_value { static const bool value = std::is_arithmetic<V>::value && (sizeof(V) == 4 || sizeof(V) == 8); };
// Round 4, not the default yet (VEXCL_SBK_PIPELINE=1): the single pass as a PIPELINE.  A workgroup keeps taking tiles; its worker
// waves hold the elements of the NEXT tile in flight while they work on this one, and one extra wave that holds no elements does
// the look-back (its polls are not queued behind a tile's worth of loads).  See vexcl_sbk_pipe in source().
inline bool pipe_enabled() { const char *e = std::getenv("VEXCL_SBK_PIPELINE"); return e && std::atoi(e) != 0; }
inline int pipe_waves() { const int w = lb_env("VEXCL_SBK_PIPE_WAVES", 7); return w < 1 ? 1 : (w > 15 ? 15 : w); }
inline int pipe_rows() { const int r = lb_env("VEXCL_SBK_PIPE_ROWS", 4); return r < 1 ? 1 : (r > 8 ? 8 : r); }
// VEXCL_SBK_DPP=1 (round 4): the wave-level segmented scan of the single pass moves its values with DPP (row_shr 1/2/4/8, then
// row_bcast:15 and row_bcast:31) instead of six ds_bpermute round trips through the LDS per row
inline bool dpp_enabled() { const char *e = std::getenv("VEXCL_SBK_DPP"); return e ? std::atoi(e) != 0 : SBK_DPP_DEFAULT; }
inline bool lookback_enabled() { const char *e = std::getenv("VEXCL_SCAN_BY_KEY"); return !(e && std::string(e) == "tree"); }

/// vexcl_sbk_pipe: the single pass of vexcl_sbk_lookback as a pipeline (uses the helpers that kernel's text defines).
/// The one-tile kernel spends a tile's time twice: ~9 us streaming its 196 KB, then as long again in wave scans, two barriers, the
/// look-back and the second pass, with nothing else resident on the CU (104 registers x 16 waves).  Here a workgroup of PW worker
/// waves + one scan wave takes tile after tile (tickets, two ahead):
///   step:  workers: rows of THIS tile (elements requested one step ago) -> aggregates -> request the NEXT tile's elements
///          barrier;  scan wave: fold the aggregates, publish, look back, publish the inclusive prefix;  barrier
///          workers: second pass over this tile's values, stores;  barrier
/// The next tile's loads stay in flight through all of it: the barriers are bare s_barrier behind an LDS-only wait (a fence at
/// workgroup scope would drain vmcnt, i.e. wait for the loads), and the wave that polls the predecessors' status words holds no
/// elements -- on gfx950 loads return in order, a poll behind 12 KB of requests would wait for all of them.
template <class V, class Comp, class Oper>
void pipe_source(std::ostringstream &s, const std::vector<std::string> &K, scan_mode mode) {
    const size_t nk = K.size();
    const int PW = pipe_waves();
    auto key_params = [&]() { std::ostringstream p; for (size_t k = 0; k < nk; ++k) p << "const " << K[k] << " *key" << k << ", "; return p.str(); };
    auto key_args = [&]() { std::ostringstream p; for (size_t k = 0; k < nk; ++k) p << "key" << k << ", "; return p.str(); };
    s << "\n#define PW " << PW << "\n#define PLBR " << pipe_rows() << "\n"
         "#define SBK_BAR() asm volatile(\"s_waitcnt lgkmcnt(0)\\n\\ts_barrier\" ::: \"memory\")\n"
         "struct sbk_in {\n";
    for (size_t k = 0; k < nk; ++k) s << "  " << K[k] << " k" << k << "[PLBR][ITEMS], e" << k << "[PLBR];\n";      // e: lane 0's key before the row's first element
    s << "  val_t v[PLBR][ITEMS];\n"
         "};\n"
         // complete rows only (a scalar test: no divergent arms whose merge would wait for the loads just issued); the rows of the
         // ragged last tile are loaded where they are used (sbk_fetch_ragged)
         "__device__ inline void sbk_fetch(sbk_in &x, ulong n, ulong wbase, int lane, " << key_params() << "const val_t *vals) {\n"
         "  #pragma unroll\n"
         "  for (int r = 0; r < PLBR; ++r) {\n"
         "    const ulong row0 = wbase + (ulong)r * (64 * ITEMS), i0 = row0 + (ulong)lane * ITEMS;\n"
         "    if (row0 + 64 * ITEMS <= n) {\n"
         "      #pragma unroll\n"
         "      for (int j = 0; j < ITEMS; ++j) {\n";
    for (size_t k = 0; k < nk; ++k) s << "        x.k" << k << "[r][j] = key" << k << "[i0 + j];\n";
    s << "        x.v[r][j] = vals[i0 + j];\n"
         "      }\n"
         "    }\n";
    for (size_t k = 0; k < nk; ++k)
        s << "    x.e" << k << "[r] = (" << K[k] << ")0;\n"
             "    if (lane == 0 && i0 > 0 && i0 < n) x.e" << k << "[r] = key" << k << "[i0 - 1];\n";
    s << "  }\n"
         "}\n"
         "__device__ inline void sbk_fetch_ragged(sbk_in &x, int r, ulong n, ulong i0, " << key_params() << "const val_t *vals) {\n"
         "  #pragma unroll\n"
         "  for (int j = 0; j < ITEMS; ++j) {\n"
         "    const bool in = i0 + j < n;\n";
    for (size_t k = 0; k < nk; ++k) s << "    x.k" << k << "[r][j] = in ? key" << k << "[i0 + j] : (" << K[k] << ")0;\n";
    s << "    x.v[r][j] = in ? vals[i0 + j] : val_t();\n"
         "  }\n"
         "}\n"
         "extern \"C\" __global__ void __launch_bounds__(" << (PW + 1) * 64 << ") vexcl_sbk_pipe(ulong n, long nt, " << key_params()
      << "const val_t *vals, sbk_word *ws, ";
    if (mode == REDUCE) {
        for (size_t k = 0; k < nk; ++k) s << K[k] << " *okey" << k << ", ";
        s << "val_t *ovals, int cap) {\n";
    } else {
        s << "val_t *ovals, val_t init) {\n";
    }
    s << "  __shared__ sbk_t agg[PW];\n"
         "  __shared__ val_t s_x[sizeof(val_t) == 8 ? PW : 1][ITEMS * 64] __attribute__((aligned(16)));\n"
         "  __shared__ long s_tile[4];\n"           // ring: s_tile[i & 3] = ticket of this workgroup's i-th tile
         "  __shared__ sbk_t s_pre;\n"
         "  sbk_word *status = ws + 2;\n"
         "  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n"      // (a scalar: the tests on a wave's rows are scalar branches)
         "  const bool scanner = wave == PW;\n"
         "  const unsigned long long below = (1ull << lane) - 1ull;\n"
         "  if (threadIdx.x == PW * 64) { s_tile[0] = (long)atomicAdd(&ws[0], 1ull); s_tile[1] = (long)atomicAdd(&ws[0], 1ull); }\n"
         "  SBK_BAR();\n"
         "  if (s_tile[0] >= nt) return;\n"
         "  sbk_in A, B;\n"
         "  if (!scanner) sbk_fetch(A, n, ((ulong)s_tile[0] * PW + wave) * (PLBR * ITEMS * 64), lane, " << key_args() << "vals);\n"
         // one step: `cur` holds this tile's elements (requested a step ago), `nxt` receives the next tile's; false: that was the last tile
         "  auto step = [&](sbk_in &cur, sbk_in &nxt, long it) -> bool {\n"
         "    const long tile = s_tile[it & 3], next = s_tile[(it + 1) & 3];\n"
         "    const ulong wbase = ((ulong)tile * PW + (scanner ? 0 : wave)) * (PLBR * ITEMS * 64);\n"
         "    sbk_t pre[PLBR];\n"
         "    unsigned heads = 0;\n"
         "    if (scanner) {\n"
         "      if (lane == 0) s_tile[(it + 2) & 3] = (long)atomicAdd(&ws[0], 1ull);\n"      // the ticket after next: read two barriers later
         "    } else {\n"
         "      sbk_t carry = sbk_empty();\n"
         "      #pragma unroll\n"
         "      for (int r = 0; r < PLBR; ++r) {\n"
         "        const ulong row0 = wbase + (ulong)r * (64 * ITEMS), i0 = row0 + (ulong)lane * ITEMS;\n"
         "        unsigned h = 0, ok = 0;\n"
         "        if (row0 + 64 * ITEMS <= n) ok = (1u << ITEMS) - 1u;\n"
         "        else {\n"
         "          sbk_fetch_ragged(cur, r, n, i0, " << key_args() << "vals);\n"
         "          #pragma unroll\n"
         "          for (int j = 0; j < ITEMS; ++j) ok |= (unsigned)(i0 + j < n) << j;\n"
         "        }\n";
    for (size_t k = 0; k < nk; ++k)
        s << "        " << K[k] << " p" << k << " = __shfl_up(cur.k" << k << "[r][ITEMS - 1], 1, 64);\n"
          << "        if (lane == 0) p" << k << " = cur.e" << k << "[r];\n";
    s << "        val_t tail = val_t();\n"
         "        #pragma unroll\n"
         "        for (int j = 0; j < ITEMS; ++j) {\n";
    for (size_t k = 0; k < nk; ++k)
        s << "          const " << K[k] << " pk" << k << " = j ? cur.k" << k << "[r][j ? j - 1 : 0] : p" << k << ";\n";
    s << "          const bool hd = ((ok >> j) & 1u) && ((i0 + j == 0) || !" << Comp::name() << "(";
    for (size_t k = 0; k < nk; ++k) s << "pk" << k << ", ";
    for (size_t k = 0; k < nk; ++k) s << "cur.k" << k << "[r][j]" << (k + 1 < nk ? ", " : "");
    s << "));\n"
         "          h |= (unsigned)hd << j;\n"
         "          if ((ok >> j) & 1u) tail = (hd || j == 0) ? cur.v[r][j] : " << Oper::name() << "(tail, cur.v[r][j]);\n"
         "        }\n"
         "        heads |= h << (r * ITEMS);\n"
         "        const unsigned long long H = __ballot(h != 0u), any = __ballot(ok != 0u);\n"
         "        const unsigned long long upto = H & (below | (1ull << lane));\n"
         "        const int hl = upto ? 63 - __builtin_clzll(upto) : 0;\n"
         "        val_t T = tail;\n"
         "        #pragma unroll\n"
         "        for (int o = 1; o < 64; o <<= 1) {\n"
         "          const val_t u = __shfl_up(T, o, 64);\n"
         "          if (lane - o >= hl && ((any >> lane) & 1ull)) T = " << Oper::name() << "(u, T);\n"
         "        }\n"
         "        int cb = 0;\n"
         "        #pragma unroll\n"
         "        for (int j = 0; j < ITEMS; ++j) cb += __popcll(__ballot((h >> j) & 1u) & below);\n"
         "        sbk_t p; p.c = cb; p.f = ((any & below) ? 2 : 0) | ((H & below) ? 1 : 0); p.v = __shfl_up(T, 1, 64);\n"
         "        if (lane == 0) p = sbk_empty();\n"
         "        pre[r] = sbk_combine(carry, p);\n"
         "        sbk_t ra; ra.c = 0;\n"
         "        #pragma unroll\n"
         "        for (int j = 0; j < ITEMS; ++j) ra.c += __popcll(__ballot((h >> j) & 1u));\n"
         "        ra.f = (any ? 2 : 0) | (H ? 1 : 0);\n"
         "        ra.v = __shfl(T, any ? 63 - __builtin_clzll(any) : 0, 64);\n"
         "        carry = sbk_combine(carry, ra);\n"
         "      }\n"
         "      if (lane == 0) agg[wave] = carry;\n"
         "      if (next < nt) sbk_fetch(nxt, n, ((ulong)next * PW + wave) * (PLBR * ITEMS * 64), lane, " << key_args() << "vals);\n"
         "    }\n"
         "    SBK_BAR();\n"
         "    if (scanner) {\n"
         "      sbk_t t = agg[0];\n"
         "      for (int w = 1; w < PW; ++w) t = sbk_combine(t, agg[w]);\n"
         "      if (lane == 0) sbk_publish(status, tile, t, tile == 0 ? 2u : 1u);\n"
         "      const sbk_t excl = sbk_look_back(status, tile, lane);\n"
         "      if (lane == 0) {\n"
         "        if (tile > 0) sbk_publish(status, tile, sbk_combine(excl, t), 2u);\n"
         "        s_pre = excl;\n"
         "      }\n"
         "    }\n"
         "    SBK_BAR();\n"
         "    if (!scanner) {\n"
         "      sbk_t W = s_pre;\n"
         "      for (int w = 0; w < wave; ++w) W = sbk_combine(W, agg[w]);\n"
         "      #pragma unroll\n"
         "      for (int r = 0; r < PLBR; ++r) {\n"
         "        const ulong row0 = wbase + (ulong)r * (64 * ITEMS), i0 = row0 + (ulong)lane * ITEMS;\n"
         "        sbk_t prev = sbk_combine(W, pre[r]);\n";
    if (mode != REDUCE) {
        s << "        if (row0 + 64 * ITEMS <= n && ((ulong)ovals & 15) == 0) {\n"
             "          val_t out[ITEMS];\n"
             "          #pragma unroll\n"
             "          for (int j = 0; j < ITEMS; ++j) {\n"
             "            const bool head = (heads >> (r * ITEMS + j)) & 1u;\n"
             "            sbk_t x; x.c = head; x.f = 2 | (int)head; x.v = cur.v[r][j];\n"
             "            const sbk_t fin = sbk_combine(prev, x);\n";
        if (mode == INCLUSIVE) s << "            (void)init; out[j] = fin.v;\n";
        else                   s << "            out[j] = head ? init : " << Oper::name() << "(init, prev.v);\n";
        s << "            prev = fin;\n"
             "          }\n"
             "          typedef val_t sbk_vecp __attribute__((ext_vector_type(16 / sizeof(val_t))));\n"
             "          #define NQ ((int)(ITEMS * sizeof(val_t) / 16))\n"
             "          sbk_vecp o[NQ];\n"
             "          #pragma unroll\n"
             "          for (int q = 0; q < NQ; ++q)\n"
             "            #pragma unroll\n"
             "            for (int e = 0; e < (int)(16 / sizeof(val_t)); ++e) o[q][e] = out[q * (16 / sizeof(val_t)) + e];\n"
             "          if (NQ == 1) { ((sbk_vecp *)(ovals + row0))[lane] = o[0]; continue; }\n"
             "          sbk_vecp *sx = (sbk_vecp *)s_x[wave];\n"
             "          #pragma unroll\n"
             "          for (int q = 0; q < NQ; ++q) sx[lane * NQ + q] = o[q];\n"
             "          __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n"
             "          #pragma unroll\n"
             "          for (int q = 0; q < NQ; ++q) ((sbk_vecp *)(ovals + row0))[q * 64 + lane] = sx[q * 64 + lane];\n"
             "          __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier();\n"
             "          #undef NQ\n"
             "          continue;\n"
             "        }\n";
    }
    s << "        #pragma unroll\n"
         "        for (int j = 0; j < ITEMS; ++j) {\n"
         "          const ulong i = i0 + j;\n"
         "          if (i < n) {\n"
         "            const bool head = (heads >> (r * ITEMS + j)) & 1u;\n"
         "            sbk_t x; x.c = head; x.f = 2 | (int)head; x.v = cur.v[r][j];\n"
         "            const sbk_t fin = sbk_combine(prev, x);\n";
    if (mode == INCLUSIVE) {
        s << "            (void)init; ovals[i] = fin.v;\n";
    } else if (mode == EXCLUSIVE) {
        s << "            ovals[i] = head ? init : " << Oper::name() << "(init, prev.v);\n";
    } else {
        s << "            if (head && fin.c <= cap) {\n";
        for (size_t k = 0; k < nk; ++k) s << "              okey" << k << "[fin.c - 1] = cur.k" << k << "[r][j];\n";       // (the key is in a register: no second read)
        s << "              if (fin.c > 1) ovals[fin.c - 2] = prev.v;\n"
             "            }\n"
             "            if (i == n - 1) { if (fin.c <= cap) ovals[fin.c - 1] = fin.v; ((int *)(ws + 1))[0] = fin.c; }\n";
    }
    s << "            prev = fin;\n"
         "          }\n"
         "        }\n"
         "      }\n"
         "    }\n"
         "    SBK_BAR();\n"                        // agg, s_pre and the ticket slots are free for the next step
         "    return next < nt;\n"
         "  };\n"
         "  for (long it = 0;; it += 2) {\n"
         "    if (!step(A, B, it)) break;\n"
         "    if (!step(B, A, it + 1)) break;\n"
         "  }\n"
         "}\n";
}

/// Source of the three kernels for the given key types, value type, functions and mode.
template <class V, class Comp, class Oper>
std::string source(const backend::command_queue &q, const std::vector<std::string> &K, scan_mode mode) {
    backend::source_generator src(q);
    { gen_context c(src, q); Comp::preamble(c); Oper::preamble(c); }
    const std::string T = type_name<V>();
    const size_t nk = K.size();
    std::ostringstream s;
    s << "\n#define ROWS " << ROWS << "\n#define ITEMS " << ITEMS << "\n#define WAVES " << WAVES << "\n";
    s << "typedef " << T << " val_t;\n"
         "struct sbk_t { int c; int f; val_t v; };\n"
         "__device__ inline sbk_t sbk_empty() { sbk_t r; r.c = 0; r.f = 0; r.v = val_t(); return r; }\n"
         "__device__ inline sbk_t sbk_combine(sbk_t a, sbk_t b) {\n"
         "  sbk_t r; r.c = a.c + b.c; r.f = a.f | b.f;\n"
         "  if ((b.f & 1) || !(a.f & 2)) r.v = b.v;\n"
         "  else if (b.f & 2) r.v = " << Oper::name() << "(a.v, b.v);\n"
         "  else r.v = a.v;\n"
         "  return r;\n"
         "}\n"
         "__device__ inline sbk_t sbk_up(sbk_t x, int o) { sbk_t r; r.c = __shfl_up(x.c, o, 64); r.f = __shfl_up(x.f, o, 64); r.v = __shfl_up(x.v, o, 64); return r; }\n"
         "__device__ inline sbk_t sbk_from(sbk_t x, int l) { sbk_t r; r.c = __shfl(x.c, l, 64); r.f = __shfl(x.f, l, 64); r.v = __shfl(x.v, l, 64); return r; }\n"
         "__device__ inline sbk_t sbk_wave_scan(sbk_t x, int lane) {\n"
         "  for (int o = 1; o < 64; o <<= 1) { sbk_t y = sbk_up(x, o); if (lane >= o) x = sbk_combine(y, x); }\n"
         "  return x;\n"
         "}\n";
    auto key_params = [&](bool trailing_comma) {
        std::ostringstream p;
        for (size_t k = 0; k < nk; ++k) p << "const " << K[k] << " *key" << k << (k + 1 < nk || trailing_comma ? ", " : "");
        return p.str();
    };
    auto key_args = [&]() {
        std::ostringstream p;
        for (size_t k = 0; k < nk; ++k) p << "key" << k << ", ";
        return p.str();
    };
    // One wave, ROWS rows of 64 x ITEMS consecutive elements.  A lane folds its ITEMS consecutive
    // elements serially (head flags against its own previous element: no shuffle), the wave scans
    // ONE aggregate per lane, and the prefix comes back into the lane's elements:
    // y[r * ITEMS + j] = inclusive prefix of element (r, lane, j) relative to the wave's first element.
    s << "__device__ inline sbk_t sbk_wave_tile(ulong n, ulong wbase, int lane, " << key_params(true) << "const val_t *vals, sbk_t *y) {\n"
         "  sbk_t carry = sbk_empty();\n"
         "  #pragma unroll\n"
         "  for (int r = 0; r < ROWS; ++r) {\n"
         "    const ulong i0 = wbase + (ulong)r * (64 * ITEMS) + (ulong)lane * ITEMS;\n";
    for (size_t k = 0; k < nk; ++k) {
        s << "    " << K[k] << " k" << k << "[ITEMS];\n"
          << "    #pragma unroll\n"
          << "    for (int j = 0; j < ITEMS; ++j) k" << k << "[j] = i0 + j < n ? key" << k << "[i0 + j] : (" << K[k] << ")0;\n"
          << "    " << K[k] << " p" << k << " = __shfl_up(k" << k << "[ITEMS - 1], 1, 64);\n"
          << "    if (lane == 0 && i0 > 0 && i0 < n) p" << k << " = key" << k << "[i0 - 1];\n";
    }
    s << "    sbk_t t[ITEMS];\n"
         "    sbk_t acc = sbk_empty();\n"
         "    #pragma unroll\n"
         "    for (int j = 0; j < ITEMS; ++j) {\n"
         "      sbk_t x = sbk_empty();\n"
         "      if (i0 + j < n) {\n";
    for (size_t k = 0; k < nk; ++k)
        s << "        const " << K[k] << " pk" << k << " = j ? k" << k << "[j ? j - 1 : 0] : p" << k << ";\n";
    s << "        const bool head = (i0 + j == 0) || !" << Comp::name() << "(";
    for (size_t k = 0; k < nk; ++k) s << "pk" << k << ", ";
    for (size_t k = 0; k < nk; ++k) s << "k" << k << "[j]" << (k + 1 < nk ? ", " : "");
    s << ");\n"
         "        x.c = head; x.f = 2 | (int)head; x.v = vals[i0 + j];\n"
         "      }\n"
         "      acc = sbk_combine(acc, x);\n"
         "      t[j] = acc;\n"
         "    }\n"
         "    const sbk_t incl = sbk_wave_scan(acc, lane);\n"
         "    sbk_t pre = sbk_up(incl, 1);\n"
         "    if (lane == 0) pre = sbk_empty();\n"
         "    pre = sbk_combine(carry, pre);\n"
         "    #pragma unroll\n"
         "    for (int j = 0; j < ITEMS; ++j) y[r * ITEMS + j] = sbk_combine(pre, t[j]);\n"
         "    carry = sbk_combine(carry, sbk_from(incl, 63));\n"
         "  }\n"
         "  return carry;\n"
         "}\n";

    s << "extern \"C\" __global__ void __launch_bounds__(" << WAVES * 64 << ") vexcl_sbk_reduce(ulong n, " << key_params(true)
      << "const val_t *vals, int *tc, int *tf, val_t *tv) {\n"
         "  __shared__ sbk_t agg[WAVES];\n"
         "  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;\n"
         "  sbk_t y[ROWS * ITEMS];\n"
         "  sbk_t a = sbk_wave_tile(n, ((ulong)blockIdx.x * WAVES + wave) * (ROWS * ITEMS * 64), lane, " << key_args() << "vals, y);\n"
         "  if (lane == 0) agg[wave] = a;\n"
         "  __syncthreads();\n"
         "  if (threadIdx.x == 0) {\n"
         "    sbk_t t = agg[0];\n"
         "    for (int w = 1; w < WAVES; ++w) t = sbk_combine(t, agg[w]);\n"
         "    tc[blockIdx.x] = t.c; tf[blockIdx.x] = t.f; tv[blockIdx.x] = t.v;\n"
         "  }\n"
         "}\n";

    // groups of 1024 tiles: exclusive scan of the tile aggregates inside the group (in place) and the
    // group's aggregate; the (few) group aggregates are then scanned by ONE workgroup below, and phase 3
    // combines group carry and in-group carry.  Two tiny launches instead of a 48-step serial loop.
    s << "extern \"C\" __global__ void __launch_bounds__(1024) vexcl_sbk_carry_local(int ntiles, int *tc, int *tf, val_t *tv, int *gc, int *gf, val_t *gv) {\n"
         "  __shared__ sbk_t wagg[16];\n"
         "  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;\n"
         "  const int t = blockIdx.x * 1024 + threadIdx.x;\n"
         "  sbk_t x = sbk_empty();\n"
         "  if (t < ntiles) { x.c = tc[t]; x.f = tf[t]; x.v = tv[t]; }\n"
         "  const sbk_t sc = sbk_wave_scan(x, lane);\n"
         "  if (lane == 63) wagg[wave] = sc;\n"
         "  __syncthreads();\n"
         "  sbk_t pre = sbk_empty();\n"
         "  for (int w = 0; w < wave; ++w) pre = sbk_combine(pre, wagg[w]);\n"
         "  const sbk_t incl = sbk_combine(pre, sc);\n"
         "  sbk_t excl = sbk_up(incl, 1);\n"
         "  if (lane == 0) excl = pre;\n"
         "  if (t < ntiles) { tc[t] = excl.c; tf[t] = excl.f; tv[t] = excl.v; }\n"
         "  if (threadIdx.x == 1023) { gc[blockIdx.x] = incl.c; gf[blockIdx.x] = incl.f; gv[blockIdx.x] = incl.v; }\n"
         "}\n";

    // one workgroup: lane t folds its own run of consecutive tile aggregates serially, ONE block scan
    // orders the 1024 runs, and the lane writes the carries of its run back (runs are one element long
    // up to 2^20 tiles = 2^31 elements)
    s << "extern \"C\" __global__ void __launch_bounds__(1024) vexcl_sbk_carry(int ntiles, int *tc, int *tf, val_t *tv, int *total) {\n"
         "  __shared__ sbk_t wagg[16];\n"
         "  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;\n"
         "  const int per = (ntiles + 1023) / 1024;\n"
         "  const int b = threadIdx.x * per, e = min(b + per, ntiles);\n"
         "  sbk_t acc = sbk_empty();\n"
         "  for (int t = b; t < e; ++t) { sbk_t x; x.c = tc[t]; x.f = tf[t]; x.v = tv[t]; acc = sbk_combine(acc, x); }\n"
         "  const sbk_t sc = sbk_wave_scan(acc, lane);\n"
         "  if (lane == 63) wagg[wave] = sc;\n"
         "  __syncthreads();\n"
         "  sbk_t pre = sbk_empty();\n"
         "  for (int w = 0; w < wave; ++w) pre = sbk_combine(pre, wagg[w]);\n"
         "  const sbk_t incl = sbk_combine(pre, sc);\n"
         "  sbk_t run = sbk_up(incl, 1);\n"
         "  if (lane == 0) run = pre;\n"
         "  for (int t = b; t < e; ++t) {\n"
         "    sbk_t x; x.c = tc[t]; x.f = tf[t]; x.v = tv[t];\n"
         "    tc[t] = run.c; tf[t] = run.f; tv[t] = run.v;\n"
         "    run = sbk_combine(run, x);\n"
         "  }\n"
         "  if (threadIdx.x == 1023) *total = incl.c;\n"
         "}\n";

    s << "extern \"C\" __global__ void __launch_bounds__(" << WAVES * 64 << ") vexcl_sbk_scan(ulong n, " << key_params(true)
      << "const val_t *vals, const int *tc, const int *tf, const val_t *tv, const int *gc, const int *gf, const val_t *gv, ";
    if (mode == REDUCE) {
        for (size_t k = 0; k < nk; ++k) s << K[k] << " *okey" << k << ", ";
        s << "val_t *ovals) {\n";
    } else {
        s << "val_t *ovals, val_t init) {\n";
    }
    // the store phase, shared by the three-phase kernel and the single-pass kernel: W = exclusive prefix of the wave's first element
    auto epilogue = [&](std::ostringstream &o) {
        o << "  sbk_t before = W;                 // inclusive prefix of the element just before this lane's first one\n"
             "  #pragma unroll\n"
             "  for (int r = 0; r < ROWS; ++r) {\n"
             "    const ulong i0 = wbase + (ulong)r * (64 * ITEMS) + (ulong)lane * ITEMS;\n"
             "    const sbk_t last = sbk_combine(W, y[r * ITEMS + ITEMS - 1]);\n"
             "    sbk_t prev = sbk_up(last, 1);\n"
             "    if (lane == 0) prev = before;\n"
             "    before = sbk_from(last, 63);\n";
        if (mode != REDUCE && (sizeof(V) == 4 || sizeof(V) == 8)) {
            // full rows: the lane's ITEMS results leave as 16-byte pieces, not element by element (see vexcl_sbk_lookback)
            o << "    if (wbase + (ulong)(r + 1) * (64 * ITEMS) <= n && ((ulong)ovals & 15) == 0) {\n"
                 "      typedef val_t sbk_vec3 __attribute__((ext_vector_type(16 / sizeof(val_t))));\n"
                 "      val_t out[ITEMS];\n"
                 "      #pragma unroll\n"
                 "      for (int j = 0; j < ITEMS; ++j) {\n"
                 "        const sbk_t fin = sbk_combine(W, y[r * ITEMS + j]);\n"
                 "        const bool head = fin.c != prev.c;\n";
            if (mode == INCLUSIVE) o << "        (void)head; (void)init; out[j] = fin.v;\n";
            else                   o << "        out[j] = head ? init : " << Oper::name() << "(init, prev.v);\n";
            o << "        prev = fin;\n"
                 "      }\n"
                 "      #pragma unroll\n"
                 "      for (int q = 0; q < (int)(ITEMS * sizeof(val_t) / 16); ++q) {\n"
                 "        sbk_vec3 t;\n"
                 "        #pragma unroll\n"
                 "        for (int e = 0; e < (int)(16 / sizeof(val_t)); ++e) t[e] = out[q * (16 / sizeof(val_t)) + e];\n"
                 "        ((sbk_vec3 *)(ovals + i0))[q] = t;\n"
                 "      }\n"
                 "      continue;\n"
                 "    }\n";
        }
        o << "    #pragma unroll\n"
             "    for (int j = 0; j < ITEMS; ++j) {\n"
             "      const ulong i = i0 + j;\n"
             "      const sbk_t fin = sbk_combine(W, y[r * ITEMS + j]);\n"
             "      if (i < n) {\n"
             "        const bool head = fin.c != prev.c;\n";
        if (mode == INCLUSIVE) {
            o << "        (void)head; (void)init; ovals[i] = fin.v;\n";
        } else if (mode == EXCLUSIVE) {
            o << "        ovals[i] = head ? init : " << Oper::name() << "(init, prev.v);\n";
        } else {
            o << "        if (head) {\n";
            for (size_t k = 0; k < nk; ++k) o << "          okey" << k << "[fin.c - 1] = key" << k << "[i];\n";
            o << "          if (fin.c > 1) ovals[fin.c - 2] = prev.v;\n"
                 "        }\n"
                 "        if (i == n - 1) ovals[fin.c - 1] = fin.v;\n";
        }
        o << "      }\n"
             "      prev = fin;\n"
             "    }\n"
             "  }\n"
             "}\n";
    };
    s << "  __shared__ sbk_t agg[WAVES];\n"
         "  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;\n"
         "  const ulong wbase = ((ulong)blockIdx.x * WAVES + wave) * (ROWS * ITEMS * 64);\n"
         "  sbk_t y[ROWS * ITEMS];\n"
         "  sbk_t a = sbk_wave_tile(n, wbase, lane, " << key_args() << "vals, y);\n"
         "  if (lane == 0) agg[wave] = a;\n"
         "  __syncthreads();\n"
         "  sbk_t W, G; W.c = tc[blockIdx.x]; W.f = tf[blockIdx.x]; W.v = tv[blockIdx.x];\n"
         "  G.c = gc[blockIdx.x >> 10]; G.f = gf[blockIdx.x >> 10]; G.v = gv[blockIdx.x >> 10];\n"
         "  W = sbk_combine(G, W);\n"
         "  for (int w = 0; w < wave; ++w) W = sbk_combine(W, agg[w]);\n";
    epilogue(s);

    if (lookback_value<V>::value) {
        const int NW = sizeof(V) == 8 ? 3 : 2;       // status words per tile: {count, flags, state}, value bits (+ state) in one or two words
        s << "\n#define LBW " << LB_WAVES << "\n#define NW " << NW << "\n"
             "typedef unsigned long long sbk_word;\n"
             "__device__ inline sbk_word sbk_ld(const sbk_word *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }\n"
             "__device__ inline void sbk_st(sbk_word *p, sbk_word v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }\n"
             // every word carries the state (1 aggregate, 2 inclusive): a reader accepts a triple only when all its words agree
             "__device__ inline void sbk_publish(sbk_word *st, long tile, sbk_t x, unsigned state) {\n"
             "  sbk_st(st + NW * tile, ((sbk_word)((state << 8) | (unsigned)x.f) << 32) | (unsigned)x.c);\n";
        if (NW == 3)
            s << "  sbk_word b; __builtin_memcpy(&b, &x.v, 8);\n"
                 "  sbk_st(st + NW * tile + 1, ((sbk_word)state << 32) | (unsigned)b);\n"
                 "  sbk_st(st + NW * tile + 2, ((sbk_word)state << 32) | (unsigned)(b >> 32));\n";
        else
            s << "  unsigned b; __builtin_memcpy(&b, &x.v, 4);\n"
                 "  sbk_st(st + NW * tile + 1, ((sbk_word)state << 32) | b);\n";
        s << "}\n"
             "__device__ inline unsigned sbk_read(const sbk_word *st, long tile, sbk_t &x) {\n"
             "  const sbk_word w0 = sbk_ld(st + NW * tile), w1 = sbk_ld(st + NW * tile + 1);\n"
             "  unsigned s0 = (unsigned)(w0 >> 40), s1 = (unsigned)(w1 >> 32);\n"
             "  x.c = (int)(unsigned)w0; x.f = (int)((w0 >> 32) & 255u);\n";
        if (NW == 3)
            s << "  const sbk_word w2 = sbk_ld(st + NW * tile + 2);\n"
                 "  const sbk_word b = (w2 << 32) | (w1 & 0xffffffffull); __builtin_memcpy(&x.v, &b, 8);\n"
                 "  if ((unsigned)(w2 >> 32) != s0) s0 = 0;\n";
        else
            s << "  const unsigned b = (unsigned)w1; __builtin_memcpy(&x.v, &b, 4);\n";
        s << "  return s0 == s1 ? s0 : 0u;\n"
             "}\n"
             "__device__ inline sbk_t sbk_down(sbk_t x, int o) { sbk_t r; r.c = __shfl_down(x.c, o, 64); r.f = __shfl_down(x.f, o, 64); r.v = __shfl_down(x.v, o, 64); return r; }\n";
        // The look-back of a tile, by ONE wave: the prefix of everything in front of the tile.
        // Counts and flags are integers: the window of 64 predecessors is folded by a shuffle tree, cut at the nearest one that has
        // published its inclusive prefix.  The VALUE (round 6) is not folded by that tree: where the tree is cut depends on how far the
        // predecessors happened to be, and with it the association of a floating-point carry that crosses several tiles -- the
        // last bits used to differ from run to run.  It is folded SERIALLY, oldest first, from the ANCHOR -- the nearest predecessor
        // whose value stands for everything in front of it: a tile that holds a run head (its aggregate's value is the open run's,
        // whatever came before) or one with an inclusive prefix.  An inclusive prefix is itself such a left fold (induction over
        // the tiles), so every choice of anchor continues the same chain: carry(k) = ((v(h) + v(h+1)) + ...) + v(k-1) from the
        // nearest head tile h, a function of the data alone.  Keys that change every few elements anchor at the nearest
        // predecessor: one step.  A run that spans hundreds of tiles folds up to 63 aggregates per window with readlane (about a
        // microsecond), and re-reads the windows between the anchor's and the nearest one (their words only ever advance from
        // aggregate to inclusive, which restarts the fold with the same value).
        s << "__device__ inline val_t sbk_lane(val_t v, int l) {\n"
             "  int b[sizeof(val_t) / 4];\n"
             "  __builtin_memcpy(b, &v, sizeof(val_t));\n"
             "  #pragma unroll\n"
             "  for (int i = 0; i < (int)(sizeof(val_t) / 4); ++i) b[i] = __builtin_amdgcn_readlane(b[i], l);\n"
             "  __builtin_memcpy(&v, b, sizeof(val_t));\n"
             "  return v;\n"
             "}\n"
             "__device__ inline sbk_t sbk_look_back(const sbk_word *status, long tile, int lane) {\n"
             "  sbk_t excl = sbk_empty();\n"
             "  if (tile == 0) return excl;\n"
             "  long base = tile - 1, spins = 0, abase = -1;\n"
             "  int alane = 0, q0f = 0;\n"
             "  val_t q0v = val_t();\n"                                                // the nearest window stays in registers
             "  while (base >= 0) {\n"
             "    const long idx = base - lane;\n"                                     // lane 0 = the nearest predecessor
             "    sbk_t q = sbk_empty();\n"
             "    unsigned st = 2u;\n"                                                 // lanes before tile 0 end the walk with the identity
             "    if (idx >= 0) st = sbk_read(status, idx, q);\n"
             "    while (__any(st == 0u)) {\n"
             "      __builtin_amdgcn_s_sleep(8);\n"
             "      if (idx >= 0 && st == 0u) st = sbk_read(status, idx, q);\n"
             "      if (++spins > (1l << 30)) __builtin_trap();\n"                    // a bug, never a truncated prefix (scan.hip)
             "    }\n"
             "    const unsigned long long incl = __ballot(st == 2u);\n"
             "    const int first = incl ? __builtin_ctzll(incl) : 63;\n"            // nearest predecessor with a complete prefix
             "    if (lane > first) q = sbk_empty();\n"
             "    if (base == tile - 1) { q0v = q.v; q0f = q.f; }\n"
             "    if (abase < 0) {\n"
             "      const unsigned long long R = __ballot(lane <= first && ((q.f & 1) || (incl && lane == first)));\n"
             "      if (R) { abase = base; alane = __builtin_ctzll(R); }\n"
             "    }\n"
             "    for (int o = 1; o < 64; o <<= 1) {\n"                                // counts and flags: older tiles (higher lanes) on the left
             "      sbk_t u = sbk_down(q, o);\n"
             "      if (lane + o < 64) q = sbk_combine(u, q);\n"
             "    }\n"
             "    excl = sbk_combine(sbk_from(q, 0), excl);\n"
             "    if (incl) break;\n"
             "    base -= 64;\n"
             "  }\n"
             "  sbk_t a = sbk_empty();\n"                                              // the value: from the anchor to the nearest predecessor, one tile after the other
             "  for (long b = abase; b <= tile - 1; b += 64) {\n"
             "    val_t wv = q0v; int wf = q0f;\n"
             "    if (b != tile - 1) {\n"
             "      const long idx = b - lane;\n"
             "      sbk_t q = sbk_empty();\n"
             "      unsigned st = 2u;\n"
             "      if (idx >= 0) {\n"
             "        st = sbk_read(status, idx, q);\n"
             "        while (st == 0u) {\n"                                            // (published long ago: only a reader that met the words half way from aggregate to inclusive)
             "          __builtin_amdgcn_s_sleep(1);\n"
             "          st = sbk_read(status, idx, q);\n"
             "          if (++spins > (1l << 30)) __builtin_trap();\n"
             "        }\n"
             "      }\n"
             "      wv = q.v; wf = q.f | (st == 2u ? 1 : 0);\n"                        // an inclusive prefix stands for everything in front of it, like a head
             "    }\n"
             "    int hi = b == abase ? alane : 63;\n"
             "    const unsigned long long Rw = __ballot((wf & 1) != 0 && lane <= hi);\n"
             "    if (Rw) hi = __builtin_ctzll(Rw);\n"                                 // (a nearer tile that has become inclusive since)
             "    for (int l = hi; l >= 0; --l) {\n"
             "      sbk_t x; x.c = 0; x.f = __builtin_amdgcn_readlane(wf, l); x.v = sbk_lane(wv, l);\n"
             "      a = sbk_combine(a, x);\n"
             "    }\n"
             "  }\n"
             "  excl.v = a.v;\n"
             "  return excl;\n"
             "}\n";
        if (dpp_enabled())
            // a value moved between lanes by DPP (CTRL: 0x110 + n = row_shr:n, 0x138 = wave_shr:1, 0x142 / 0x143 = row_bcast:15 / 31);
            // a lane without a source keeps its own value (the callers do not use it there)
            s << "template <int CTRL> __device__ inline val_t sbk_dpp(val_t v) {\n"
                 "  int b[sizeof(val_t) / 4];\n"
                 "  __builtin_memcpy(b, &v, sizeof(val_t));\n"
                 "  #pragma unroll\n"
                 "  for (int i = 0; i < (int)(sizeof(val_t) / 4); ++i) b[i] = __builtin_amdgcn_update_dpp(b[i], b[i], CTRL, 0xf, 0xf, false);\n"
                 "  __builtin_memcpy(&v, b, sizeof(val_t));\n"
                 "  return v;\n"
                 "}\n";
        // ws[0] = ticket counter, ws[1] = run count (reduce_by_key), ws + 2 = tile status words (all zero before launch).
        // Tile = 16 waves x LBR rows x 64 lanes x ITEMS consecutive elements = 16 Ki elements: a first version with 4 Ki
        // elements per tile (what the three phases use) took 0.85 ms per 1e8 (int, double) pairs against 0.77 ms for the
        // three phases -- 24 000 tiles start at 60 per microsecond, so every look-back met a few hundred predecessors
        // that had only published their aggregate.  Measured per 1e8 pairs (scan / reduce_by_key, ms; VEXCL_SBK_WAVES x
        // _ROWS): 16 x 4 0.60 / 0.56, 16 x 3 0.61 / 0.57, 8 x 4 0.64 / 0.56, 16 x 2 0.65 / 0.63, 4 x 4 0.77 / 0.65,
        // 8 x 2 0.78 / 0.73; without the look-back loop (wrong results) 16 x 4 takes 0.54 ms: what is left is the
        // 104-register, one-workgroup-per-CU body, not the look-back.  The lane keeps its 16 VALUES in registers between the two passes
        // over them (aggregate before the look-back, results after it); keys are only needed for the head flags (one bit
        // per element) and, in reduce_by_key, re-read at the run heads.
        s << "#define LBR " << LB_ROWS << "\n"
             // One 16-wave workgroup per CU: the kernel takes 104 vector registers (the lane's 16 values, the four row prefixes;
             // rocprofv3 reports them in pairs: 52).  Forcing two workgroups with a second launch bound (8 waves per SIMD, 64
             // registers) made it slower: 0.61 -> 0.75 ms (scan), 0.46 -> 0.67 ms (reduce_by_key).
             "extern \"C\" __global__ void __launch_bounds__(" << LB_WAVES * 64 << ") vexcl_sbk_lookback(ulong n, " << key_params(true)
          << "const val_t *vals, sbk_word *ws, ";
        if (mode == REDUCE) {
            for (size_t k = 0; k < nk; ++k) s << K[k] << " *okey" << k << ", ";
            s << "val_t *ovals, int cap) {\n";             // cap: the outputs hold that many runs (the true count goes to ws[1])
        } else {
            s << "val_t *ovals, val_t init) {\n";
        }
        s << "  __shared__ sbk_t agg[LBW];\n"
             "  __shared__ val_t s_x[sizeof(val_t) == 8 ? LBW : 1][ITEMS * 64] __attribute__((aligned(16)));\n"    // per wave: one row of results on its way out
             "  __shared__ long s_tile;\n"
             "  __shared__ sbk_t s_pre;\n"
             "  sbk_word *status = ws + 2;\n"
             "  if (threadIdx.x == 0) s_tile = (long)atomicAdd(&ws[0], 1ull);\n"     // tiles in launch order: a tile only waits for tiles that already run
             "  __syncthreads();\n"
             "  const long tile = s_tile;\n"
             "  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;\n"
             "  const ulong wbase = ((ulong)tile * LBW + wave) * (LBR * ITEMS * 64);\n"
             "  val_t v[LBR][ITEMS];\n"
             "  sbk_t pre[LBR];\n"                  // prefix of everything in this wave before the lane's first element of row r
             "  unsigned heads = 0;\n"              // bit r * ITEMS + j: element (r, lane, j) starts a run
             "  sbk_t carry = sbk_empty();\n"
             "  const unsigned long long below = (1ull << lane) - 1ull;\n"
             "  #pragma unroll\n"
             "  for (int r = 0; r < LBR; ++r) {\n"
             "    const ulong row0 = wbase + (ulong)r * (64 * ITEMS), i0 = row0 + (ulong)lane * ITEMS;\n"
             "    const bool full = row0 + 64 * ITEMS <= n;\n"       // uniform: no bounds checks, the loads of a lane merge into 16-byte loads
             "    unsigned h = 0, ok = 0;\n";
        for (size_t k = 0; k < nk; ++k) s << "    " << K[k] << " k" << k << "[ITEMS];\n";
        s << "    if (full) {\n"
             "      ok = (1u << ITEMS) - 1u;\n"
             "      #pragma unroll\n"
             "      for (int j = 0; j < ITEMS; ++j) {\n";
        for (size_t k = 0; k < nk; ++k) s << "        k" << k << "[j] = key" << k << "[i0 + j];\n";
        s << "        v[r][j] = vals[i0 + j];\n"
             "      }\n"
             "    } else {\n"
             "      #pragma unroll\n"
             "      for (int j = 0; j < ITEMS; ++j) {\n"
             "        const bool in = i0 + j < n;\n"
             "        ok |= (unsigned)in << j;\n";
        for (size_t k = 0; k < nk; ++k) s << "        k" << k << "[j] = in ? key" << k << "[i0 + j] : (" << K[k] << ")0;\n";
        s << "        v[r][j] = in ? vals[i0 + j] : val_t();\n"
             "      }\n"
             "    }\n";
        for (size_t k = 0; k < nk; ++k)
            s << "    " << K[k] << " p" << k << " = __shfl_up(k" << k << "[ITEMS - 1], 1, 64);\n"
              << "    if (lane == 0 && i0 > 0 && i0 < n) p" << k << " = key" << k << "[i0 - 1];\n";
        // the lane's own four elements: head flags, running value since the lane's last head
        s << "    val_t tail = val_t();\n"
             "    #pragma unroll\n"
             "    for (int j = 0; j < ITEMS; ++j) {\n";
        for (size_t k = 0; k < nk; ++k)
            s << "      const " << K[k] << " pk" << k << " = j ? k" << k << "[j ? j - 1 : 0] : p" << k << ";\n";
        s << "      const bool hd = ((ok >> j) & 1u) && ((i0 + j == 0) || !" << Comp::name() << "(";
        for (size_t k = 0; k < nk; ++k) s << "pk" << k << ", ";
        for (size_t k = 0; k < nk; ++k) s << "k" << k << "[j]" << (k + 1 < nk ? ", " : "");
        s << "));\n"
             "      h |= (unsigned)hd << j;\n"
             "      if ((ok >> j) & 1u) tail = (hd || j == 0) ? v[r][j] : " << Oper::name() << "(tail, v[r][j]);\n"
             "    }\n"
             "    heads |= h << (r * ITEMS);\n"
             // Across the lanes: which lanes hold a head is a ballot, so the segmented scan of the tails needs no flag
             // traffic -- lane L adds the value of lane L - o exactly when no lane in (L - o, L] holds a head, i.e. when
             // L - o is not below the nearest head lane at or before L; the number of heads before a lane is a popcount.
             "    const unsigned long long H = __ballot(h != 0u), any = __ballot(ok != 0u);\n"
             "    const unsigned long long upto = H & (below | (1ull << lane));\n"
             "    const int hl = upto ? 63 - __builtin_clzll(upto) : 0;\n"
             "    val_t T = tail;\n";
        if (dpp_enabled()) {
            // rows of 16 lanes by row_shr, then lane 15 / 47 into the row behind it, then lane 31 into the upper half: the same
            // rule at every step -- a lane adds what it is handed exactly when the lane that value comes from is not below the
            // nearest head lane at or before it (the value then is the sum of that lane's stretch of the SAME run)
            s << "    const bool live = (any >> lane) & 1ull;\n"
                 "    { const val_t u = sbk_dpp<0x111>(T); if ((lane & 15) >= 1 && lane - 1 >= hl && live) T = " << Oper::name() << "(u, T); }\n"
                 "    { const val_t u = sbk_dpp<0x112>(T); if ((lane & 15) >= 2 && lane - 2 >= hl && live) T = " << Oper::name() << "(u, T); }\n"
                 "    { const val_t u = sbk_dpp<0x114>(T); if ((lane & 15) >= 4 && lane - 4 >= hl && live) T = " << Oper::name() << "(u, T); }\n"
                 "    { const val_t u = sbk_dpp<0x118>(T); if ((lane & 15) >= 8 && lane - 8 >= hl && live) T = " << Oper::name() << "(u, T); }\n"
                 "    { const val_t u = sbk_dpp<0x142>(T); if ((lane & 16) && (lane & 48) - 1 >= hl && live) T = " << Oper::name() << "(u, T); }\n"
                 "    { const val_t u = sbk_dpp<0x143>(T); if (lane >= 32 && 31 >= hl && live) T = " << Oper::name() << "(u, T); }\n";
        } else {
        s << "    #pragma unroll\n"
             "    for (int o = 1; o < 64; o <<= 1) {\n"
             "      const val_t u = __shfl_up(T, o, 64);\n"
             "      if (lane - o >= hl && ((any >> lane) & 1ull)) T = " << Oper::name() << "(u, T);\n"
             "    }\n";
        }
        s << "    int cb = 0;\n"
             "    #pragma unroll\n"
             "    for (int j = 0; j < ITEMS; ++j) cb += __popcll(__ballot((h >> j) & 1u) & below);\n"
             "    sbk_t p; p.c = cb; p.f = ((any & below) ? 2 : 0) | ((H & below) ? 1 : 0); p.v = " << (dpp_enabled() ? "sbk_dpp<0x138>(T)" : "__shfl_up(T, 1, 64)") << ";\n"
             "    if (lane == 0) p = sbk_empty();\n"
             "    pre[r] = sbk_combine(carry, p);\n"
             // the row's aggregate: all its heads, the value of the run that is open at its end
             "    sbk_t ra; ra.c = 0;\n"
             "    #pragma unroll\n"
             "    for (int j = 0; j < ITEMS; ++j) ra.c += __popcll(__ballot((h >> j) & 1u));\n"
             "    ra.f = (any ? 2 : 0) | (H ? 1 : 0);\n"
             "    ra.v = __shfl(T, any ? 63 - __builtin_clzll(any) : 0, 64);\n"
             "    carry = sbk_combine(carry, ra);\n"
             "  }\n"
             "  if (lane == 0) agg[wave] = carry;\n"
             "  __syncthreads();\n"
             "  if (wave == 0) {\n"
             "    sbk_t t = agg[0];\n"
             "    for (int w = 1; w < LBW; ++w) t = sbk_combine(t, agg[w]);\n"
             "    if (lane == 0) sbk_publish(status, tile, t, tile == 0 ? 2u : 1u);\n"
             "    const sbk_t excl = sbk_look_back(status, tile, lane);\n"
             "    if (lane == 0) {\n"
             "      if (tile > 0) sbk_publish(status, tile, sbk_combine(excl, t), 2u);\n"
             "      s_pre = excl;\n"
             "    }\n"
             "  }\n"
             "  __syncthreads();\n"
             "  sbk_t W = s_pre;\n"
             "  for (int w = 0; w < wave; ++w) W = sbk_combine(W, agg[w]);\n"
             // second pass over the lane's values: prev = inclusive prefix of the element before, fin = of the element itself
             "  #pragma unroll\n"
             "  for (int r = 0; r < LBR; ++r) {\n"
             "    const ulong row0 = wbase + (ulong)r * (64 * ITEMS), i0 = row0 + (ulong)lane * ITEMS;\n"
             "    sbk_t prev = sbk_combine(W, pre[r]);\n";
        if (mode != REDUCE) {
            // Full rows: the lane's ITEMS results leave as 16-byte pieces, and for 8-byte values (32 bytes per lane) through a
            // wave-private LDS row, so that every store instruction of the wave writes 1 KiB of consecutive bytes.  The guarded
            // loop below stores element by element -- 8-byte pieces at a 32-byte lane stride, 16 store instructions per lane and
            // tile: with it alone the scan took 0.60 ms per 1e8 (int, double) pairs; 16-byte pieces at a 32-byte stride 0.505;
            // consecutive pieces 0.474.  (The same detour for the LOADS of the values changed nothing: 0.480.)
            s << "    if (row0 + 64 * ITEMS <= n && ((ulong)ovals & 15) == 0) {\n"
                 "      val_t out[ITEMS];\n"
                 "      #pragma unroll\n"
                 "      for (int j = 0; j < ITEMS; ++j) {\n"
                 "        const bool head = (heads >> (r * ITEMS + j)) & 1u;\n"
                 "        sbk_t x; x.c = head; x.f = 2 | (int)head; x.v = v[r][j];\n"
                 "        const sbk_t fin = sbk_combine(prev, x);\n";
            if (mode == INCLUSIVE) s << "        (void)init; out[j] = fin.v;\n";
            else                   s << "        out[j] = head ? init : " << Oper::name() << "(init, prev.v);\n";
            s << "        prev = fin;\n"
                 "      }\n"
                 "      typedef val_t sbk_vec __attribute__((ext_vector_type(16 / sizeof(val_t))));\n"
                 "      #define NQ ((int)(ITEMS * sizeof(val_t) / 16))\n"
                 "      sbk_vec o[NQ];\n"
                 "      #pragma unroll\n"
                 "      for (int q = 0; q < NQ; ++q)\n"
                 "        #pragma unroll\n"
                 "        for (int e = 0; e < (int)(16 / sizeof(val_t)); ++e) o[q][e] = out[q * (16 / sizeof(val_t)) + e];\n"
                 "      if (NQ == 1) { ((sbk_vec *)(ovals + row0))[lane] = o[0]; continue; }\n"
                 "      sbk_vec *sx = (sbk_vec *)s_x[wave];\n"
                 "      #pragma unroll\n"
                 "      for (int q = 0; q < NQ; ++q) sx[lane * NQ + q] = o[q];\n"
                 "      __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n"
                 "      #pragma unroll\n"
                 "      for (int q = 0; q < NQ; ++q) ((sbk_vec *)(ovals + row0))[q * 64 + lane] = sx[q * 64 + lane];\n"
                 "      __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier();\n"
                 "      #undef NQ\n"
                 "      continue;\n"
                 "    }\n";
        }
        s << "    #pragma unroll\n"
             "    for (int j = 0; j < ITEMS; ++j) {\n"
             "      const ulong i = i0 + j;\n"
             "      if (i < n) {\n"
             "        const bool head = (heads >> (r * ITEMS + j)) & 1u;\n"
             "        sbk_t x; x.c = head; x.f = 2 | (int)head; x.v = v[r][j];\n"
             "        const sbk_t fin = sbk_combine(prev, x);\n";
        if (mode == INCLUSIVE) {
            s << "        (void)init; ovals[i] = fin.v;\n";
        } else if (mode == EXCLUSIVE) {
            s << "        ovals[i] = head ? init : " << Oper::name() << "(init, prev.v);\n";
        } else {
            s << "        if (head && fin.c <= cap) {\n";
            for (size_t k = 0; k < nk; ++k) s << "          okey" << k << "[fin.c - 1] = key" << k << "[i];\n";
            s << "          if (fin.c > 1) ovals[fin.c - 2] = prev.v;\n"
                 "        }\n"
                 "        if (i == n - 1) { if (fin.c <= cap) ovals[fin.c - 1] = fin.v; ((int *)(ws + 1))[0] = fin.c; }\n";
        }
        s << "        prev = fin;\n"
             "      }\n"
             "    }\n"
             "  }\n"
             "}\n";
        // number of run heads (keys only): sizes the outputs of reduce_by_key before its single pass.  A lane owns ITEMS
        // consecutive keys (one 16-byte load for 4-byte keys on full blocks); the key before its first one comes from the
        // neighbour lane.  (First version: five guarded 4-byte loads per lane and step, 0.76 ms per 1e8 keys.)
        s << "extern \"C\" __global__ void __launch_bounds__(256) vexcl_sbk_count(ulong n, " << key_params(true) << "int *total) {\n"
             "  const int lane = threadIdx.x & 63;\n"
             "  int c = 0;\n"
             "  for (ulong b0 = (ulong)blockIdx.x * (256 * ITEMS); b0 < n; b0 += (ulong)gridDim.x * (256 * ITEMS)) {\n"
             "    const ulong i0 = b0 + (ulong)threadIdx.x * ITEMS;\n"
             "    const bool full = b0 + 256 * ITEMS <= n;\n";
        for (size_t k = 0; k < nk; ++k) s << "    " << K[k] << " k" << k << "[ITEMS];\n";
        s << "    if (full) {\n"
             "      #pragma unroll\n"
             "      for (int j = 0; j < ITEMS; ++j) {\n";
        for (size_t k = 0; k < nk; ++k) s << "        k" << k << "[j] = key" << k << "[i0 + j];\n";
        s << "      }\n"
             "    } else {\n"
             "      #pragma unroll\n"
             "      for (int j = 0; j < ITEMS; ++j) {\n";
        for (size_t k = 0; k < nk; ++k) s << "        k" << k << "[j] = i0 + j < n ? key" << k << "[i0 + j] : (" << K[k] << ")0;\n";
        s << "      }\n"
             "    }\n";
        for (size_t k = 0; k < nk; ++k)
            s << "    " << K[k] << " p" << k << " = __shfl_up(k" << k << "[ITEMS - 1], 1, 64);\n"
              << "    if (lane == 0 && i0 > 0 && i0 < n) p" << k << " = key" << k << "[i0 - 1];\n";
        s << "    #pragma unroll\n"
             "    for (int j = 0; j < ITEMS; ++j) {\n";
        for (size_t k = 0; k < nk; ++k)
            s << "      const " << K[k] << " pk" << k << " = j ? k" << k << "[j ? j - 1 : 0] : p" << k << ";\n";
        s << "      if (i0 + j < n) c += (i0 + j == 0) || !" << Comp::name() << "(";
        for (size_t k = 0; k < nk; ++k) s << "pk" << k << ", ";
        for (size_t k = 0; k < nk; ++k) s << "k" << k << "[j]" << (k + 1 < nk ? ", " : "");
        s << ");\n"
             "    }\n"
             "  }\n"
             "  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);\n"
             "  __shared__ int wc[4];\n"                    // one atomic per workgroup: 390 000 per-wave atomics on one word took 0.75 ms
             "  if (lane == 0) wc[threadIdx.x >> 6] = c;\n"
             "  __syncthreads();\n"
             "  if (threadIdx.x == 0 && (wc[0] + wc[1] + wc[2] + wc[3])) atomicAdd(total, wc[0] + wc[1] + wc[2] + wc[3]);\n"
             "}\n";
        if (pipe_enabled()) pipe_source<V, Comp, Oper>(s, K, mode);
    }
    return src.str() + s.str();
}

template <class Tuple, class F, size_t... I>
void for_each_key(const Tuple &t, F &&f, std::index_sequence<I...>) {
    int dummy[] = {0, (f(std::get<I>(t)), 0)...};
    (void)dummy;
}

/// Runs phases 1 and 2; returns the number of segments.  `finish` then launches phase 3.
template <scan_mode mode, class KTuple, class V, class Comp, class Oper, class PushOutputs>
int run(const KTuple &keys, const vector<V> &ivals, Comp, Oper, PushOutputs &&push_outputs, bool need_count, bool three_phases = false, size_t guess = 0) {
    constexpr size_t nk = std::tuple_size<KTuple>::value;
    typedef std::make_index_sequence<nk> seq;
    const auto &queue = ivals.queue_list();
    precondition(queue.size() == 1, "scan_by_key / reduce_by_key are only supported for single-device contexts");
    const backend::command_queue &q = queue[0];
    const size_t n = ivals.size();
    for_each_key(keys, [&](const auto &k) { precondition(k.size() == n, "keys and values have different sizes"); }, seq());
    if (!n) return 0;

    static object_cache<kernels> cache;
    auto it = cache.find(q);
    if (it == cache.end()) {
        backend::program prog = backend::build_sources(q, source<V, Comp, Oper>(q, key_types<KTuple>(seq()), mode));
        kernels k;
        k.reduce = backend::kernel(q, prog, "vexcl_sbk_reduce");
        k.carry_local = backend::kernel(q, prog, "vexcl_sbk_carry_local");
        k.carry = backend::kernel(q, prog, "vexcl_sbk_carry");
        k.scan = backend::kernel(q, prog, "vexcl_sbk_scan");
        if (lookback_value<V>::value) {
            k.lookback = backend::kernel(q, prog, "vexcl_sbk_lookback");
            k.count = backend::kernel(q, prog, "vexcl_sbk_count");
            k.has_lookback = true;
            if (pipe_enabled()) { k.pipe = backend::kernel(q, prog, "vexcl_sbk_pipe"); k.has_pipe = true; }
        }
        it = cache.insert(q, std::move(k));
    }
    kernels &K = it->second;

    if (K.has_lookback && !three_phases && lookback_enabled()) {
        // ---- single pass: [count the run heads (keys only) -> size the outputs] -> decoupled look-back over the tiles
        const bool piped = K.has_pipe;
        const size_t LT = size_t(piped ? pipe_rows() : LB_ROWS) * ITEMS * (piped ? pipe_waves() : LB_WAVES) * 64;
        const size_t nt = (n + LT - 1) / LT;
        precondition(nt < (size_t(1) << 31), "input too large");
        const size_t words = 2 + (sizeof(V) == 8 ? 3 : 2) * nt;
        backend::device_vector<char> wsb = scratch_pool::instance().get(q, 4, words * 8);
        unsigned long long *ws = reinterpret_cast<unsigned long long *>(wsb.raw());
        backend::check(vexhip_memset(q.device_ordinal(), ws, 0, words * 8, q.raw()));      // ticket, run count, tile states
        int count = 0;
        int *total = reinterpret_cast<int *>(ws + 1);
        auto read_total = [&]() { int c = 0; backend::device_vector<int> t = backend::device_vector<int>::wrap(total, 1); t.read(q, 0, 1, &c, true); return c; };
        auto launch = [&](int cap) {
            if (piped) {
                // as many workgroups as can be resident at once (one per CU: ~170 registers x 12 waves), each takes tiles until none is left
                K.pipe.push_arg(n);
                K.pipe.push_arg(static_cast<long>(nt));
                for_each_key(keys, [&](const auto &k) { K.pipe.push_arg(k(0).raw()); }, seq());
                K.pipe.push_arg(ivals(0).raw());
                K.pipe.push_arg(ws);
                push_outputs(K.pipe, cap);
                if (mode == REDUCE) K.pipe.push_arg(cap);
                K.pipe.config(std::min<size_t>(nt, static_cast<size_t>(lb_env("VEXCL_SBK_PIPE_GROUPS", 256))), (pipe_waves() + 1) * 64);
                K.pipe(q);
                return;
            }
            K.lookback.push_arg(n);
            for_each_key(keys, [&](const auto &k) { K.lookback.push_arg(k(0).raw()); }, seq());
            K.lookback.push_arg(ivals(0).raw());
            K.lookback.push_arg(ws);
            push_outputs(K.lookback, cap);
            if (mode == REDUCE) K.lookback.push_arg(cap);
            K.lookback.config(nt, LB_WAVES * 64);
            K.lookback(q);
        };
        if (need_count && guess > 0 && guess < (size_t(1) << 31)) {
            // The outputs already hold `guess` runs (the previous call on data of this shape): ONE pass that stores what fits
            // and reports the true count.  Right guess (an iteration that calls this again and again): done -- no keys-only
            // pass, no read-back between two kernels.  Wrong guess: the count is known now, the pass runs again.
            launch(static_cast<int>(guess));
            count = read_total();
            if (static_cast<size_t>(count) == guess) return count;
            backend::check(vexhip_memset(q.device_ordinal(), ws, 0, words * 8, q.raw()));
            launch(count);
            return count;
        }
        if (need_count) {
            K.count.push_arg(n);
            for_each_key(keys, [&](const auto &k) { K.count.push_arg(k(0).raw()); }, seq());
            K.count.push_arg(total);
            K.count.config(std::min<size_t>((n + 256 * ITEMS - 1) / (256 * ITEMS), size_t(256) * 16), 256);
            K.count(q);
            count = read_total();
        }
        launch(count);
        return count;
    }

    const size_t ntiles = (n + TILE - 1) / TILE;
    precondition(ntiles < (size_t(1) << 31), "input too large");
    auto &pool = scratch_pool::instance();
    const size_t ngroups = (ntiles + 1023) / 1024;
    backend::device_vector<char> tcb = pool.get(q, 4, (2 * ntiles + 2 * ngroups + 1) * sizeof(int));
    backend::device_vector<char> tvb = pool.get(q, 5, (ntiles + ngroups) * sizeof(V));
    int *tc = reinterpret_cast<int *>(tcb.raw()), *tf = tc + ntiles, *gc = tf + ntiles, *gf = gc + ngroups, *total = gf + ngroups;
    V *tv = reinterpret_cast<V *>(tvb.raw()), *gv = tv + ntiles;

    K.reduce.push_arg(n);
    for_each_key(keys, [&](const auto &k) { K.reduce.push_arg(k(0).raw()); }, seq());
    K.reduce.push_arg(ivals(0).raw()); K.reduce.push_arg(tc); K.reduce.push_arg(tf); K.reduce.push_arg(tv);
    K.reduce.config(ntiles, WAVES * 64);
    K.reduce(q);

    K.carry_local.push_arg(static_cast<int>(ntiles)); K.carry_local.push_arg(tc); K.carry_local.push_arg(tf); K.carry_local.push_arg(tv);
    K.carry_local.push_arg(gc); K.carry_local.push_arg(gf); K.carry_local.push_arg(gv);
    K.carry_local.config(ngroups, 1024);
    K.carry_local(q);

    K.carry.push_arg(static_cast<int>(ngroups)); K.carry.push_arg(gc); K.carry.push_arg(gf); K.carry.push_arg(gv); K.carry.push_arg(total);
    K.carry.config(1, 1024);
    K.carry(q);

    int count = 0;
    if (need_count) {
        backend::device_vector<int> t = backend::device_vector<int>::wrap(total, 1);
        t.read(q, 0, 1, &count, true);
    }

    K.scan.push_arg(n);
    for_each_key(keys, [&](const auto &k) { K.scan.push_arg(k(0).raw()); }, seq());
    K.scan.push_arg(ivals(0).raw());
    K.scan.push_arg(static_cast<const int *>(tc)); K.scan.push_arg(static_cast<const int *>(tf)); K.scan.push_arg(static_cast<const V *>(tv));
    K.scan.push_arg(static_cast<const int *>(gc)); K.scan.push_arg(static_cast<const int *>(gf)); K.scan.push_arg(static_cast<const V *>(gv));
    push_outputs(K.scan, count);
    K.scan.config(ntiles, WAVES * 64);
    K.scan(q);
    return count;
}

template <bool exclusive, class KTuple, class V, class Comp, class Oper>
void scan_by_key(const KTuple &keys, const vector<V> &ivals, vector<V> &ovals, Comp comp, Oper oper, V init, bool three_phases = false) {
    precondition(ivals.size() == ovals.size(), "input and output have different sizes");
    run<exclusive ? EXCLUSIVE : INCLUSIVE>(keys, ivals, comp, oper,
            [&](backend::kernel &k, int) { k.push_arg(ovals(0).raw()); k.push_arg(init); }, false, three_phases);
}

} // namespace sbk
} // namespace detail

/// ovals[i] = init (+) ivals[first of i's run] (+) ... (+) ivals[i - 1]; init at the first
/// element of every run of equal keys (scan_by_key.hpp:713-721).
template <class K, typename V, class Comp, class Oper>
void exclusive_scan_by_key(const K &keys, const vector<V> &ivals, vector<V> &ovals, Comp comp, Oper oper, V init = V()) {
    detail::sbk::scan_by_key<true>(detail::sbk::key_seq<K>::get(keys), ivals, ovals, comp, oper, init);
}
/// ovals[i] = ivals[first of i's run] (+) ... (+) ivals[i] (scan_by_key.hpp:725-733).
template <class K, typename V, class Comp, class Oper>
void inclusive_scan_by_key(const K &keys, const vector<V> &ivals, vector<V> &ovals, Comp comp, Oper oper, V init = V()) {
    detail::sbk::scan_by_key<false>(detail::sbk::key_seq<K>::get(keys), ivals, ovals, comp, oper, init);
}

namespace detail { namespace sbk {
    // default functions: keys compared with ==, values added (scan_by_key.hpp:742-744)
    template <class K> struct equal_fn : UserFunction<equal_fn<K>, bool> {
        static std::string name() { return "sbk_equal"; }
        static void params(std::vector<std::pair<std::string, std::string>> &p) {
            p.push_back(std::make_pair(type_name<K>(), std::string("x"))); p.push_back(std::make_pair(type_name<K>(), std::string("y")));
        }
        static std::string body() { return "return x == y;"; }
    };
    template <class V> struct plus_fn : UserFunction<plus_fn<V>, V> {
        static std::string name() { return "sbk_plus"; }
        static void params(std::vector<std::pair<std::string, std::string>> &p) {
            p.push_back(std::make_pair(type_name<V>(), std::string("x"))); p.push_back(std::make_pair(type_name<V>(), std::string("y")));
        }
        static std::string body() { return "return x + y;"; }
    };
}}

template <typename K, typename V>
void exclusive_scan_by_key(const vector<K> &keys, const vector<V> &ivals, vector<V> &ovals, V init = V()) {
    exclusive_scan_by_key(keys, ivals, ovals, detail::sbk::equal_fn<K>(), detail::sbk::plus_fn<V>(), init);
}
template <typename K, typename V>
void inclusive_scan_by_key(const vector<K> &keys, const vector<V> &ivals, vector<V> &ovals, V init = V()) {
    inclusive_scan_by_key(keys, ivals, ovals, detail::sbk::equal_fn<K>(), detail::sbk::plus_fn<V>(), init);
}

namespace detail {
/// Scan with a user operator: per device one segment of the segmented scan (all keys equal); between
/// devices the carry of the preceding partitions is applied with one fused kernel per device, the
/// reference's scheme (scan.hpp:436-457, :478-506).
template <class T, class Oper>
void generic_scan(const vector<T> &input, vector<T> &output, bool exclusive, T init, Oper oper) {
    precondition(input.size() == output.size() && input.nparts() == output.nparts(), "scan: incompatible vectors");
    typedef typename std::decay<decltype(oper.device)>::type device_oper;
    const auto &queue = input.queue_list();
    const unsigned nd = static_cast<unsigned>(queue.size());
    std::vector<T> last_in(nd, T()), tail(nd, T());
    std::vector<char> used(nd, 0);
    for (unsigned d = 0; d < nd; ++d) {
        const size_t n = input.part_size(d);
        if (!n) continue;
        used[d] = 1;
        if (nd > 1 && exclusive) input(d).read(queue[d], n - 1, 1, &last_in[d], true);
        std::vector<backend::command_queue> one(1, queue[d]);
        vector<int> key(one, n);
        key = 0;
        vector<T> in(queue[d], input(d), n), out(queue[d], output(d), n);
        auto keys = std::tuple<const vector<int> &>(key);
        // one run as long as the partition: the three deterministic phases (a look-back would associate the carry as it goes)
        if (exclusive) sbk::scan_by_key<true>(keys, in, out, sbk::equal_fn<int>(), device_oper(), init, true);
        else           sbk::scan_by_key<false>(keys, in, out, sbk::equal_fn<int>(), device_oper(), init, true);
        if (nd > 1) {
            T last_out; output(d).read(queue[d], n - 1, 1, &last_out, true);
            tail[d] = exclusive ? oper(last_out, last_in[d]) : last_out;     // the partition's total (exclusive: includes init)
        }
    }
    if (nd > 1) {
        // exclusive partitions after the first were started from init as well: the carry replaces it only
        // when init is the operator's identity -- the reference makes the same assumption (scan.hpp:489-506)
        bool have = false; T carry = T();
        for (unsigned d = 0; d < nd; ++d) {
            if (!used[d]) continue;
            if (have) {
                vector<T> seg(queue[d], output(d), input.part_size(d));
                seg = oper.device(carry, seg);
                carry = oper(carry, tail[d]);
            } else { carry = tail[d]; have = true; }
        }
    }
}
} // namespace detail

} // namespace vex
#endif
