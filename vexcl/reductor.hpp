#ifndef VEXCL_REDUCTOR_HPP
#define VEXCL_REDUCTOR_HPP
// vex::Reductor<T, RDC> (reference: vexcl/reductor.hpp:47-280 operations,
// :302-439 two-stage reduction).  Stage 1 is generated per expression type:
// per-lane grid-stride accumulation (reductor.hpp:511-564), then a wave-64
// __shfl_down fold and ONE LDS hop across the workgroup's waves (the reference
// tree-reduces 1024 -> 1 through LDS with a barrier per level, :371-379).
// Stage 2 runs on the device too (libvexhip: vexhip_reduce_finish), so each GPU
// hands back one scalar; the host only combines one value per GPU
// (the reference folds 8 x CU partials per device on the host, :412-436).
#include <cstring>
#include <chrono>
#include <atomic>
#include <limits>
#include <map>
#include <memory>
#include "operations.hpp"
#include "exchange.hpp"
#include "vector.hpp"
#include "multivector.hpp"

namespace vex {

template <class T> struct cl_vec2 { T s[2]; };

/// Summation (reductor.hpp:47-72).
struct SUM {
    template <class T> struct impl {
        typedef T result_type;
        static const int op = VEXHIP_SUM;
        static T initial() { return T(); }
        static std::string device(const std::string &a, const std::string &b) { return a + " + " + b; }
        T operator()(const T &a, const T &b) const { return a + b; }
    };
};
/// Compensated summation (reductor.hpp:74-80, kernel body :537-564).
struct SUM_Kahan : SUM {};
/// Maximum (reductor.hpp:82-104).
struct MAX {
    template <class T> struct impl {
        typedef T result_type;
        static const int op = VEXHIP_MAX;
        static T initial() { return std::numeric_limits<T>::lowest(); }
        static std::string device(const std::string &a, const std::string &b) { return "(" + a + " > " + b + " ? " + a + " : " + b + ")"; }
        T operator()(const T &a, const T &b) const { return a > b ? a : b; }
    };
};
/// Minimum (reductor.hpp:106-128).
struct MIN {
    template <class T> struct impl {
        typedef T result_type;
        static const int op = VEXHIP_MIN;
        static T initial() { return std::numeric_limits<T>::max(); }
        static std::string device(const std::string &a, const std::string &b) { return "(" + a + " < " + b + " ? " + a + " : " + b + ")"; }
        T operator()(const T &a, const T &b) const { return a < b ? a : b; }
    };
};
/// Minimum and maximum in one pass (reductor.hpp CombineReductors<MIN, MAX>).
struct MIN_MAX {};

namespace detail {
    template <class T> struct reduce_dtype;
    template <> struct reduce_dtype<double> { static const int value = VEXHIP_F64; };
    template <> struct reduce_dtype<float> { static const int value = VEXHIP_F32; };
    template <> struct reduce_dtype<int> { static const int value = VEXHIP_I32; };
    template <> struct reduce_dtype<unsigned> { static const int value = VEXHIP_U32; };
    template <> struct reduce_dtype<long> { static const int value = VEXHIP_I64; };
    template <> struct reduce_dtype<unsigned long> { static const int value = VEXHIP_U64; };
    template <> struct reduce_dtype<long long> { static const int value = VEXHIP_I64; };
    template <> struct reduce_dtype<unsigned long long> { static const int value = VEXHIP_U64; };

    template <class T> std::string literal(T v) {
        std::ostringstream s;
        s.precision(std::numeric_limits<T>::max_digits10);
        if (std::is_floating_point<T>::value) s << std::scientific;
        s << v;
        return "(" + type_name<T>() + ")(" + s.str() + ")";
    }

    // How the workgroup that folds the partials of a single-launch reduction gets to see them (VEXCL_REDUCTOR_ORDER):
    //   tagged (default) -- every partial travels as 8-byte words {32 bits of the value | number of this reduction}, written
    //                       and read with relaxed agent-scope atomics (a 4-byte value is one word, an 8-byte value two).  The
    //                       folding workgroup re-reads a word until it carries this reduction's number: coherence of ONE atomic
    //                       object is all that is used -- no ordering between different locations, hence no fence, and nothing
    //                       is inferred from the arrival counter but who folds.  Correct by the HIP / LLVM memory model.
    //   release          -- relaxed store of the partial, release fence, arrival; acquire fences in the workgroups that pass
    //                       counts on and in the one that folds.  Correct by the model, and slow: 2048 L2 write-backs per
    //                       launch (0.263 -> 0.31 ms at 1e8, 0.049 -> 0.066 ms at 2^24: profiles/r05_reduce_order.log).
    //   relaxed          -- round 4's form (atomic exchange + compiler barrier): relies on gfx942 / gfx950 performing a returning
    //                       sc1 atomic at the memory side before it returns; NOT ordered by the model.  A/B only.
    //   two_launch       -- stage 2 is libvexhip's vexhip_reduce_finish on the same queue (kernel boundary).
    // All four fold in the same order: the bits are the same.
    enum reduction_order { order_release = 0, order_relaxed = 1, order_two_launch = 2, order_tagged = 3 };
    inline int &reductor_order_override() { static int o = -1; return o; }      // tests: a mode per generated source (-1: the environment's)
    inline reduction_order reductor_order() {
        if (reductor_order_override() >= 0) return static_cast<reduction_order>(reductor_order_override());
        static const reduction_order m = [] {
            const char *e = std::getenv("VEXCL_REDUCTOR_ORDER");
            const std::string v = e ? e : "";
            if (v == "relaxed") return order_relaxed;
            if (v == "two_launch") return order_two_launch;
            if (v == "release") return order_release;
            precondition(v.empty() || v == "tagged", "VEXCL_REDUCTOR_ORDER: tagged | release | relaxed | two_launch");
            return order_tagged;
        }();
        return m;
    }

    struct reductor_buffers {
        backend::device_vector<char> partials, result, counter;      // counter: how many workgroups of the running reduction have stored their partial (round 4)
        // Round 3: stage 2 stores the scalar straight into host memory the GPU can write (pinned, mapped): the host only
        // waits for the queue -- no copy command between the kernel and the value (the 8-byte read-back was a third of a
        // 2^24-element reduction: 0.066 ms).  The device-side `result` stays for the RCCL combine, which works in place.
        void *pinned = nullptr;
        unsigned long long seq = 0;          // reductions issued on this device (the kernel stores it behind the result)
        reductor_buffers() {}
        reductor_buffers(const reductor_buffers &) = delete;
        reductor_buffers &operator=(const reductor_buffers &) = delete;
        ~reductor_buffers() { if (pinned) (void)vexhip_host_free(pinned); }
    };
}

template <typename ScalarType, class RDC = SUM>
class Reductor {
    public:
        typedef typename std::conditional<std::is_same<RDC, MIN_MAX>::value, cl_vec2<ScalarType>, ScalarType>::type result_type;

        Reductor(const std::vector<backend::command_queue> &queue = current_context().queue()) : queue(queue) {
            for (const auto &q : this->queue) {
                int groups = 0, block = 0;
                backend::check(vexhip_reduce_num_groups(q.device_ordinal(), &groups, &block));
                auto b = std::make_shared<detail::reductor_buffers>();
                // per workgroup: two outputs x two tagged 8-byte words (order `tagged`; the other orders use the front as a plain
                // array of partials).  Zeroed: no word carries the number of a reduction yet (they start at 1)
                b->partials = backend::device_vector<char>(q, (size_t)groups * 32);
                { const std::vector<char> z((size_t)groups * 32, 0); b->partials.write(q, 0, z.size(), z.data(), true); }
                b->result = backend::device_vector<char>(q, 2 * sizeof(ScalarType));
                // arrival counters of the single-launch reduction: [0] the top one, [32 * (1 + k)] one per residue of the workgroup
                // number mod 32, 128 bytes apart (2048 same-address atomics at the end of the kernel took 40 us; 64 do not)
                const std::vector<unsigned> zeros(33 * 32, 0u);
                b->counter = backend::device_vector<char>(q, zeros.size() * sizeof(unsigned));
                b->counter.write(q, 0, zeros.size() * sizeof(unsigned), reinterpret_cast<const char *>(zeros.data()), true);
                backend::check(vexhip_host_alloc(128, &b->pinned));          // [0..1] the scalar(s); byte 64: the number of the reduction that stored them
                std::memset(b->pinned, 0, 128);
                bufs.push_back(b);
                ngroups.push_back(groups);
            }
        }

        /// Reduces the expression; blocking, returns a host value (reductor.hpp:302-439).
        template <class Expr>
        typename std::enable_if<detail::mv_dim<detail::as_expr_t<Expr>>::value == 0, result_type>::type
        operator()(const Expr &expr_) const {
            using namespace detail;
            typedef as_expr_t<Expr> E;
            const E &expr = as_expr<Expr>::get(expr_);
            static_assert(expr_kind<E>::value == 0, "only vector expressions can be reduced");

            prop_context prop;
            expr.get_props(prop);
            if (prop.queue.empty()) prop.queue = queue;
            if (prop.part.empty()) prop.part = vex::partition(prop.size, prop.queue);
            precondition(prop.queue.size() == queue.size(), "expression and Reductor live on different queue lists");
            const std::vector<size_t> &part = prop.part;

            static kernel_cache cache;
            constexpr bool minmax = std::is_same<RDC, MIN_MAX>::value;
            constexpr int nout = minmax ? 2 : 1;
            const int op = op_code();
            const bool two_launch = reductor_order() == order_two_launch;

            std::vector<char> active(queue.size(), 0);
            std::vector<ScalarType> host(queue.size() * 2);
            for (unsigned d = 0; d < queue.size(); ++d) {
                size_t psize = part[d + 1] - part[d];
                if (!psize) continue;
                active[d] = 1;
                auto kernel = cache.find(queue[d]);
                if (kernel == cache.end())
                    kernel = cache.insert(queue[d], backend::kernel(queue[d], source(expr, queue[d]), "vexcl_reductor_kernel"));
                backend::kernel &krn = kernel->second;
                krn.push_arg(psize);
                arg_context a(krn, d, part[d]);
                expr.set_args(a);
                krn.push_arg(bufs[d]->partials.raw());
                // Round 4: ONE launch.  The workgroup that stores its partial last (a counter in device memory tells) folds all of
                // them -- in the order of libvexhip's stage-2 kernel, so the bits are those of the two-launch form -- and stores the
                // scalar (pinned host memory, or the device word the RCCL combine works on): the second launch and the gap in
                // front of it were 5-7 us of a 53 us reduction at 2^24 elements.
                krn.push_arg(reinterpret_cast<unsigned *>(bufs[d]->counter.raw()));
                krn.push_arg(static_cast<ScalarType *>(rccl_combine(minmax) ? static_cast<void *>(bufs[d]->result.raw()) : bufs[d]->pinned));
                krn.push_arg(reinterpret_cast<unsigned long long *>(static_cast<char *>(bufs[d]->pinned) + 64));
                krn.push_arg((unsigned long long)++bufs[d]->seq);
                krn.config(ngroups[d], 256);
                krn(queue[d]);
                if (two_launch)
                    backend::check(vexhip_reduce_finish(queue[d].device_ordinal(), queue[d].raw(), op, reduce_dtype<ScalarType>::value,
                                bufs[d]->partials.raw(), ngroups[d],
                                rccl_combine(minmax) ? static_cast<void *>(bufs[d]->result.raw()) : bufs[d]->pinned));
            }
            (void)op;
            if (rccl_combine(minmax)) {
                // VEXCL_REDUCTOR_COMBINE=rccl: the D per-device scalars are combined by ONE all-reduce over xGMI
                // (vexhip_allreduce_scalar) and a single 8-byte read-back, instead of D read-backs and a host fold
                // (reductor.hpp:412-436).  Used when every device holds a part; the host fold is the default.
                bool all = true;
                for (char a : active) all = all && a;
                if constexpr (!minmax) if (all) {
                    if (!comm) comm = detail::make_comm(queue);
                    std::vector<void *> b(queue.size()), st(queue.size());
                    for (unsigned d = 0; d < queue.size(); ++d) { b[d] = bufs[d]->result.raw(); st[d] = queue[d].raw(); }
                    backend::check(vexhip_allreduce_scalar(comm.get(), op, reduce_dtype<ScalarType>::value, b.data(), 1, st.data()));
                    bufs[0]->result.read(queue[0], 0, sizeof(ScalarType), reinterpret_cast<char *>(&host[0]), true);
                    for (unsigned d = 1; d < queue.size(); ++d) queue[d].finish();
                    return static_cast<result_type>(host[0]);
                }
                for (unsigned d = 0; d < queue.size(); ++d)
                    if (active[d]) bufs[d]->result.read(queue[d], 0, nout * sizeof(ScalarType), reinterpret_cast<char *>(&host[2 * d]), false);
            }
            if (!rccl_combine(minmax)) {
                // the kernel stores the scalar in pinned host memory and, behind it (system-scope release), the number of this
                // reduction: the host watches that word instead of waiting for the queue to drain (the wake-up of a blocked
                // hipStreamSynchronize is several microseconds of a 50 us reduction); a second of silence falls back to the wait
                for (unsigned d = 0; d < queue.size(); ++d) if (active[d]) {
                    if (two_launch) { queue[d].finish(); continue; }
                    const volatile unsigned long long *seen = reinterpret_cast<const volatile unsigned long long *>(static_cast<char *>(bufs[d]->pinned) + 64);
                    const auto t0 = std::chrono::steady_clock::now();
                    unsigned spins = 0;
                    while (*seen != bufs[d]->seq) {
#if defined(__x86_64__) || defined(__i386__)
                        __builtin_ia32_pause();
#elif defined(__aarch64__)
                        asm volatile("yield" ::: "memory");
#endif
                        if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(1)) { queue[d].finish(); break; }
                    }
                    std::atomic_thread_fence(std::memory_order_acquire);
                }
            } else
            for (unsigned d = 0; d < queue.size(); ++d) if (active[d]) queue[d].finish();
            if (!rccl_combine(minmax))
                for (unsigned d = 0; d < queue.size(); ++d)
                    if (active[d]) for (int k = 0; k < nout; ++k) host[2 * d + k] = static_cast<const volatile ScalarType *>(bufs[d]->pinned)[k];
            return combine(host, active, std::integral_constant<bool, minmax>());
        }

        /// Multi-expression: one result per component (reductor.hpp:443-470).
        template <class Expr>
        typename std::enable_if<(detail::mv_dim<detail::as_expr_t<Expr>>::value > 0),
            std::array<result_type, detail::mv_dim<detail::as_expr_t<Expr>>::value>>::type
        operator()(const Expr &expr) const {
            return reduce_components(detail::as_expr<Expr>::get(expr),
                    std::make_index_sequence<detail::mv_dim<detail::as_expr_t<Expr>>::value>());
        }

    private:
        template <class E, size_t... I>
        std::array<result_type, sizeof...(I)> reduce_components(const E &expr, std::index_sequence<I...>) const {
            std::array<result_type, sizeof...(I)> r = {{(*this)(detail::component_of<I, E>::get(expr))...}};
            return r;
        }

        std::vector<backend::command_queue> queue;
        std::vector<std::shared_ptr<detail::reductor_buffers>> bufs;
        std::vector<int> ngroups;
        mutable std::shared_ptr<vexhip_comm> comm;       // created on first use of the RCCL combine

        bool rccl_combine(bool minmax) const {
            static const bool on = [] { const char *e = std::getenv("VEXCL_REDUCTOR_COMBINE"); return e && std::string(e) == "rccl"; }();
            return on && !minmax && queue.size() > 1;
        }

        static int op_code() {
            if (std::is_same<RDC, SUM>::value) return VEXHIP_SUM;
            if (std::is_same<RDC, SUM_Kahan>::value) return VEXHIP_SUM;   // lanes already compensated; fold plainly (reductor.hpp:537-564)
            if (std::is_same<RDC, MAX>::value) return VEXHIP_MAX;
            if (std::is_same<RDC, MIN>::value) return VEXHIP_MIN;
            return VEXHIP_MIN_MAX;
        }

        ScalarType combine(const std::vector<ScalarType> &h, const std::vector<char> &active, std::false_type) const {
            typedef typename std::conditional<std::is_same<RDC, SUM_Kahan>::value, SUM, RDC>::type R;
            typename R::template impl<ScalarType> fn;
            ScalarType r = R::template impl<ScalarType>::initial();
            for (unsigned d = 0; d < active.size(); ++d) if (active[d]) r = fn(r, h[2 * d]);
            return r;
        }
        cl_vec2<ScalarType> combine(const std::vector<ScalarType> &h, const std::vector<char> &active, std::true_type) const {
            cl_vec2<ScalarType> r = {{std::numeric_limits<ScalarType>::max(), std::numeric_limits<ScalarType>::lowest()}};
            for (unsigned d = 0; d < active.size(); ++d) if (active[d]) {
                r.s[0] = std::min(r.s[0], h[2 * d]); r.s[1] = std::max(r.s[1], h[2 * d + 1]);
            }
            return r;
        }

    public:
        /// Text of the reduction kernel of one expression type (stage 1, and stage 2 unless VEXCL_REDUCTOR_ORDER=two_launch).
        template <class E>
        static std::string source(const E &expr, const backend::command_queue &q) {
            using namespace detail;
            const std::string T = type_name<ScalarType>();
            constexpr bool minmax = std::is_same<RDC, MIN_MAX>::value;
            constexpr bool kahan = std::is_same<RDC, SUM_Kahan>::value;
            backend::source_generator src(q);
            { gen_context c(src, q); expr.preamble(c); }
            const reduction_order order = reductor_order();
            if (order == order_tagged) {
                // a partial as tagged words: W = 1 (4-byte value) or 2 (8-byte value) words per value
                constexpr int W = sizeof(ScalarType) == 8 ? 2 : 1;
                static_assert(sizeof(ScalarType) == 4 || sizeof(ScalarType) == 8, "Reductor: 4- or 8-byte scalars");
                const std::string B = W == 2 ? "unsigned long long" : "unsigned";
                src.new_line() << "__device__ __forceinline__ void vex_publish(unsigned long long *w, " << T << " v, unsigned long long tag)";
                src.open("{");
                src.new_line() << B << " b; __builtin_memcpy(&b, &v, sizeof(b));";
                src.new_line() << "__hip_atomic_store(w, (unsigned long long)(unsigned)b | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);";
                if (W == 2) src.new_line() << "__hip_atomic_store(w + 1, (b >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);";
                src.close("}");
                src.new_line() << "__device__ __forceinline__ " << T << " vex_collect(const unsigned long long *w, unsigned long long tag)";
                src.open("{");
                // a word that does not carry this reduction's number yet has not arrived: read it again (the workgroup that owns it
                // has stored it before it was counted; a relaxed atomic load of the same object must return it eventually)
                src.new_line() << "unsigned long long w0 = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);";
                src.new_line() << "while ((w0 >> 32 << 32) != tag) { __builtin_amdgcn_s_sleep(1); w0 = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }";
                if (W == 2) {
                    src.new_line() << "unsigned long long w1 = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);";
                    src.new_line() << "while ((w1 >> 32 << 32) != tag) { __builtin_amdgcn_s_sleep(1); w1 = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }";
                    src.new_line() << B << " b = (w0 & 0xffffffffull) | (w1 << 32);";
                } else {
                    src.new_line() << B << " b = (unsigned)w0;";
                }
                src.new_line() << T << " v; __builtin_memcpy(&v, &b, sizeof(b)); return v;";
                src.close("}");
            }
            src.begin_kernel("vexcl_reductor_kernel");
            src.begin_kernel_parameters();
            src.template parameter<size_t>("n");
            { gen_context c(src, q); expr.params(c); }
            src.template parameter<global_ptr<ScalarType>>("g_odata");
            src.template parameter<global_ptr<unsigned>>("g_counter");
            src.template parameter<global_ptr<ScalarType>>("g_result");
            src.template parameter<global_ptr<cl_ulong>>("g_seen");
            src.template parameter<cl_ulong>("seq");
            src.end_kernel_parameters();
            constexpr int TW = sizeof(ScalarType) == 8 ? 2 : 1;                 // tagged words per value
            if (order == order_tagged) {
                src.new_line() << "unsigned long long *g_words = (unsigned long long *)g_odata;";
                src.new_line() << "const unsigned long long vex_tag = seq << 32;";
            }

            auto fold = [&](const std::string &a, const std::string &b) -> std::string {
                if (std::is_same<RDC, MAX>::value) return MAX::impl<ScalarType>::device(a, b);
                if (std::is_same<RDC, MIN>::value) return MIN::impl<ScalarType>::device(a, b);
                return SUM::impl<ScalarType>::device(a, b);
            };

            if (minmax) {
                src.new_line() << T << " myMin = " << literal(std::numeric_limits<ScalarType>::max())
                               << ", myMax = " << literal(std::numeric_limits<ScalarType>::lowest()) << ";";
            } else if (kahan) {
                src.new_line() << T << " mySum = (" << T << ")0, c = (" << T << ")0;";
            } else {
                typedef typename std::conditional<minmax, SUM, RDC>::type R;
                src.new_line() << T << " mySum = " << literal(R::template impl<ScalarType>::initial()) << ";";
            }
            // One element per trip, on purpose: tools/r02_reduce_ablation.py (profiles/r02_reduce_ablation.json) -- sum(a*b),
            // 1e8 fp64, this loop 0.255 ms = 6.28 TB/s; 2 / 4 / 8 strided elements per trip with their loads issued first
            // 0.293 / 0.292 / 0.287 ms; 16-byte loads 0.267-0.274 ms; a contiguous chunk per workgroup 0.263-0.269 ms.
            src.grid_stride_loop().open("{");
            { gen_context c(src, q); expr.local_init(c); }
            src.new_line() << T << " v = ";
            { gen_context c(src, q); expr.emit(c); }
            src << ";";
            if (minmax) {
                src.new_line() << "myMin = v < myMin ? v : myMin;";
                src.new_line() << "myMax = v > myMax ? v : myMax;";
            } else if (kahan) {
                src.new_line() << T << " y = v - c;";
                src.new_line() << T << " t = mySum + y;";
                src.new_line() << "c = (t - mySum) - y;";
                src.new_line() << "mySum = t;";
            } else {
                src.new_line() << "mySum = " << fold("mySum", "v") << ";";
            }
            src.close("}");

            // wave-64 shuffle fold, then one LDS hop across the waves of the workgroup
            auto wave_fold = [&](const std::string &var, const std::string &kind) {
                src.new_line() << "for (int o = 32; o > 0; o >>= 1)";
                src.open("{");
                src.new_line() << T << " other = __shfl_down(" << var << ", o, 64);";
                if (kind == "min") src.new_line() << var << " = other < " << var << " ? other : " << var << ";";
                else if (kind == "max") src.new_line() << var << " = other > " << var << " ? other : " << var << ";";
                else src.new_line() << var << " = " << fold(var, "other") << ";";
                src.close("}");
            };
            const int nout = minmax ? 2 : 1;
            src.new_line() << "__shared__ " << T << " sdata[" << 16 * nout << "];";
            src.new_line() << "const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;";
            if (minmax) {
                wave_fold("myMin", "min"); wave_fold("myMax", "max");
                src.new_line() << "if (lane == 0) { sdata[2 * wave] = myMin; sdata[2 * wave + 1] = myMax; }";
                src.new_line() << "__syncthreads();";
                src.new_line() << "if (threadIdx.x == 0)";
                src.open("{");
                src.new_line() << "for (int w = 1; w < nwaves; ++w)";
                src.open("{");
                src.new_line() << "myMin = sdata[2 * w] < myMin ? sdata[2 * w] : myMin;";
                src.new_line() << "myMax = sdata[2 * w + 1] > myMax ? sdata[2 * w + 1] : myMax;";
                src.close("}");
                if (order == order_relaxed) {
                    src.new_line() << T << " prev0 = __hip_atomic_exchange(&g_odata[2 * blockIdx.x], myMin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);";
                    src.new_line() << T << " prev1 = __hip_atomic_exchange(&g_odata[2 * blockIdx.x + 1], myMax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);";
                    src.new_line() << "asm volatile(\"\" :: \"v\"(prev0), \"v\"(prev1) : \"memory\");";
                } else if (order == order_tagged) {
                    src.new_line() << "vex_publish(g_words + " << 2 * TW << " * blockIdx.x, myMin, vex_tag);";
                    src.new_line() << "vex_publish(g_words + " << 2 * TW << " * blockIdx.x + " << TW << ", myMax, vex_tag);";
                } else {
                    src.new_line() << "__hip_atomic_store(&g_odata[2 * blockIdx.x], myMin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);";
                    src.new_line() << "__hip_atomic_store(&g_odata[2 * blockIdx.x + 1], myMax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);";
                }
                src.close("}");
            } else {
                wave_fold("mySum", "fold");
                src.new_line() << "if (lane == 0) sdata[wave] = mySum;";
                src.new_line() << "__syncthreads();";
                src.new_line() << "if (threadIdx.x == 0)";
                src.open("{");
                src.new_line() << "for (int w = 1; w < nwaves; ++w) mySum = " << fold("mySum", "sdata[w]") << ";";
                if (order == order_relaxed) {
                    // (round 4's form, VEXCL_REDUCTOR_ORDER=relaxed: an atomic exchange at agent scope is performed where every XCD
                    //  sees it and its return tells when; no fence.  Relies on gfx942/gfx950 behaviour -- a returning sc1 atomic has
                    //  been performed at the memory side -- that the HIP memory model does not promise: kept for A/B only)
                    src.new_line() << T << " prev = __hip_atomic_exchange(&g_odata[blockIdx.x], mySum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);";
                    src.new_line() << "asm volatile(\"\" :: \"v\"(prev) : \"memory\");";
                } else if (order == order_tagged) {
                    src.new_line() << "vex_publish(g_words + " << TW << " * blockIdx.x, mySum, vex_tag);";
                } else {
                    // (order release: the partial is ordered before the arrival count by the release fence in front of that count)
                    src.new_line() << "__hip_atomic_store(&g_odata[blockIdx.x], mySum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);";
                }
                src.close("}");
            }
            if (order == order_two_launch) { src.end_kernel(); return src.str(); }
            // ---- stage 2 inside the same launch: the workgroup that arrives last folds the partials (order of reduce_stage2 in
            // libvexhip: lane t takes partials t, t + 256, ...; wave fold; waves in order) ----
            src.new_line() << "__shared__ int s_last;";
            src.new_line() << "if (threadIdx.x == 0)";
            src.open("{");
            src.new_line() << "const unsigned sub = blockIdx.x & 31u, members = (gridDim.x - sub + 31u) >> 5, groups = gridDim.x < 32u ? gridDim.x : 32u;";
            src.new_line() << "int last = 0;";
            if (order == order_relaxed || order == order_tagged) {
                // (order tagged: the counters only elect the workgroup that folds; nothing about the partials is inferred from them)
                src.new_line() << "if (atomicAdd(g_counter + 32u * (1u + sub), 1u) == members - 1u)";      // the last of its residue class ...
                src.open("{");
                src.new_line() << "g_counter[32u * (1u + sub)] = 0u;";
                src.new_line() << "last = atomicAdd(g_counter, 1u) == groups - 1u;";                       // ... counts for the class; the last class closes the reduction
                src.close("}");
            } else {
                // Ordered by the memory model (round 5): every arrival RELEASEs its partial on the class counter; the last of a
                // class ACQUIREs the others' (fence, executed by 32 workgroups per launch) and passes everything on with a
                // second release in front of its arrival on the top counter; the last of all acquires and has every partial.
                // (release FENCE + relaxed arrival, and the wait behind the write-back restated in asm: ROCm 7.2 drops the
                //  s_waitcnt after buffer_wbl2 when its scoreboard says nothing of this wave is outstanding, and the arrival
                //  could overtake the write-back -- MI355X_MICROARCH.md, inter-workgroup visibility)
                src.new_line() << "__builtin_amdgcn_fence(__ATOMIC_RELEASE, \"agent\");";
                src.new_line() << "asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");";
                src.new_line() << "if (__hip_atomic_fetch_add(g_counter + 32u * (1u + sub), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u)";
                src.open("{");
                src.new_line() << "__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"agent\");";
                src.new_line() << "__hip_atomic_store(g_counter + 32u * (1u + sub), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);";
                src.new_line() << "__builtin_amdgcn_fence(__ATOMIC_RELEASE, \"agent\");";
                src.new_line() << "asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");";
                src.new_line() << "last = __hip_atomic_fetch_add(g_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1u;";
                src.new_line() << "if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"agent\");";
                src.close("}");
            }
            src.new_line() << "s_last = last;";
            src.close("}");
            src.new_line() << "__syncthreads();";
            src.new_line() << "if (s_last)";
            src.open("{");
            // every lane of the closing workgroup reads partials: each acquires for itself (once per launch, one workgroup)
            if (order == order_release) src.new_line() << "__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"agent\");";
            if (minmax) {
                src.new_line() << "myMin = " << literal(std::numeric_limits<ScalarType>::max()) << "; myMax = " << literal(std::numeric_limits<ScalarType>::lowest()) << ";";
                src.new_line() << "for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x)";
                src.open("{");
                if (order == order_tagged)
                    src.new_line() << T << " a = vex_collect(g_words + " << 2 * TW << " * i, vex_tag), b = vex_collect(g_words + " << 2 * TW << " * i + " << TW << ", vex_tag);";
                else
                src.new_line() << T << " a = __hip_atomic_load(&g_odata[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(&g_odata[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);";
                src.new_line() << "myMin = a < myMin ? a : myMin; myMax = b > myMax ? b : myMax;";
                src.close("}");
                wave_fold("myMin", "min"); wave_fold("myMax", "max");
                src.new_line() << "if (lane == 0) { sdata[2 * wave] = myMin; sdata[2 * wave + 1] = myMax; }";
                src.new_line() << "__syncthreads();";
                src.new_line() << "if (threadIdx.x == 0)";
                src.open("{");
                src.new_line() << "for (int w = 1; w < nwaves; ++w)";
                src.open("{");
                src.new_line() << "myMin = sdata[2 * w] < myMin ? sdata[2 * w] : myMin;";
                src.new_line() << "myMax = sdata[2 * w + 1] > myMax ? sdata[2 * w + 1] : myMax;";
                src.close("}");
                src.new_line() << "g_result[0] = myMin; g_result[1] = myMax; *g_counter = 0u;";
                src.new_line() << "__hip_atomic_store(g_seen, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);";
                src.close("}");
            } else {
                typedef typename std::conditional<minmax || kahan, SUM, RDC>::type R2;
                src.new_line() << "mySum = " << literal(R2::template impl<ScalarType>::initial()) << ";";
                if (order == order_tagged)
                    src.new_line() << "for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x) mySum = " << fold("mySum", "vex_collect(g_words + " + std::to_string(TW) + " * i, vex_tag)") << ";";
                else
                src.new_line() << "for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x) mySum = " << fold("mySum", "__hip_atomic_load(&g_odata[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)") << ";";
                wave_fold("mySum", "fold");
                src.new_line() << "if (lane == 0) sdata[wave] = mySum;";
                src.new_line() << "__syncthreads();";
                src.new_line() << "if (threadIdx.x == 0)";
                src.open("{");
                src.new_line() << "for (int w = 1; w < nwaves; ++w) mySum = " << fold("mySum", "sdata[w]") << ";";
                src.new_line() << "g_result[0] = mySum; *g_counter = 0u;";
                src.new_line() << "__hip_atomic_store(g_seen, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);";
                src.close("}");
            }
            src.close("}");
            src.end_kernel();
            return src.str();
        }
};

} // namespace vex
#endif
