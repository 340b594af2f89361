#ifndef VEXCL_DEVLIST_HPP
#define VEXCL_DEVLIST_HPP
// vex::Filter::* and vex::Context (reference: vexcl/devlist.hpp:53-391).
#include <cctype>
#include <cstdlib>
#include <iostream>
#include <functional>
#include <string>
#include <vector>
#include "backend.hpp"
#include "cache.hpp"

namespace vex {
namespace Filter {

struct AnyFilter { bool operator()(const backend::device &) const { return true; } };
const AnyFilter Any = {};
const AnyFilter All = {};

/// Accepts devices with double precision support (every CDNA part).
struct DoublePrecisionFilter { bool operator()(const backend::device &) const { return true; } };
const DoublePrecisionFilter DoublePrecision = {};

struct GPUFilter { bool operator()(const backend::device &) const { return true; } };
const GPUFilter GPU = {};
struct CPUFilter { bool operator()(const backend::device &) const { return false; } };
const CPUFilter CPU = {};
const CPUFilter Accelerator = {};

/// Vendor / platform name contains the given string: every device here is an AMD GPU on the HIP platform
/// (backend/opencl/filter.hpp:62-84).
struct Vendor {
    explicit Vendor(std::string v) : name(std::move(v)) {}
    bool operator()(const backend::device &) const { return std::string("Advanced Micro Devices, Inc. (AMD)").find(name) != std::string::npos; }
    std::string name;
};
struct Platform {
    explicit Platform(std::string v) : name(std::move(v)) {}
    bool operator()(const backend::device &) const { return std::string("AMD HIP / ROCm").find(name) != std::string::npos; }
    std::string name;
};

/// OpenCL extension by name (backend/opencl/filter.hpp:144-160).  There are no OpenCL extensions here; the two
/// capability names user code asks about are answered by what the hardware does: fp64 and 64-bit atomics exist.
struct Extension {
    explicit Extension(std::string e) : extension(std::move(e)) {}
    bool operator()(const backend::device &) const {
        return extension == "cl_khr_fp64" || extension == "cl_khr_int64_base_atomics" || extension == "cl_khr_int64_extended_atomics";
    }
    std::string extension;
};
const Extension GLSharing("cl_khr_gl_sharing");
/// OpenCL version (backend/opencl/filter.hpp:165-180): asked for by code that wants OpenCL 2.0 features -- shared
/// virtual memory (vex::svm_vector: hipMallocManaged) and generic address spaces, both present.
struct CLVersion {
    CLVersion(int major, int minor) : major_(major), minor_(minor) {}
    bool operator()(const backend::device &) const { return major_ < 2 || (major_ == 2 && minor_ <= 0); }
    int major_, minor_;
};

/// Compute capability at least major.minor (backend/cuda/filter.hpp:71-83).  For an AMD GPU the corresponding
/// number is the ISA version: gfx950 is 9.5, gfx942 is 9.4, gfx1100 is 11.0.
struct CC {
    CC(int major, int minor) : major_(major), minor_(minor) {}
    bool operator()(const backend::device &d) const {
        const std::string a = d.arch();                       // "gfx950:sramecc+:xnack-"
        size_t b = a.find("gfx"), e = b == std::string::npos ? 0 : b + 3;
        while (e < a.size() && std::isalnum((unsigned char)a[e])) ++e;
        if (b == std::string::npos || e - (b + 3) < 3) return false;
        const std::string v = a.substr(b + 3, e - (b + 3));
        const int major = std::atoi(v.substr(0, v.size() - 2).c_str());
        const int minor = std::isdigit((unsigned char)v[v.size() - 2]) ? v[v.size() - 2] - '0' : 10 + (std::tolower(v[v.size() - 2]) - 'a');
        return major > major_ || (major == major_ && minor >= minor_);
    }
    int major_, minor_;
};

/// Device name contains the given string (devlist.hpp Filter::Name).
struct Name {
    explicit Name(std::string name) : devname(std::move(name)) {}
    bool operator()(const backend::device &d) const { return d.name().find(devname) != std::string::npos; }
    std::string devname;
};

/// At most n devices (devlist.hpp:76-89).
struct Count {
    explicit Count(int c) : count(c) {}
    bool operator()(const backend::device &) const { return --count >= 0; }
    mutable int count;
};

/// The device at the given position among those seen (devlist.hpp:91-106).
struct Position {
    explicit Position(int p) : pos(p) {}
    bool operator()(const backend::device &) const { return 0 == pos--; }
    mutable int pos;
};

/// Type-erased filter (devlist.hpp General).
struct General {
    template <class F> General(F f) : filter(f) {}
    bool operator()(const backend::device &d) const { return filter(d); }
    std::function<bool(const backend::device &)> filter;
};

template <class L, class R> struct AndFilter {
    L l; R r;
    bool operator()(const backend::device &d) const { return l(d) && r(d); }
};
template <class L, class R> struct OrFilter {
    L l; R r;
    bool operator()(const backend::device &d) const { return l(d) || r(d); }
};
template <class F> struct NotFilter {
    F f;
    bool operator()(const backend::device &d) const { return !f(d); }
};

template <class T> struct is_filter : std::false_type {};
template <> struct is_filter<AnyFilter> : std::true_type {};
template <> struct is_filter<DoublePrecisionFilter> : std::true_type {};
template <> struct is_filter<GPUFilter> : std::true_type {};
template <> struct is_filter<CPUFilter> : std::true_type {};
template <> struct is_filter<Name> : std::true_type {};
template <> struct is_filter<Extension> : std::true_type {};
template <> struct is_filter<CC> : std::true_type {};
template <> struct is_filter<CLVersion> : std::true_type {};
template <> struct is_filter<Vendor> : std::true_type {};
template <> struct is_filter<Platform> : std::true_type {};
template <> struct is_filter<Count> : std::true_type {};
template <> struct is_filter<Position> : std::true_type {};
template <> struct is_filter<General> : std::true_type {};
template <class L, class R> struct is_filter<AndFilter<L, R>> : std::true_type {};
template <class L, class R> struct is_filter<OrFilter<L, R>> : std::true_type {};
template <class F> struct is_filter<NotFilter<F>> : std::true_type {};

template <class L, class R>
typename std::enable_if<is_filter<L>::value && is_filter<R>::value, AndFilter<L, R>>::type
operator&&(const L &l, const R &r) { return AndFilter<L, R>{l, r}; }
template <class L, class R>
typename std::enable_if<is_filter<L>::value && is_filter<R>::value, OrFilter<L, R>>::type
operator||(const L &l, const R &r) { return OrFilter<L, R>{l, r}; }
template <class F>
typename std::enable_if<is_filter<F>::value, NotFilter<F>>::type
operator!(const F &f) { return NotFilter<F>{f}; }

/// Environment filter: OCL_DEVICE (name substring), OCL_MAX_DEVICES,
/// OCL_POSITION (devlist.hpp:185-205).
struct EnvFilter {
    EnvFilter() : name(nullptr), max_dev(-1), position(-1), seen(0) {
        name = std::getenv("OCL_DEVICE");
        if (const char *m = std::getenv("OCL_MAX_DEVICES")) max_dev = std::atoi(m);
        if (const char *p = std::getenv("OCL_POSITION")) position = std::atoi(p);
    }
    bool operator()(const backend::device &d) const {
        if (name && d.name().find(name) == std::string::npos) return false;
        int my = seen++;
        if (position >= 0 && my != position) return false;
        if (max_dev >= 0 && --max_dev < 0) return false;
        return true;
    }
    const char *name; mutable int max_dev; int position; mutable int seen;
};
template <> struct is_filter<EnvFilter> : std::true_type {};
const EnvFilter Env;

/// Exclusive access to the selected devices across the processes of a node
/// (backend/opencl/filter.hpp:203-320, there with Boost.Interprocess file locks): a device passes only if
/// this process obtains the advisory lock on `$VEXCL_LOCK_DIR/vexcl_device_<ordinal>.lock` (default /tmp);
/// the lock is held until the process exits.  One process per GPU is the deployment model on an 8-GPU node:
/// `Context ctx(Filter::Exclusive(Filter::Env && Filter::Count(1)))` gives every process its own GPU.
template <class F>
struct ExclusiveFilter {
    F filter;
    bool operator()(const backend::device &d) const { return filter(d) && backend::lock_device(d.raw()); }
};
template <class F> struct is_filter<ExclusiveFilter<F>> : std::true_type {};
template <class F>
typename std::enable_if<is_filter<typename std::decay<F>::type>::value, ExclusiveFilter<typename std::decay<F>::type>>::type
Exclusive(F &&f) { return ExclusiveFilter<typename std::decay<F>::type>{std::forward<F>(f)}; }

} // namespace Filter

class Context;

template <bool dummy = true>
class StaticContext {
    public:
        static void set(Context &ctx) { instance = &ctx; }
        static const Context &get() {
            precondition(instance != 0, "Uninitialized static context");
            return *instance;
        }
    private:
        static Context *instance;
};
template <bool dummy> Context *StaticContext<dummy>::instance = 0;

inline const Context &current_context() { return StaticContext<>::get(); }

/// VexCL context: the contexts and command queues of the selected devices
/// (devlist.hpp:273-391).
class Context {
    public:
        template <class DevFilter, class = typename std::enable_if<
            std::is_convertible<decltype(std::declval<DevFilter &>()(std::declval<const backend::device &>())), bool>::value>::type>
        explicit Context(DevFilter &&filter, backend::command_queue_properties properties = 0) {
            std::tie(c, q) = backend::queue_list(std::forward<DevFilter>(filter), properties);
#ifdef VEXCL_THROW_ON_EMPTY_CONTEXT
            precondition(!q.empty(), "No compute devices found");
#endif
            StaticContext<>::set(*this);
        }

        Context(const std::vector<std::pair<backend::context, backend::command_queue>> &user_ctx) {
            for (const auto &u : user_ctx) { c.push_back(u.first); q.push_back(u.second); }
            StaticContext<>::set(*this);
        }

        /// Wraps contexts and queues the user already holds (devlist.hpp:300-310).
        Context(std::vector<backend::context> contexts, std::vector<backend::command_queue> queues)
            : c(std::move(contexts)), q(std::move(queues))
        {
            precondition(c.size() == q.size(), "Context: as many contexts as queues are expected");
            StaticContext<>::set(*this);
        }

        ~Context() { purge_caches(q); }

        const std::vector<backend::context> &context() const { return c; }
        const backend::context &context(unsigned d) const { return c[d]; }
        const std::vector<backend::command_queue> &queue() const { return q; }
        operator const std::vector<backend::command_queue> &() const { return q; }
        const backend::command_queue &queue(unsigned d) const { return q[d]; }
        backend::device device(unsigned d) const { return q[d].device(); }
        size_t size() const { return q.size(); }
        bool empty() const { return q.empty(); }
        operator bool() const { return !empty(); }
        void finish() const { for (const auto &queue : q) queue.finish(); }
    private:
        std::vector<backend::context> c;
        std::vector<backend::command_queue> q;
};

namespace backend {      // next to the types they print, so that `std::cout << device` finds them from any namespace
inline std::ostream &operator<<(std::ostream &os, const device &d) {
    return os << d.name() << " (" << d.arch() << ", " << d.multiprocessor_count() << " CUs)";
}
inline std::ostream &operator<<(std::ostream &os, const command_queue &q) { return os << q.device(); }
}
inline std::ostream &operator<<(std::ostream &os, const std::vector<backend::command_queue> &queue) {
    unsigned p = 0;
    for (const auto &q : queue) os << ++p << ". " << q << std::endl;
    return os;
}
inline std::ostream &operator<<(std::ostream &os, const Context &ctx) { return os << ctx.queue(); }

} // namespace vex
#endif
