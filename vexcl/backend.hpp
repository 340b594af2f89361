#ifndef VEXCL_BACKEND_HPP
#define VEXCL_BACKEND_HPP
// The single backend of this implementation: HIP on gfx950 through the C ABI of
// libvexhip.so (include/vexhip.h).  It provides the names of the reference's
// compile-time backend concept (vexcl/backend.hpp:40-96; concrete specimen
// backend/cuda/{context,device_vector,kernel,source,compiler,event,error}.hpp)
// so that upper layers and user code (tests/custom_kernel.cpp style) compile
// unchanged.  No HIP headers are needed here: host C++17 only.
#include <cstring>
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include <functional>
#include <atomic>

#include "../include/vexhip.h"
#include "util.hpp"
#include "types.hpp"

namespace vex {
namespace backend {

// ---- errors (backend/cuda/error.hpp:119-156) --------------------------------
class error : public std::runtime_error {
    public:
        explicit error(const std::string &msg, int hip_code = 0) : std::runtime_error(msg), code_(hip_code) {}
        /// the hipError_t behind the failure (0: a check of the library's own)
        int code() const { return code_; }
        bool out_of_memory() const { return code_ == VEXHIP_ERROR_OUT_OF_MEMORY; }
    private:
        int code_;
};

inline void check(int rc) {
    if (rc != 0) throw error(vexhip_last_error(), vexhip_last_error_code());
}

inline std::ostream &operator<<(std::ostream &os, const error &e) {
    return os << "HIP error: " << e.what();
}

// ---- device / context / queue (backend/cuda/context.hpp:96-413) -------------
class device {
    public:
        device() : id_(-1) {}
        explicit device(int id) : id_(id) {
            check(vexhip_device_get_props(id, &props_));
        }
        int raw() const { return id_; }
        std::string name() const { return props_.name; }
        std::string arch() const { return props_.arch; }
        size_t multiprocessor_count() const { return props_.compute_units; }
        size_t max_threads_per_block() const { return props_.max_threads_per_block; }
        size_t max_shared_memory_per_block() const { return props_.lds_bytes_per_block; }
        size_t wavefront_size() const { return props_.wavefront_size; }
        size_t global_mem_size() const { return props_.global_mem_bytes; }
        bool operator==(const device &o) const { return id_ == o.id_; }
    private:
        int id_;
        vexhip_device_props props_{};
};

typedef int device_id;
typedef size_t context_id;

/// One logical context = one device handle with a unique id.  Two contexts may
/// sit on the same physical GPU (the reference's test fixture builds its
/// 2-"device" context exactly so, tests/context_setup.hpp:24-39).
class context {
    public:
        context() {}
        explicit context(const device &d) : p_(std::make_shared<impl>(d)) {}
        const device &dev() const { return p_->dev; }
        context_id id() const { return p_ ? p_->id : 0; }
        void set_current() const {}
        bool operator==(const context &o) const { return id() == o.id(); }
        bool operator<(const context &o) const { return id() < o.id(); }
    private:
        struct impl {
            device dev; context_id id;
            explicit impl(const device &d) : dev(d) { static std::atomic<size_t> next{1}; id = next++; }
        };
        std::shared_ptr<impl> p_;
};

typedef unsigned command_queue_properties;

class command_queue {
    public:
        command_queue() {}
        command_queue(const backend::context &ctx, const backend::device &dev, command_queue_properties flags = 0)
            : p_(std::make_shared<impl>(ctx, dev, flags)) {}

        void finish() const { check(vexhip_stream_sync(p_->dev.raw(), p_->stream)); }
        const backend::context &context() const { return p_->ctx; }
        const backend::device &device() const { return p_->dev; }
        command_queue_properties flags() const { return p_->flags; }
        void *raw() const { return p_->stream; }
        int device_ordinal() const { return p_->dev.raw(); }
        size_t id() const { return reinterpret_cast<size_t>(p_.get()); }
        bool operator==(const command_queue &o) const { return p_ == o.p_; }
        explicit operator bool() const { return (bool)p_; }
    private:
        struct impl {
            backend::context ctx; backend::device dev; command_queue_properties flags; void *stream = nullptr;
            impl(const backend::context &c, const backend::device &d, command_queue_properties f)
                : ctx(c), dev(d), flags(f) { check(vexhip_stream_create(d.raw(), &stream)); }
            ~impl() { if (stream) vexhip_stream_destroy(dev.raw(), stream); }
        };
        std::shared_ptr<impl> p_;
};

inline void select_context(const command_queue &) {}
inline device get_device(const command_queue &q) { return q.device(); }
inline device_id get_device_id(const command_queue &q) { return q ? q.device().raw() : -1; }
inline context get_context(const command_queue &q) { return q.context(); }
inline context_id get_context_id(const command_queue &q) { return q.context().id(); }
inline command_queue duplicate_queue(const command_queue &q) {
    return command_queue(q.context(), q.device(), q.flags());
}
inline bool is_cpu(const command_queue &) { return false; }

struct compare_contexts {
    bool operator()(const context &a, const context &b) const { return a.id() < b.id(); }
};
struct compare_queues {
    bool operator()(const command_queue &a, const command_queue &b) const { return a.id() < b.id(); }
};

// ---- events (backend/cuda/event.hpp:51-124) ---------------------------------
class event {
    public:
        event() {}
        explicit event(const command_queue &q) : p_(std::make_shared<impl>(q)) {
            check(vexhip_event_record(q.device_ordinal(), p_->e, q.raw()));
        }
        void wait() const { if (p_) check(vexhip_event_sync(p_->dev, p_->e)); }
        void *raw() const { return p_ ? p_->e : nullptr; }
        int device_ordinal() const { return p_ ? p_->dev : -1; }
    private:
        struct impl {
            int dev; void *e = nullptr;
            explicit impl(const command_queue &q) : dev(q.device_ordinal()) { check(vexhip_event_create(dev, 0, &e)); }
            ~impl() { if (e) vexhip_event_destroy(dev, e); }
        };
        std::shared_ptr<impl> p_;
};
typedef std::vector<event> wait_list;

inline event enqueue_marker(const command_queue &q) { return event(q); }
inline event enqueue_barrier(const command_queue &q, const wait_list &events) {
    for (const auto &e : events)
        if (e.raw()) check(vexhip_stream_wait_event(q.device_ordinal(), q.raw(), e.raw()));
    return event(q);
}
inline void wait_for_events(const wait_list &events) { for (const auto &e : events) e.wait(); }

// ---- device_vector<T> (backend/cuda/device_vector.hpp:66-214) ---------------
typedef unsigned mem_flags;
static const mem_flags MEM_READ_ONLY = 1, MEM_WRITE_ONLY = 2, MEM_READ_WRITE = 4;

template <class T>
class device_vector {
    public:
        typedef T value_type;
        typedef T *raw_type;

        device_vector() : n_(0) {}

        template <class H>
        device_vector(const command_queue &q, size_t n, const H *host = 0, mem_flags = MEM_READ_WRITE)
            : n_(n)
        {
            if (n) {
                void *p = nullptr;
                int dev = q.device_ordinal();
                check(vexhip_malloc(dev, n * sizeof(T), &p));
                buf_.reset(static_cast<char *>(p), [dev](char *ptr) { vexhip_free(dev, ptr); });
                if (host) {
                    if (std::is_same<H, T>::value)
                        write(q, 0, n, reinterpret_cast<const T *>(host), true);
                    else {
                        std::vector<T> tmp(host, host + n);
                        write(q, 0, n, tmp.data(), true);
                    }
                }
            }
        }
        device_vector(const command_queue &q, size_t n) : device_vector(q, n, static_cast<const T *>(0)) {}

        /// Wraps a raw device pointer owned by someone else (vector.hpp raw-buffer ctor).
        static device_vector wrap(T *ptr, size_t n) {
            device_vector v; v.n_ = n; v.buf_.reset(reinterpret_cast<char *>(ptr), [](char *) {}); return v;
        }

        /// Takes ownership of a device buffer allocated by libvexhip (released with vexhip_free).
        static device_vector adopt(const command_queue &q, T *ptr, size_t n) {
            device_vector v; v.n_ = n; int dev = q.device_ordinal();
            v.buf_.reset(reinterpret_cast<char *>(ptr), [dev](char *p) { vexhip_free(dev, p); });
            return v;
        }

        void write(const command_queue &q, size_t offset, size_t size, const T *host, bool blocking = false) const {
            if (size) check(vexhip_memcpy_h2d(q.device_ordinal(), raw() + offset, host, size * sizeof(T), q.raw(), blocking));
        }
        void read(const command_queue &q, size_t offset, size_t size, T *host, bool blocking = false) const {
            if (size) check(vexhip_memcpy_d2h(q.device_ordinal(), host, raw() + offset, size * sizeof(T), q.raw(), blocking));
        }
        size_t size() const { return n_; }

        struct buffer_unmapper {
            const command_queue &queue; const device_vector &buffer;
            buffer_unmapper(const command_queue &q, const device_vector &b) : queue(q), buffer(b) {}
            void operator()(T *ptr) const { buffer.write(queue, 0, buffer.size(), ptr, true); delete[] ptr; }
        };
        typedef std::unique_ptr<T[], buffer_unmapper> mapped_array;
        mapped_array map(const command_queue &q) const {
            mapped_array ptr(new T[n_], buffer_unmapper(q, *this));
            read(q, 0, n_, ptr.get(), true);
            return ptr;
        }

        T *raw() const { return reinterpret_cast<T *>(buf_.get()); }
        const void *raw_ptr() const { return buf_.get(); }

        template <class U> device_vector<U> reinterpret() const {
            device_vector<U> r; r.n_ = n_ * sizeof(T) / sizeof(U); r.buf_ = buf_; return r;
        }
        bool operator==(const device_vector &o) const { return buf_ == o.buf_; }
    private:
        template <class U> friend class device_vector;
        size_t n_;
        std::shared_ptr<char> buf_;
};

// ---- source_generator (backend/cuda/source.hpp:45-292) ----------------------
template <class T> struct global_ptr {};
template <class T> struct shared_ptr {};
template <class T> struct regstr_ptr {};
template <class T> struct constant_ptr {};

} // namespace backend

using backend::global_ptr;
using backend::shared_ptr;
using backend::regstr_ptr;
using backend::constant_ptr;

template <class T> struct type_name_impl< backend::global_ptr<T> > {
    static std::string get() { return type_name<T>() + " *"; }
};
template <class T> struct type_name_impl< backend::global_ptr<const T> > {
    static std::string get() { return "const " + type_name<T>() + " *"; }
};
/// OpenCL's __constant pointers: read-only, non-aliasing data (scalar loads where the index is wave-uniform).
template <class T> struct type_name_impl< backend::constant_ptr<T> > {
    static std::string get() { return "const " + type_name<T>() + " * __restrict__"; }
};
template <class T> struct type_name_impl< backend::shared_ptr<T> > {
    static std::string get() { return type_name<T>() + " *"; }
};
template <class T> struct type_name_impl< backend::regstr_ptr<T> > {
    static std::string get() { return type_name<T>() + " *"; }
};

namespace backend {

inline std::string standard_kernel_header(const command_queue &);

class source_generator {
    public:
        source_generator() : indent(0), first_prm(true) {}
        explicit source_generator(const command_queue &q, bool include_standard_header = true)
            : indent(0), first_prm(true)
        {
            if (include_standard_header) src << standard_kernel_header(q);
        }

        source_generator &new_line() { src << "\n" << std::string(2 * indent, ' '); return *this; }
        source_generator &open(const char *bracket) { new_line() << bracket; ++indent; return *this; }
        source_generator &close(const char *bracket) { --indent; new_line() << bracket; return *this; }

        source_generator &begin_function(const std::string &return_type, const std::string &name) {
            new_line() << "__device__ " << return_type << " " << name;
            return *this;
        }
        template <class Return> source_generator &begin_function(const std::string &name) {
            return begin_function(type_name<Return>(), name);
        }
        source_generator &begin_function_parameters() { first_prm = true; return open("("); }
        source_generator &end_function_parameters() { return close(")").open("{"); }
        source_generator &end_function() { return close("}"); }

        source_generator &begin_kernel(const std::string &name) {
            new_line() << "extern \"C\" __global__ void " << name;
            return *this;
        }
        source_generator &begin_kernel_parameters() { first_prm = true; return open("("); }
        source_generator &end_kernel_parameters() { return close(")").open("{"); }
        source_generator &end_kernel() { return close("}"); }

        source_generator &parameter(const std::string &prm_type, const std::string &name) {
            prm_separator().new_line() << prm_type << " " << name;
            return *this;
        }
        template <class Prm> source_generator &parameter(const std::string &name) {
            return parameter(type_name<typename std::decay<Prm>::type>(), name);
        }
        template <class Prm> source_generator &smem_parameter(const std::string & = "smem") { return *this; }
        template <class Prm> source_generator &smem_declaration(const std::string &name = "smem") {
            new_line() << "extern __shared__ __attribute__((aligned(16))) char vex_dyn_smem[];";
            new_line() << type_name<Prm>() << " *" << name << " = (" << type_name<Prm>() << " *)vex_dyn_smem;";
            return *this;
        }
        source_generator &smem_static_var(const std::string &type, const std::string &name) {
            new_line() << "__shared__ " << type << " " << name << ";";
            return *this;
        }

        source_generator &grid_stride_loop(const std::string &idx = "idx", const std::string &bnd = "n") {
            new_line() << "for";
            open("(");
            new_line() << "ulong " << idx << " = blockDim.x * (ulong)blockIdx.x + threadIdx.x, "
                          "grid_size = blockDim.x * (ulong)gridDim.x;";
            new_line() << idx << " < " << bnd << ";";
            new_line() << idx << " += grid_size";
            close(")");
            return *this;
        }

        source_generator &barrier(bool /*global*/ = false) { src << "__syncthreads();"; return *this; }

        std::string global_id(int d) const {
            const char dim[] = {'x', 'y', 'z'};
            std::ostringstream s;
            s << "(threadIdx." << dim[d] << " + blockIdx." << dim[d] << " * blockDim." << dim[d] << ")";
            return s.str();
        }
        std::string global_size(int d) const {
            const char dim[] = {'x', 'y', 'z'};
            std::ostringstream s;
            s << "(blockDim." << dim[d] << " * gridDim." << dim[d] << ")";
            return s.str();
        }
        std::string local_id(int d) const { const char dim[] = {'x', 'y', 'z'}; return std::string("threadIdx.") + dim[d]; }
        std::string local_size(int d) const { const char dim[] = {'x', 'y', 'z'}; return std::string("blockDim.") + dim[d]; }
        std::string group_id(int d) const { const char dim[] = {'x', 'y', 'z'}; return std::string("blockIdx.") + dim[d]; }
        std::string num_groups(int d) const { const char dim[] = {'x', 'y', 'z'}; return std::string("gridDim.") + dim[d]; }

        std::string str() const { return src.str(); }

        template <class T>
        friend source_generator &operator<<(source_generator &s, const T &t) { s.src << t; return s; }

        source_generator &prm_separator() {
            if (first_prm) first_prm = false; else src << ",";
            return *this;
        }
    private:
        unsigned indent;
        bool first_prm;
        std::ostringstream src;
};

// ---- per-device compile options / program header stacks
//      (backend/common.hpp:61-205) ---------------------------------------------
namespace detail_opts {
    template <bool dummy = true> struct stacks {
        static std::mutex mx;
        static std::map<device_id, std::vector<std::string>> options, headers;
    };
    template <bool d> std::mutex stacks<d>::mx;
    template <bool d> std::map<device_id, std::vector<std::string>> stacks<d>::options;
    template <bool d> std::map<device_id, std::vector<std::string>> stacks<d>::headers;

    inline std::string top(std::map<device_id, std::vector<std::string>> &m, device_id d) {
        std::lock_guard<std::mutex> lock(stacks<>::mx);
        auto it = m.find(d);
        return (it == m.end() || it->second.empty()) ? std::string() : it->second.back();
    }
}

inline std::string get_compile_options(const command_queue &q) { return detail_opts::top(detail_opts::stacks<>::options, get_device_id(q)); }
inline std::string get_program_header(const command_queue &q) { return detail_opts::top(detail_opts::stacks<>::headers, get_device_id(q)); }
inline void push_compile_options(const command_queue &q, const std::string &s) {
    std::lock_guard<std::mutex> lock(detail_opts::stacks<>::mx); detail_opts::stacks<>::options[get_device_id(q)].push_back(s);
}
inline void pop_compile_options(const command_queue &q) {
    std::lock_guard<std::mutex> lock(detail_opts::stacks<>::mx);
    auto &v = detail_opts::stacks<>::options[get_device_id(q)]; if (!v.empty()) v.pop_back();
}
inline void push_program_header(const command_queue &q, const std::string &s) {
    std::lock_guard<std::mutex> lock(detail_opts::stacks<>::mx); detail_opts::stacks<>::headers[get_device_id(q)].push_back(s);
}
inline void pop_program_header(const command_queue &q) {
    std::lock_guard<std::mutex> lock(detail_opts::stacks<>::mx);
    auto &v = detail_opts::stacks<>::headers[get_device_id(q)]; if (!v.empty()) v.pop_back();
}

inline void push_compile_options(const std::vector<command_queue> &queue, const std::string &s) { for (const auto &q : queue) push_compile_options(q, s); }
inline void pop_compile_options(const std::vector<command_queue> &queue) { for (const auto &q : queue) pop_compile_options(q); }
inline void push_program_header(const std::vector<command_queue> &queue, const std::string &s) { for (const auto &q : queue) push_program_header(q, s); }
inline void pop_program_header(const std::vector<command_queue> &queue) { for (const auto &q : queue) pop_program_header(q); }

/// Options / header in force for the lifetime of the object (backend/common.hpp:146-206).
struct scoped_compile_options {
    std::vector<command_queue> q;
    scoped_compile_options(const std::vector<command_queue> &q, const std::string &s) : q(q) { push_compile_options(this->q, s); }
    scoped_compile_options(const command_queue &q, const std::string &s) : q(1, q) { push_compile_options(this->q, s); }
    ~scoped_compile_options() { pop_compile_options(q); }
};
struct scoped_program_header {
    std::vector<command_queue> q;
    scoped_program_header(const std::vector<command_queue> &q, const std::string &s) : q(q) { push_program_header(this->q, s); }
    scoped_program_header(const command_queue &q, const std::string &s) : q(1, q) { push_program_header(this->q, s); }
    ~scoped_program_header() { pop_program_header(q); }
};

inline std::string standard_kernel_header(const command_queue &q) {
    // Lengths 8 and 16 of the short vector types: HIP itself stops at 4 (types.hpp).
    static const char *wide =
        "#define VEX_WIDE(T) typedef T T##8 __attribute__((ext_vector_type(8))); typedef T T##16 __attribute__((ext_vector_type(16)));\n"
        "VEX_WIDE(char) VEX_WIDE(uchar) VEX_WIDE(short) VEX_WIDE(ushort) VEX_WIDE(int) VEX_WIDE(uint) VEX_WIDE(long) VEX_WIDE(ulong) VEX_WIDE(float) VEX_WIDE(double)\n"
        "#undef VEX_WIDE\n";
    return std::string("// vexcl kernel (gfx950)\n") + wide + get_program_header(q);
}

// ---- build_sources / program (backend/cuda/compiler.hpp:53-116) -------------
class program {
    public:
        program() {}
        program(const command_queue &q, const std::string &source, const std::string &options)
            : p_(std::make_shared<impl>(q, source, options)) {}
        void *raw() const { return p_ ? p_->module : nullptr; }
        int device_ordinal() const { return p_->dev; }
    private:
        struct impl {
            int dev; void *module = nullptr;
            impl(const command_queue &q, const std::string &src, const std::string &opt) : dev(q.device_ordinal()) {
                check(vexhip_module_compile(dev, src.c_str(), opt.c_str(), &module));
            }
            ~impl() { if (module) vexhip_module_unload(dev, module); }
        };
        std::shared_ptr<impl> p_;
};

inline program build_sources(const command_queue &q, const std::string &source, const std::string &options = "") {
    return program(q, source, options + " " + get_compile_options(q));
}

/// Compiles the source with hiprtc for the given architecture without loading
/// it (no GPU needed): lets CPU-only builds check generated kernels.
inline void check_sources(const std::string &source, const std::string &options = "", const std::string &arch = "gfx950") {
    check(vexhip_jit_check(source.c_str(), options.c_str(), arch.c_str()));
}

// ---- kernel (backend/cuda/kernel.hpp:45-244) --------------------------------
struct ndrange {
    size_t x, y, z;
    ndrange(size_t x = 1, size_t y = 1, size_t z = 1) : x(x), y(y), z(z) {}
};

class kernel {
    public:
        kernel() : fn_(nullptr), smem_(0) {}

        kernel(const command_queue &q, const std::string &src, const std::string &name,
               size_t smem_per_thread = 0, const std::string &options = "")
            : prog_(build_sources(q, src, options)), smem_(0)
        {
            init(q, name);
            config(q, [smem_per_thread](size_t wgs) { return wgs * smem_per_thread; });
        }
        kernel(const command_queue &q, const std::string &src, const std::string &name,
               std::function<size_t(size_t)> smem, const std::string &options = "")
            : prog_(build_sources(q, src, options)), smem_(0)
        {
            init(q, name);
            config(q, smem);
        }
        kernel(const command_queue &q, const program &p, const std::string &name, size_t smem_per_thread = 0)
            : prog_(p), smem_(0)
        {
            init(q, name);
            config(q, [smem_per_thread](size_t wgs) { return wgs * smem_per_thread; });
        }
        kernel(const command_queue &q, const program &p, const std::string &name, std::function<size_t(size_t)> smem)
            : prog_(p), smem_(0)
        {
            init(q, name);
            config(q, smem);
        }

        template <class Arg> void push_arg(const Arg &arg) {
            static_assert(std::is_trivially_copyable<Arg>::value, "kernel arguments are passed by value");
            size_t off = stack_.size();
            off = (off + alignof(Arg) - 1) / alignof(Arg) * alignof(Arg);
            stack_.resize(off + sizeof(Arg));
            std::memcpy(stack_.data() + off, &arg, sizeof(Arg));
            offsets_.push_back(off);
        }
        template <class T> void push_arg(const device_vector<T> &arg) { push_arg(arg.raw()); }
        void set_smem(size_t bytes) { smem_ = bytes; }
        template <class F> void set_smem(F &&f) { smem_ = f(block_.x); }

        void operator()(const command_queue &q) {
            std::vector<void *> ptrs(offsets_.size());
            for (size_t i = 0; i < offsets_.size(); ++i) ptrs[i] = stack_.data() + offsets_[i];
            check(vexhip_launch(q.device_ordinal(), fn_, (unsigned)grid_.x, (unsigned)grid_.y, (unsigned)grid_.z,
                        (unsigned)block_.x, (unsigned)block_.y, (unsigned)block_.z, (unsigned)smem_, q.raw(),
                        ptrs.empty() ? nullptr : ptrs.data()));
            reset();
        }
        template <class Head, class... Tail>
        void operator()(const command_queue &q, const Head &head, const Tail &...tail) {
            push_arg(head);
            (*this)(q, tail...);
        }

        size_t workgroup_size() const { return block_.x * block_.y * block_.z; }
        static size_t num_workgroups(const command_queue &q) { return 8 * q.device().multiprocessor_count(); }
        size_t max_threads_per_block(const command_queue &) const { return max_threads_; }
        size_t max_shared_memory_per_block(const command_queue &q) const {
            return q.device().max_shared_memory_per_block() - static_lds_;
        }
        size_t preferred_work_group_size_multiple(const command_queue &q) const { return q.device().wavefront_size(); }

        kernel &config(const command_queue &q, std::function<size_t(size_t)> smem) {
            // block = (max threads for this kernel)/2, halved until the LDS request fits
            // (backend/cuda/kernel.hpp:182-195), capped at 256 lanes = 4 waves
            size_t ws = std::min<size_t>(256, max_threads_ / 2 ? max_threads_ / 2 : max_threads_);
            size_t max_smem = max_shared_memory_per_block(q);
            while (ws > 64 && smem(ws) > max_smem) ws /= 2;
            return config(num_workgroups(q), ws, smem(ws));
        }
        /// Launch geometry of a streaming (elementwise) kernel over n elements: about `per_lane`
        /// elements per lane instead of the reference's fixed 8 workgroups per CU.  Measured on
        /// MI355X (tools/bw_probe.py, a = b*c + d at 1e8 fp64): 2 048 workgroups 5.2 TB/s, 131 072
        /// workgroups 6.0 TB/s, one trip per lane 6.07 TB/s -- workgroups dispatched in order walk
        /// memory front to back, a resident grid strides across it.
        kernel &config_streaming(const command_queue &q, size_t n, size_t per_lane = 2) {
            const size_t ws = block_.x ? block_.x : 256;
            size_t blocks = (n + ws * per_lane - 1) / (ws * per_lane);
            blocks = std::max(blocks, std::min(num_workgroups(q), (n + ws - 1) / ws));
            blocks = std::max<size_t>(1, std::min<size_t>(blocks, (size_t(1) << 31) - 1));
            grid_ = ndrange(blocks);
            return *this;
        }
        kernel &config(ndrange blocks, ndrange threads, size_t shared_memory) {
            grid_ = blocks; block_ = threads; smem_ = shared_memory; return *this;
        }
        kernel &config(ndrange blocks, ndrange threads) { grid_ = blocks; block_ = threads; return *this; }
        kernel &config(size_t blocks, size_t threads) { return config(ndrange(blocks), ndrange(threads)); }
        kernel &config(size_t blocks, size_t threads, size_t shared_memory) {
            return config(ndrange(blocks), ndrange(threads), shared_memory);
        }

        void reset() { stack_.clear(); offsets_.clear(); }
        void *get() const { return fn_; }
    private:
        void init(const command_queue &q, const std::string &name) {
            check(vexhip_module_get_function(q.device_ordinal(), prog_.raw(), name.c_str(), &fn_));
            int mt = 0, sl = 0;
            check(vexhip_function_max_threads(q.device_ordinal(), fn_, &mt, &sl));
            max_threads_ = mt > 0 ? mt : 256; static_lds_ = sl;
        }
        program prog_;
        void *fn_;
        ndrange grid_, block_;
        size_t smem_, max_threads_ = 256, static_lds_ = 0;
        std::vector<char> stack_;
        std::vector<size_t> offsets_;
};

// ---- exclusive device locks (Filter::Exclusive) ---------------------------------
/// Takes (once per process) the advisory lock file of a device; false if another process holds it.
inline bool lock_device(int ordinal) {
    static std::mutex mx;
    static std::map<int, int> held;                      // ordinal -> open file descriptor, kept until exit
    std::lock_guard<std::mutex> l(mx);
    if (held.count(ordinal)) return true;
    const char *dir = std::getenv("VEXCL_LOCK_DIR");
    const std::string path = std::string(dir ? dir : "/tmp") + "/vexcl_device_" + std::to_string(ordinal) + ".lock";
    int fd = ::open(path.c_str(), O_CREAT | O_RDWR, 0666);
    if (fd < 0) return true;                             // cannot create lock files: exclusive mode is off (as the reference warns)
    if (::flock(fd, LOCK_EX | LOCK_NB) != 0) { ::close(fd); return false; }
    held[ordinal] = fd;
    return true;
}

// ---- device enumeration (backend/cuda/context.hpp:383-413) ------------------
inline int device_count() {
    int n = 0;
    check(vexhip_device_count(&n));
    return n;
}

template <class DevFilter>
std::vector<device> device_list(DevFilter &&filter) {
    std::vector<device> out;
    int n = device_count();
    for (int d = 0; d < n; ++d) {
        device dev(d);
        if (filter(dev)) out.push_back(dev);
    }
    return out;
}

template <class DevFilter>
std::pair<std::vector<context>, std::vector<command_queue>>
queue_list(DevFilter &&filter, command_queue_properties flags = 0) {
    std::vector<context> c;
    std::vector<command_queue> q;
    int n = device_count();
    for (int d = 0; d < n; ++d) {
        device dev(d);
        if (!filter(dev)) continue;
        try {
            context ctx(dev);
            command_queue queue(ctx, dev, flags);
            c.push_back(ctx);
            q.push_back(queue);
        } catch (const error &) { }   // skipped silently, as backend/opencl/context.hpp:179-185
    }
    return std::make_pair(c, q);
}

} // namespace backend

using backend::command_queue;
using backend::device_vector;
using backend::push_compile_options; using backend::pop_compile_options;
using backend::push_program_header; using backend::pop_program_header;
using backend::scoped_compile_options; using backend::scoped_program_header;
typedef backend::error error;
using backend::operator<<;

} // namespace vex
#endif
