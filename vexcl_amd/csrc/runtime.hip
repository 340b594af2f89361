// libvexhip.so runtime layer: devices, streams, events, memory, hiprtc JIT and
// generic kernel launch.  Stands in for the reference's backend concept
// (backend/cuda/context.hpp, device_vector.hpp, kernel.hpp, compiler.hpp,
// event.hpp, error.hpp; kernel-binary cache backend/common.hpp:215-285).
#include "common.hpp"

#include <hip/hiprtc.h>

#include <atomic>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>
#include <unordered_map>
#include <vector>
#include <sys/stat.h>
#include <unistd.h>

namespace vexhip {

std::string &last_error() {
    static thread_local std::string s;
    return s;
}

int &last_error_code() {
    static thread_local int c = 0;
    return c;
}

int fail(const char *file, int line, const std::string &what) {
    std::ostringstream o;
    o << file << ":" << line << "\n\t" << what;
    last_error() = o.str();
    last_error_code() = 0;                     // a failure of the library's own checks; check() below overwrites it for a HIP error
    return 1;
}

void warm_misc();
void warm_reduce();
void warm_scan();
void warm_sort();
void warm_stencil();
void warm_ccsr();
void warm_sell8();
void warm_plane();
void warm_plane32();
void warm_grid();
void warm_grid32();
void warm_spmm();
void warm_spmv();
void warm_split();
void warm_comm();
void warm_fft();
void warm_mba();

const device_info &info(int dev) {
    static std::mutex mx;
    static device_info cache[64];              // fixed storage: references stay valid across threads
    static device_info fallback;
    std::lock_guard<std::mutex> lock(mx);
    if (dev < 0 || dev >= 64) { fallback.cus = 256; return fallback; }
    if (!cache[dev].ok) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) {
            cache[dev].cus = p.multiProcessorCount;
            cache[dev].ok = true;
            // HIP sets up its staging path for pageable host memory at the first copy that needs it: 6.7 ms for a 32 KiB
            // read-back, measured inside the first matrix set-up of a process (VEXHIP_SETUP_TRACE).  Pay it here, when the
            // library first meets the device (context creation), not in the first set-up.
            // (if the calling thread is capturing a stream into a hipGraph -- a step object meeting a device for the first time
            //  inside a capture -- the synchronous calls below must not invalidate that capture: relaxed mode for their duration)
            hipStreamCaptureMode cmode = hipStreamCaptureModeRelaxed;
            const bool swapped = hipThreadExchangeStreamCaptureMode(&cmode) == hipSuccess;
            int cur = -1;
            if (hipGetDevice(&cur) == hipSuccess && hipSetDevice(dev) == hipSuccess) {
                void *d = nullptr;
                if (hipMalloc(&d, 1 << 16) == hipSuccess) {
                    std::vector<char> h(1 << 16);
                    (void)hipMemcpy(h.data(), d, h.size(), hipMemcpyDeviceToHost);
                    (void)hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
                    (void)hipFree(d);
                }
                // ... and the code objects of the library's files (see VEXHIP_WARM_TU)
                warm_misc(); warm_reduce(); warm_scan(); warm_sort(); warm_stencil(); warm_ccsr(); warm_sell8(); warm_plane(); warm_plane32(); warm_grid(); warm_grid32(); warm_spmm(); warm_spmv(); warm_split(); warm_comm(); warm_fft(); warm_mba();
                (void)hipDeviceSynchronize();
                (void)hipGetLastError();
                (void)hipSetDevice(cur);
            }
            if (swapped) (void)hipThreadExchangeStreamCaptureMode(&cmode);
        } else {
            cache[dev].cus = 256;
        }
    }
    return cache[dev];
}

} // namespace vexhip

using namespace vexhip;

// ---------------------------------------------------------------- JIT
namespace {

std::atomic<uint64_t> g_compiled{0}, g_disk_hits{0};

// 128-bit FNV-1a style digest, hex-printed: names the cache entry the way the
// reference names it by SHA-1 (backend/common.hpp:235-285).
std::string digest(const std::string &s) {
    uint64_t h1 = 0xcbf29ce484222325ull, h2 = 0x84222325cbf29ce4ull;
    for (unsigned char c : s) {
        h1 = (h1 ^ c) * 0x100000001b3ull;
        h2 = (h2 ^ (c + 0x9e)) * 0x100000001b3ull;
        h2 ^= h2 >> 29;
    }
    char buf[40];
    std::snprintf(buf, sizeof(buf), "%016llx%016llx", (unsigned long long)h1, (unsigned long long)h2);
    return buf;
}

std::string cache_root() {
    if (const char *e = env(ENV_VEXCL_CACHE_DIR)) return e;
    const char *home = std::getenv("HOME");
    return std::string(home ? home : "/tmp") + "/.vexcl_amd";
}

bool cache_enabled() {
    const char *e = env(ENV_VEXCL_CACHE_KERNELS);
    return !(e && e[0] == '0');
}

bool mkdirs(const std::string &path) {
    std::string cur;
    for (size_t i = 0; i < path.size(); ++i) {
        cur += path[i];
        if (path[i] == '/' || i + 1 == path.size())
            if (::mkdir(cur.c_str(), 0755) != 0 && errno != EEXIST) return false;
    }
    return true;
}

bool read_file(const std::string &p, std::vector<char> &out) {
    std::ifstream f(p, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    std::streamsize n = f.tellg();
    if (n <= 0) return false;
    f.seekg(0);
    out.resize((size_t)n);
    return (bool)f.read(out.data(), n);
}

void write_file_atomic(const std::string &dir, const std::string &name, const std::vector<char> &data) {
    if (!mkdirs(dir)) return;
    std::string tmp = dir + "/." + name + "." + std::to_string((long)::getpid());
    {
        std::ofstream f(tmp, std::ios::binary);
        if (!f) return;
        f.write(data.data(), (std::streamsize)data.size());
        if (!f) { ::unlink(tmp.c_str()); return; }
    }
    if (::rename(tmp.c_str(), (dir + "/" + name).c_str()) != 0) ::unlink(tmp.c_str());
}

int rtc_fail(const char *file, int line, hiprtcResult r, const std::string &log) {
    return fail(file, line, std::string("hiprtc: ") + hiprtcGetErrorString(r) + (log.empty() ? "" : "\n" + log));
}

} // namespace

// ---------------------------------------------------------------- the environment (common.hpp)
namespace vexhip {
namespace {
const char *const kEnvNames[ENV_COUNT] = {
    "VEXCL_CACHE_DIR",
    "VEXCL_CACHE_KERNELS",
    "VEXCL_SHOW_KERNELS",
    "VEXCL_SHOW_SCRATCH",
    "VEXHIP_COMM_PEER",
    "VEXHIP_DEBUG",
    "VEXHIP_DIST_PACK",
    "VEXHIP_FFT_EXTRA_LDS",
    "VEXHIP_FFT_LANES_DIV",
    "VEXHIP_FFT_NO_SINGLE",
    "VEXHIP_FFT_ROW_ELEMS",
    "VEXHIP_FFT_STRIDED_ELEMS",
    "VEXHIP_GRID32_DEPTH",
    "VEXHIP_GRID_2D_LINE",
    "VEXHIP_GRID_BUILD_WGS",
    "VEXHIP_GRID_SEGMENT",
    "VEXHIP_HALO_ACQUIRE",
    "VEXHIP_HALO_DEBUG",
    "VEXHIP_HALO_DEPTH",
    "VEXHIP_HALO_EDGE_PLANES",
    "VEXHIP_HALO_HI_PLANES",
    "VEXHIP_HALO_LO_PLANES",
    "VEXHIP_HALO_NO_GHOST",
    "VEXHIP_HALO_NO_PUSH",
    "VEXHIP_HALO_PUSH_BLOCKS",
    "VEXHIP_HALO_TWO_LAUNCHES",
    "VEXHIP_HALO_TWO_PASS",
    "VEXHIP_IPC_PUSH_PER_BLOCK",
    "VEXHIP_IPC_TIMEOUT_MS",
    "VEXHIP_IPC_WINDOW_MEM",
    "VEXHIP_MALLOC_STAGGER",
    "VEXHIP_MARCH_LDS",
    "VEXHIP_MARCH_RUN",
    "VEXHIP_NO_GRID",
    "VEXHIP_NO_GRID_BUILD",
    "VEXHIP_NO_PLANE512",
    "VEXHIP_PLANE32_DEPTH",
    "VEXHIP_PLANE_DEPTH",
    "VEXHIP_PLANE_FORCE",
    "VEXHIP_PLANE_STORE",
    "VEXHIP_PLANE_TILE",
    "VEXHIP_RCCL_SELF",
    "VEXHIP_SELL_XLOAD",
    "VEXHIP_SETUP_TRACE",
};
struct env_snapshot { bool set[ENV_COUNT]; char text[ENV_COUNT][96]; };
env_snapshot g_env;
std::mutex g_env_mx;
std::once_flag g_env_once;
void read_env() {
    std::lock_guard<std::mutex> lock(g_env_mx);
    for (int i = 0; i < ENV_COUNT; ++i) {
        const char *v = std::getenv(kEnvNames[i]);          // THE place where the library reads its switches
        const bool set = v != nullptr;
        // entries that did not change are not touched: a reader on another thread never sees a half-written one of those
        if (set == g_env.set[i] && (!set || std::strncmp(g_env.text[i], v, sizeof(g_env.text[i]) - 1) == 0)) continue;
        if (set) { std::strncpy(g_env.text[i], v, sizeof(g_env.text[i]) - 1); g_env.text[i][sizeof(g_env.text[i]) - 1] = 0; }
        g_env.set[i] = set;
    }
}
}
const char *env(env_id id) {
    std::call_once(g_env_once, read_env);
    return g_env.set[id] ? g_env.text[id] : nullptr;
}
void reload_env() {
    std::call_once(g_env_once, read_env);
    read_env();
}
}


extern "C" {

const char *vexhip_last_error(void) { return last_error().c_str(); }
int vexhip_last_error_code(void) { return last_error_code(); }
int vexhip_abi_version(void) { return VEXHIP_ABI_VERSION; }

// ---------------------------------------------------------------- devices
int vexhip_device_count(int *count) {
    VEXHIP_REQUIRE(count, "count is NULL");
    hipError_t e = hipGetDeviceCount(count);
    if (e == hipErrorNoDevice) { *count = 0; (void)hipGetLastError(); return 0; }
    VEXHIP_TRY(e);
    return 0;
}

int vexhip_device_get_props(int dev, vexhip_device_props *out) {
    VEXHIP_REQUIRE(out, "props is NULL");
    hipDeviceProp_t p;
    VEXHIP_TRY(hipGetDeviceProperties(&p, dev));
    std::memset(out, 0, sizeof(*out));
    std::strncpy(out->name, p.name, sizeof(out->name) - 1);
    std::strncpy(out->arch, p.gcnArchName, sizeof(out->arch) - 1);
    out->compute_units = p.multiProcessorCount;
    out->wavefront_size = p.warpSize;
    out->max_threads_per_block = p.maxThreadsPerBlock;
    out->lds_bytes_per_block = (int32_t)p.sharedMemPerBlock;
    out->clock_khz = p.clockRate;
    out->l2_bytes = p.l2CacheSize;
    out->global_mem_bytes = p.totalGlobalMem;
    out->pci_bus_id = p.pciBusID;
    return 0;
}

int vexhip_device_sync(int dev) {
    VEXHIP_SET_DEVICE(dev);
    VEXHIP_TRY(hipDeviceSynchronize());
    return 0;
}

int vexhip_mem_info(int dev, uint64_t *free_bytes, uint64_t *total_bytes) {
    VEXHIP_SET_DEVICE(dev);
    size_t f = 0, t = 0;
    VEXHIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}

// ---------------------------------------------------------------- streams
int vexhip_stream_create(int dev, void **stream) {
    VEXHIP_REQUIRE(stream, "stream is NULL");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s;
    VEXHIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return 0;
}

int vexhip_stream_destroy(int dev, void *stream) {
    VEXHIP_SET_DEVICE(dev);
    VEXHIP_TRY(hipStreamDestroy(as_stream(stream)));
    return 0;
}

int vexhip_stream_sync(int dev, void *stream) {
    VEXHIP_SET_DEVICE(dev);
    VEXHIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return 0;
}

// ---------------------------------------------------------------- events
int vexhip_event_create(int dev, int timing, void **event) {
    VEXHIP_REQUIRE(event, "event is NULL");
    VEXHIP_SET_DEVICE(dev);
    hipEvent_t e;
    VEXHIP_TRY(hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming));
    *event = e;
    return 0;
}

int vexhip_event_destroy(int dev, void *event) {
    VEXHIP_SET_DEVICE(dev);
    VEXHIP_TRY(hipEventDestroy((hipEvent_t)event));
    return 0;
}

int vexhip_event_record(int dev, void *event, void *stream) {
    VEXHIP_SET_DEVICE(dev);
    VEXHIP_TRY(hipEventRecord((hipEvent_t)event, as_stream(stream)));
    return 0;
}

int vexhip_event_sync(int dev, void *event) {
    VEXHIP_SET_DEVICE(dev);
    VEXHIP_TRY(hipEventSynchronize((hipEvent_t)event));
    return 0;
}

int vexhip_stream_wait_event(int dev, void *stream, void *event) {
    VEXHIP_SET_DEVICE(dev);
    VEXHIP_TRY(hipStreamWaitEvent(as_stream(stream), (hipEvent_t)event, 0));
    return 0;
}

int vexhip_event_elapsed_ms(int dev, void *start, void *stop, float *ms) {
    VEXHIP_SET_DEVICE(dev);
    VEXHIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}


// ---------------------------------------------------------------- memory
// Where a large allocation starts (round 6).  The headline product reads x a few planes ahead of where it writes y; its time
// depends on (y - x) mod 64 MiB and on nothing else of the placement (profiles/r06_xy_gap.json: two sweeps of 257 gaps correlate
// at 0.994): 0.375 - 0.382 ms where the difference lies within +-10 MiB of a multiple of 64 MiB, up to 0.417 ms between 12 and
// 32 MiB (bases at 2 MiB multiples -- what any allocator hands out; a third of all placements).  The library owns the allocation of
// every vex::vector, so it places them: allocations of 64 MiB and more start at a multiple of 64 MiB plus a stagger of 0, 2, 4, 6,
// 8 MiB in turn (any two of them then differ by less than 10 MiB mod 64 MiB, and vectors that are walked in lockstep -- a = b * c +
// sin(d) -- do not all start on the same channel).  Costs up to 72 MiB of address space per large allocation.
static constexpr size_t kStaggerAlign = size_t(64) << 20, kStaggerStep = size_t(2) << 20, kStaggerSlots = 5, kStaggerFrom = size_t(64) << 20;

extern "C" size_t vexhip_malloc_stagger(size_t bytes, unsigned ordinal) {
    if (bytes < kStaggerFrom) return 0;
    static const bool off = env(ENV_VEXHIP_MALLOC_STAGGER) && std::atoi(env(ENV_VEXHIP_MALLOC_STAGGER)) == 0;
    if (off) return 0;
    return (ordinal % kStaggerSlots) * kStaggerStep;
}
extern "C" size_t vexhip_malloc_placement(size_t bytes, uint64_t raw_address, unsigned ordinal) {
    if (bytes < kStaggerFrom) return 0;
    const uint64_t aligned = (raw_address + kStaggerAlign - 1) / kStaggerAlign * kStaggerAlign;
    return (size_t)(aligned - raw_address) + vexhip_malloc_stagger(bytes, ordinal);
}

namespace {
std::mutex g_alloc_mx;
std::unordered_map<void *, void *> g_alloc_base;       // what the caller holds -> what hipFree takes
std::atomic<unsigned> g_alloc_ordinal{0};
}

int vexhip_reload_env(void) { reload_env(); return 0; }

int vexhip_malloc(int dev, size_t bytes, void **ptr) {
    VEXHIP_REQUIRE(ptr, "ptr is NULL");
    VEXHIP_SET_DEVICE(dev);
    *ptr = nullptr;
    if (bytes == 0) return 0;
    static const bool off = env(ENV_VEXHIP_MALLOC_STAGGER) && std::atoi(env(ENV_VEXHIP_MALLOC_STAGGER)) == 0;
    if (bytes < kStaggerFrom || off) { VEXHIP_TRY(hipMalloc(ptr, bytes)); return 0; }
    void *raw = nullptr;
    VEXHIP_TRY(hipMalloc(&raw, bytes + kStaggerAlign + (kStaggerSlots - 1) * kStaggerStep));
    const size_t shift = vexhip_malloc_placement(bytes, (uint64_t)reinterpret_cast<uintptr_t>(raw), g_alloc_ordinal.fetch_add(1));
    void *p = static_cast<char *>(raw) + shift;
    if (p != raw) { std::lock_guard<std::mutex> lock(g_alloc_mx); g_alloc_base[p] = raw; }
    *ptr = p;
    return 0;
}

int vexhip_malloc_managed(int dev, size_t bytes, void **ptr) {
    VEXHIP_REQUIRE(ptr, "ptr is NULL");
    VEXHIP_SET_DEVICE(dev);
    *ptr = nullptr;
    if (bytes == 0) return 0;
    VEXHIP_TRY(hipMallocManaged(ptr, bytes, hipMemAttachGlobal));
    return 0;
}

int vexhip_free(int dev, void *ptr) {
    if (!ptr) return 0;
    VEXHIP_SET_DEVICE(dev);
    {
        std::lock_guard<std::mutex> lock(g_alloc_mx);
        auto it = g_alloc_base.find(ptr);
        if (it != g_alloc_base.end()) { ptr = it->second; g_alloc_base.erase(it); }
    }
    VEXHIP_TRY(hipFree(ptr));
    return 0;
}

int vexhip_memcpy_h2d(int dev, void *dst, const void *src, size_t bytes, void *stream, int blocking) {
    if (!bytes) return 0;
    VEXHIP_SET_DEVICE(dev);
    VEXHIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    if (blocking) VEXHIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return 0;
}

int vexhip_memcpy_d2h(int dev, void *dst, const void *src, size_t bytes, void *stream, int blocking) {
    if (!bytes) return 0;
    VEXHIP_SET_DEVICE(dev);
    VEXHIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    if (blocking) VEXHIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return 0;
}

int vexhip_memcpy_d2d(int dev, void *dst, const void *src, size_t bytes, void *stream) {
    if (!bytes) return 0;
    VEXHIP_SET_DEVICE(dev);
    VEXHIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return 0;
}

int vexhip_memcpy_peer(int dst_dev, void *dst, int src_dev, const void *src, size_t bytes, void *stream) {
    if (!bytes) return 0;
    VEXHIP_SET_DEVICE(dst_dev);
    if (dst_dev == src_dev)
        VEXHIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    else
        VEXHIP_TRY(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, as_stream(stream)));
    return 0;
}

int vexhip_memset(int dev, void *ptr, int byte, size_t bytes, void *stream) {
    if (!bytes) return 0;
    VEXHIP_SET_DEVICE(dev);
    VEXHIP_TRY(hipMemsetAsync(ptr, byte, bytes, as_stream(stream)));
    return 0;
}

int vexhip_host_alloc(size_t bytes, void **ptr) {
    VEXHIP_REQUIRE(ptr, "ptr is NULL");
    VEXHIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return 0;
}

int vexhip_host_free(void *ptr) {
    if (!ptr) return 0;
    VEXHIP_TRY(hipHostFree(ptr));
    return 0;
}

// ---------------------------------------------------------------- JIT

int vexhip_module_compile(int dev, const char *source, const char *options, void **module) {
    reload_env();                              // a module is created: VEXCL_CACHE_DIR / _CACHE_KERNELS / _SHOW_KERNELS are read now
    VEXHIP_REQUIRE(source && module, "source/module is NULL");
    VEXHIP_SET_DEVICE(dev);
    hipDeviceProp_t prop;
    VEXHIP_TRY(hipGetDeviceProperties(&prop, dev));
    std::string arch = prop.gcnArchName;
    std::string opts = options ? options : "";

    if (const char *show = env(ENV_VEXCL_SHOW_KERNELS))
        if (show[0] != '0') std::printf("%s\n", source);

    int rtc_major = 0, rtc_minor = 0;
    hiprtcVersion(&rtc_major, &rtc_minor);
    std::string key = digest(std::string(source) + "\x01" + opts + "\x01" + arch + "\x01" +
                             std::to_string(rtc_major) + "." + std::to_string(rtc_minor));
    std::string dir = cache_root() + "/" + key.substr(0, 2) + "/" + key.substr(2);

    std::vector<char> code;
    bool from_disk = cache_enabled() && read_file(dir + "/kernel.hsaco", code);

    if (!from_disk) {
        hiprtcProgram prog;
        hiprtcResult r = hiprtcCreateProgram(&prog, source, "vexcl_kernel.hip", 0, nullptr, nullptr);
        if (r != HIPRTC_SUCCESS) return rtc_fail(__FILE__, __LINE__, r, "");

        std::vector<std::string> ostr;
        ostr.push_back("--offload-arch=" + arch);
        ostr.push_back("-O3");
        ostr.push_back("-std=c++17");
        {
            std::istringstream is(opts);
            std::string tok;
            while (is >> tok) ostr.push_back(tok);
        }
        std::vector<const char *> oc;
        for (auto &s : ostr) oc.push_back(s.c_str());

        r = hiprtcCompileProgram(prog, (int)oc.size(), oc.data());
        if (r != HIPRTC_SUCCESS) {
            size_t ls = 0;
            hiprtcGetProgramLogSize(prog, &ls);
            std::string log(ls, '\0');
            if (ls) hiprtcGetProgramLog(prog, &log[0]);
            hiprtcDestroyProgram(&prog);
            // the reference prints source + build log, then rethrows
            // (backend/opencl/compiler.hpp:164-174)
            std::fprintf(stderr, "%s\n%s\n", source, log.c_str());
            return rtc_fail(__FILE__, __LINE__, r, log);
        }
        size_t cs = 0;
        r = hiprtcGetCodeSize(prog, &cs);
        if (r != HIPRTC_SUCCESS) { hiprtcDestroyProgram(&prog); return rtc_fail(__FILE__, __LINE__, r, ""); }
        code.resize(cs);
        r = hiprtcGetCode(prog, code.data());
        hiprtcDestroyProgram(&prog);
        if (r != HIPRTC_SUCCESS) return rtc_fail(__FILE__, __LINE__, r, "");
        ++g_compiled;
        if (cache_enabled()) write_file_atomic(dir, "kernel.hsaco", code);
    } else {
        ++g_disk_hits;
    }

    hipModule_t m;
    VEXHIP_TRY(hipModuleLoadData(&m, code.data()));
    *module = m;
    return 0;
}

int vexhip_jit_check(const char *source, const char *options, const char *arch) {
    VEXHIP_REQUIRE(source && arch, "NULL argument");
    hiprtcProgram prog;
    hiprtcResult r = hiprtcCreateProgram(&prog, source, "vexcl_kernel.hip", 0, nullptr, nullptr);
    if (r != HIPRTC_SUCCESS) return rtc_fail(__FILE__, __LINE__, r, "");
    std::vector<std::string> ostr;
    ostr.push_back(std::string("--offload-arch=") + arch);
    ostr.push_back("-O3");
    ostr.push_back("-std=c++17");
    {
        std::istringstream is(options ? options : "");
        std::string tok;
        while (is >> tok) ostr.push_back(tok);
    }
    std::vector<const char *> oc;
    for (auto &s : ostr) oc.push_back(s.c_str());
    r = hiprtcCompileProgram(prog, (int)oc.size(), oc.data());
    std::string log;
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        log.resize(ls);
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
    }
    hiprtcDestroyProgram(&prog);
    if (r != HIPRTC_SUCCESS) return rtc_fail(__FILE__, __LINE__, r, log);
    return 0;
}

int vexhip_module_unload(int dev, void *module) {
    if (!module) return 0;
    VEXHIP_SET_DEVICE(dev);
    VEXHIP_TRY(hipModuleUnload((hipModule_t)module));
    return 0;
}

int vexhip_module_get_function(int dev, void *module, const char *name, void **function) {
    VEXHIP_REQUIRE(module && name && function, "NULL argument");
    VEXHIP_SET_DEVICE(dev);
    hipFunction_t f;
    VEXHIP_TRY(hipModuleGetFunction(&f, (hipModule_t)module, name));
    *function = f;
    // VEXCL_SHOW_SCRATCH: report generated kernels whose private arrays ended up in scratch memory (a performance trap
    // of generated code: an array indexed by a loop the compiler did not unroll)
    if (env(ENV_VEXCL_SHOW_SCRATCH)) {
        int local = 0, regs = 0;
        if (hipFuncGetAttribute(&local, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, f) == hipSuccess && local > 0) {
            (void)hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, f);
            std::fprintf(stderr, "[vexhip] kernel %s uses %d bytes of scratch per lane (%d registers)\n", name, local, regs);
        }
    }
    return 0;
}

int vexhip_function_max_threads(int dev, void *function, int *max_threads, int *static_lds) {
    VEXHIP_SET_DEVICE(dev);
    int v = 0;
    if (max_threads) {
        VEXHIP_TRY(hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_MAX_THREADS_PER_BLOCK, (hipFunction_t)function));
        *max_threads = v;
    }
    if (static_lds) {
        VEXHIP_TRY(hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, (hipFunction_t)function));
        *static_lds = v;
    }
    return 0;
}

int vexhip_launch(int dev, void *function,
        unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz,
        unsigned lds, void *stream, void **args) {
    VEXHIP_REQUIRE(function, "function is NULL");
    VEXHIP_SET_DEVICE(dev);
    if (gx == 0 || gy == 0 || gz == 0) return 0;
    VEXHIP_TRY(hipModuleLaunchKernel((hipFunction_t)function, gx, gy, gz, bx, by, bz, lds,
                                     as_stream(stream), args, nullptr));
    return 0;
}

int vexhip_jit_stats(uint64_t *compiled, uint64_t *disk_hits) {
    if (compiled) *compiled = g_compiled.load();
    if (disk_hits) *disk_hits = g_disk_hits.load();
    return 0;
}

} // extern "C"
