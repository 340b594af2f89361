// Shared host-side helpers for libvexhip.so (error capture, device guard).
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/vexhip.h"

namespace vexhip {

// Thread-local last-error text, in the reference's "file:line\n\t<code text>"
// shape (backend/cuda/error.hpp:119-145).
std::string &last_error();
int &last_error_code();          // the hipError_t behind the last failure on this thread (0: none of HIP's -- a check of the library's own)
int fail(const char *file, int line, const std::string &what);

inline int check(hipError_t e, const char *file, int line) {
    if (e == hipSuccess) return 0;
    (void)hipGetLastError();      // HIP's last-error slot is sticky: clear it so a later launch check does not report this failure again
    const int rc = fail(file, line, std::string(hipGetErrorName(e)) + ": " + hipGetErrorString(e));
    last_error_code() = (int)e;
    return rc;
}

constexpr int kWave = 64;   // gfx950 wavefront

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

struct device_info {
    int cus = 0;
    bool ok = false;
};
const device_info &info(int dev);

// VEXHIP_SETUP_TRACE=1: host wall time of every stage of a set-up on stderr (each mark synchronises the stream: the trace is
// for finding where a set-up spends its time, the figures of a traced run are not those of an untraced one)

// ---- the environment, read in ONE place (round 6: 60 getenv sites until then) -------------------------------------------------
// Every switch of the library -- A/B experiments, diagnostics, test hooks; none is needed in normal use -- is looked up in a snapshot
// of the environment that runtime.hip takes at the first use and again whenever an OBJECT is created (a matrix, a plan, a window,
// a step, a communicator: reload_env() at the entry of those calls; a test may set a variable between two of them).  Products read
// the snapshot, never the environment.  env(id) = the variable's text, NULL if it is not set; the names are the enumerators' without
// their ENV_ prefix.
enum env_id {
    ENV_VEXCL_CACHE_DIR,
    ENV_VEXCL_CACHE_KERNELS,
    ENV_VEXCL_SHOW_KERNELS,
    ENV_VEXCL_SHOW_SCRATCH,
    ENV_VEXHIP_COMM_PEER,
    ENV_VEXHIP_DEBUG,
    ENV_VEXHIP_DIST_PACK,
    ENV_VEXHIP_FFT_EXTRA_LDS,
    ENV_VEXHIP_FFT_LANES_DIV,
    ENV_VEXHIP_FFT_NO_SINGLE,
    ENV_VEXHIP_FFT_ROW_ELEMS,
    ENV_VEXHIP_FFT_STRIDED_ELEMS,
    ENV_VEXHIP_GRID32_DEPTH,
    ENV_VEXHIP_GRID_2D_LINE,
    ENV_VEXHIP_GRID_BUILD_WGS,
    ENV_VEXHIP_GRID_SEGMENT,
    ENV_VEXHIP_HALO_ACQUIRE,
    ENV_VEXHIP_HALO_DEBUG,
    ENV_VEXHIP_HALO_DEPTH,
    ENV_VEXHIP_HALO_EDGE_PLANES,
    ENV_VEXHIP_HALO_HI_PLANES,
    ENV_VEXHIP_HALO_LO_PLANES,
    ENV_VEXHIP_HALO_NO_GHOST,
    ENV_VEXHIP_HALO_NO_PUSH,
    ENV_VEXHIP_HALO_PUSH_BLOCKS,
    ENV_VEXHIP_HALO_TWO_LAUNCHES,
    ENV_VEXHIP_HALO_TWO_PASS,
    ENV_VEXHIP_IPC_PUSH_PER_BLOCK,
    ENV_VEXHIP_IPC_TIMEOUT_MS,
    ENV_VEXHIP_IPC_WINDOW_MEM,
    ENV_VEXHIP_MALLOC_STAGGER,
    ENV_VEXHIP_MARCH_LDS,
    ENV_VEXHIP_MARCH_RUN,
    ENV_VEXHIP_NO_GRID,
    ENV_VEXHIP_NO_GRID_BUILD,
    ENV_VEXHIP_NO_PLANE512,
    ENV_VEXHIP_PLANE32_DEPTH,
    ENV_VEXHIP_PLANE_DEPTH,
    ENV_VEXHIP_PLANE_FORCE,
    ENV_VEXHIP_PLANE_STORE,
    ENV_VEXHIP_PLANE_TILE,
    ENV_VEXHIP_RCCL_SELF,
    ENV_VEXHIP_SELL_XLOAD,
    ENV_VEXHIP_SETUP_TRACE,
    ENV_COUNT
};
const char *env(env_id id);
void reload_env();

struct setup_trace {
    bool on; hipStream_t s; std::chrono::steady_clock::time_point t0, last;
    explicit setup_trace(hipStream_t st) : on(env(ENV_VEXHIP_SETUP_TRACE) != nullptr), s(st) { t0 = last = std::chrono::steady_clock::now(); }
    void mark(const char *what) {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[vexhip set-up] %-28s %8.3f ms  (at %8.3f)\n", what,
                     std::chrono::duration<double, std::milli>(now - last).count(), std::chrono::duration<double, std::milli>(now - t0).count());
        last = now;
    }
};

} // namespace vexhip

#define VEXHIP_TRY(expr)                                                      \
    do {                                                                      \
        if (int _rc = ::vexhip::check((expr), __FILE__, __LINE__)) return _rc;\
    } while (0)

#define VEXHIP_REQUIRE(cond, msg)                                             \
    do {                                                                      \
        if (!(cond)) return ::vexhip::fail(__FILE__, __LINE__, msg);          \
    } while (0)

#define VEXHIP_SET_DEVICE(dev) VEXHIP_TRY(hipSetDevice(dev))

// Every .hip file of the library is its own code object, and HIP loads a code object when the first kernel of it is launched on
// a device: 4.8 ms for sell8.hip's, measured inside the first matrix set-up of a process (VEXHIP_SETUP_TRACE: the first launch
// of the analysis kernel took 9.4 ms, every later one 4.9).  Each file defines an empty kernel; info(dev) launches them all when
// the library first meets a device (context creation), so that no set-up or product pays for the loading.
#define VEXHIP_WARM_TU(name)                                                   \
    namespace vexhip {                                                         \
    namespace { __global__ void warm_##name##_kernel() {} }                    \
    void warm_##name() { warm_##name##_kernel<<<1, 64>>>(); }                  \
    }

// launch check: hipGetLastError right after a <<<>>> launch
#define VEXHIP_LAUNCH_CHECK() VEXHIP_TRY(hipGetLastError())
