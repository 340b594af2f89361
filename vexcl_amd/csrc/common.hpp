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
struct setup_trace {
    bool on; hipStream_t s; std::chrono::steady_clock::time_point t0, last;
    explicit setup_trace(hipStream_t st) : on(std::getenv("VEXHIP_SETUP_TRACE") != nullptr), s(st) { t0 = last = std::chrono::steady_clock::now(); }
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
