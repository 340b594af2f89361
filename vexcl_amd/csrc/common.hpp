// Shared host-side helpers for libvexhip.so (error capture, device guard).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/vexhip.h"

namespace vexhip {

// Thread-local last-error text, in the reference's "file:line\n\t<code text>"
// shape (backend/cuda/error.hpp:119-145).
std::string &last_error();
int fail(const char *file, int line, const std::string &what);

inline int check(hipError_t e, const char *file, int line) {
    if (e == hipSuccess) return 0;
    (void)hipGetLastError();      // HIP's last-error slot is sticky: clear it so a later launch check does not report this failure again
    return fail(file, line, std::string(hipGetErrorName(e)) + ": " + hipGetErrorString(e));
}

constexpr int kWave = 64;   // gfx950 wavefront

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

struct device_info {
    int cus = 0;
    bool ok = false;
};
const device_info &info(int dev);

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

// launch check: hipGetLastError right after a <<<>>> launch
#define VEXHIP_LAUNCH_CHECK() VEXHIP_TRY(hipGetLastError())
