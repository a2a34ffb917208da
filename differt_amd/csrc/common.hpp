// common.hpp -- error plumbing and launch helpers shared by every translation unit of
// libdiffert_amd.so.  gfx950 (MI355X) only; wave = 64 lanes.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/differt_amd.h"

namespace drt {

constexpr int kWave = 64;

std::string &last_error_ref();

inline int32_t fail(int32_t code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

#define DRT_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return ::drt::fail(DRT_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                               __FILE__, __LINE__);                                           \
    } while (0)

#define DRT_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return ::drt::fail(DRT_E_INVALID, __VA_ARGS__); \
    } while (0)

// checks the launch itself (configuration errors); execution errors surface at the next sync
#define DRT_LAUNCH_CHECK() DRT_HIP(hipGetLastError())

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// memset replacement that stays correct inside a replayed HIP graph (core.hip)
hipError_t fill_bytes_async(void *p, int byte, size_t n, hipStream_t s);

}  // namespace drt
