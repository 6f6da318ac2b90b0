// Host-side plumbing shared by the translation units of libogg_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/ogg_hip.h"

namespace ogg {

int set_error(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace ogg

#define OGG_HIP_CHECK(expr)                                                                              \
    do {                                                                                                 \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess)                                                                           \
            return ogg::set_error(OGG_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

#define OGG_LAUNCH_CHECK() OGG_HIP_CHECK(hipGetLastError())

#define OGG_REQUIRE(cond, code, ...)                      \
    do {                                                  \
        if (!(cond)) return ogg::set_error(code, __VA_ARGS__); \
    } while (0)
