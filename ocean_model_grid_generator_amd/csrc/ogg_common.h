// Host-side plumbing shared by the translation units of libogg_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/ogg_hip.h"

namespace ogg {

int set_error(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// OGG_SYM_DEFAULT / OGG_SYM_MIRROR / OGG_SYM_NONE of a caller -> mirror the caps' columns or not (include/ogg_hip.h)
bool cap_symmetry(int requested);

// Scratch from the stream-ordered allocator (no host synchronisation), returned to it on EVERY exit path of the call.
class AsyncScratch {
   public:
    explicit AsyncScratch(hipStream_t s) : s_(s) {}
    AsyncScratch(const AsyncScratch&) = delete;
    AsyncScratch& operator=(const AsyncScratch&) = delete;
    ~AsyncScratch() {
        if (p_) (void)hipFreeAsync(p_, s_);
    }
    int alloc(void** out, size_t bytes) {
        hipError_t e = hipMallocAsync(&p_, bytes ? bytes : 1, s_);
        if (e != hipSuccess) {
            p_ = nullptr;
            return set_error(e == hipErrorOutOfMemory ? OGG_ENOMEM : OGG_EHIP, "hipMallocAsync(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        }
        *out = p_;
        return OGG_OK;
    }

   private:
    void* p_ = nullptr;
    hipStream_t s_;
};

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
