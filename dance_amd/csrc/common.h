// Shared host-side helpers of libdance_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "../../include/dance_hip.h"

namespace dance {

// thread-local text returned by dance_last_error() (defined in exact_ops.hip)
char *error_buffer();

inline int fail(int code, const char *fn, const char *what)
{
    snprintf(error_buffer(), 512, "%s: %s", fn, what);
    return code;
}

// after a launch: report launch-configuration errors without synchronising
inline int launch_check(const char *fn)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        return 0;
    return fail((int)e, fn, hipGetErrorString(e));
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// compute units of the current device (0 when the query fails); cached per device
inline int cu_count()
{
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return 0;
    if (cached[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return 0;
        cached[dev] = cus;
    }
    return cached[dev];
}

// A launch with more than 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize
// raised on the kernel -- per DEVICE (the attribute lives with the device's copy of the code
// object).  Issued once per (kernel, device), again only when a larger size is asked for;
// thread-safe; no allocation (safe under graph capture).  The one helper every file uses.
inline int allow_lds(const void *kernel, size_t bytes, const char *fn)
{
    if (bytes <= 64 * 1024)
        return 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return fail((int)e, fn, hipGetErrorString(e));
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> granted;
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = granted[std::make_pair(kernel, dev)];
    if (have >= bytes)
        return 0;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess)
        return fail((int)e, fn, hipGetErrorString(e));
    have = bytes;
    return 0;
}

// compile-time loop: f(std::integral_constant<int, B>{}), ..., f(<E - 1>)
template <int B, int E, typename F>
__host__ __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

}  // namespace dance

#define DANCE_REQUIRE(cond)                                                               \
    do {                                                                                  \
        if (!(cond))                                                                      \
            return dance::fail((int)hipErrorInvalidValue, __func__,                       \
                               "requirement failed: " #cond);                             \
    } while (0)
