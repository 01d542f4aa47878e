// Shared host-side helpers of libdance_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

#ifdef __HIPCC__
// running max of |x| as a float whose bit pattern orders like an unsigned; a NaN wins and stays
__device__ __forceinline__ float absmax_acc(float m, float x)
{
    const unsigned a = __builtin_bit_cast(unsigned, m), b = __builtin_bit_cast(unsigned, x) & 0x7fffffffu;
    return __builtin_bit_cast(float, a > b ? a : b);
}

// Publishes a workgroup's max |value| (a non-negative float's bit pattern orders like an unsigned)
// into one of the image's DANCE_ABSMAX_WORDS partial-maximum words; `wm`: 16 words of LDS.  All
// threads of the workgroup must call (it holds a barrier).  The max of a set is independent of the
// order it is taken in: the words, and everything scaled by them, are deterministic.
__device__ __forceinline__ void publish_absmax(float local_max, unsigned *wm, uint32_t *words, unsigned spread)
{
    unsigned m = __builtin_bit_cast(unsigned, local_max) & 0x7fffffffu;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1)
        m = max(m, (unsigned)__shfl_xor((int)m, s, 64));
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0)
        wm[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; ++i)
            m = max(m, wm[i]);
        uint32_t *w = words + (spread % DANCE_ABSMAX_WORDS);
        // a word that already holds as much needs no atomic (a stale read only costs a redundant one)
        if (m > __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(w, m);
    }
}
#endif

#ifdef __HIPCC__
// Zero fill as a kernel of the library's own.  hipMemsetAsync is NOT used anywhere: captured into a HIP
// graph it becomes a memset node, and on this runtime (ROCm 7.0 / 7.2) such a node, replayed after
// any eager hipMemsetAsync has run in between, no longer writes zeros but pointer-sized words of its
// own launch state (seen: the absmax words of the layout pass holding the low half of their own
// address after a replay -- tools/graph_probe7.py; the deformation then ran at another input scale
// and the graph test lost bit-identity).  A kernel node replays as launched.
static __global__ __launch_bounds__(256) void k_zero_fill(unsigned char *p, size_t bytes)
{
    const size_t head = ((size_t)16 - ((uintptr_t)p & 15)) & 15;          // bytes up to the first 16-byte word
    const size_t lead = head < bytes ? head : bytes;
    const size_t vec = (bytes - lead) >> 4;
    uint4 *q = reinterpret_cast<uint4 *>(p + lead);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < vec; i += (size_t)gridDim.x * 256)
        q[i] = uint4{0u, 0u, 0u, 0u};
    if (blockIdx.x == 0) {
        if (threadIdx.x < lead)
            p[threadIdx.x] = 0;
        const size_t tail0 = lead + (vec << 4);
        if (tail0 + threadIdx.x < bytes && threadIdx.x < 16)
            p[tail0 + threadIdx.x] = 0;
    }
}

inline int zero_async(void *p, size_t bytes, hipStream_t stream, const char *fn)
{
    if (bytes == 0)
        return 0;
    size_t blocks = (bytes / 16 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(k_zero_fill, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned char *)p, bytes);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail((int)e, fn, hipGetErrorString(e));
}
#endif

// is x a normal positive power of two (an inverse weight scale the kernels take the exponent of)?
inline bool is_pow2_scale(float x)
{
    uint32_t b;
    memcpy(&b, &x, sizeof b);
    const uint32_t e = (b >> 23) & 0xffu;
    return (b >> 31) == 0 && (b & 0x7fffffu) == 0 && e != 0 && e != 0xffu;
}

// the policy an entry point runs under: the caller's, or the defaults (NULL)
inline dance_policy policy_or_default(const dance_policy *p)
{
    dance_policy d;
    d.few_contour_forms = 1;
    d.fuse_tail_split = 1;
    d.tail_group = 0;
    return p ? *p : d;
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
