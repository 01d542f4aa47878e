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

// Per-image max |value| of a tensor a kernel produces, measured on the way out WITHOUT atomics: every
// workgroup stores its own maximum (a non-negative float's bit pattern orders like an unsigned) into
// its slot of a partials array, a second tiny kernel folds an image's partials into the
// DANCE_ABSMAX_WORDS words the consumers read.  (Atomic maxima into a handful of words were tried
// first: 60,800 workgroups of the layout pass on one 32-byte line per image cost +0.13 ms of 0.35,
// and any load of that line queues behind the atomics.)  The max of a set does not depend on the
// order it is taken in: deterministic.
//   absmax_stage    every thread: wave-level max -> wm[wave] (16 words of LDS); then a barrier;
//   absmax_partial  thread 0 after that barrier: merge, one plain store.
__device__ __forceinline__ void absmax_stage(float local_max, unsigned *wm)
{
    unsigned m = __builtin_bit_cast(unsigned, local_max) & 0x7fffffffu;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1)
        m = max(m, (unsigned)__shfl_xor((int)m, s, 64));
    if ((threadIdx.x & 63) == 0)
        wm[threadIdx.x >> 6] = m;
}

__device__ __forceinline__ void absmax_partial(const unsigned *wm, uint32_t *slot)
{
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        unsigned m = wm[0];
        for (int i = 1; i < nw; ++i)
            m = max(m, wm[i]);
        *slot = m;
    }
}

// grid: one workgroup of 256 per image; partial (b, per_image) -> words (b, DANCE_ABSMAX_WORDS)
static __global__ __launch_bounds__(256) void k_absmax_reduce(const uint32_t *__restrict__ partial, long long per_image,
                                                              uint32_t *__restrict__ words)
{
    const uint32_t *p = partial + (long long)blockIdx.x * per_image;
    unsigned m0 = 0, m1 = 0, m2 = 0, m3 = 0;        // four loads in flight per thread
    long long i = threadIdx.x;
    for (; i + 768 < per_image; i += 1024) {
        m0 = max(m0, p[i]);
        m1 = max(m1, p[i + 256]);
        m2 = max(m2, p[i + 512]);
        m3 = max(m3, p[i + 768]);
    }
    for (; i < per_image; i += 256)
        m0 = max(m0, p[i]);
    unsigned m = max(max(m0, m1), max(m2, m3));
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1)
        m = max(m, (unsigned)__shfl_xor((int)m, s, 64));
    if ((threadIdx.x & 63) == 0)
        words[blockIdx.x * DANCE_ABSMAX_WORDS + (threadIdx.x >> 6)] = m;
    if (threadIdx.x >= 4 && threadIdx.x < DANCE_ABSMAX_WORDS)
        words[blockIdx.x * DANCE_ABSMAX_WORDS + threadIdx.x] = 0u;
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

// Fixed-order sums for statistics that used to be fp64 atomics (run-to-run different in the last
// bits): every producer workgroup stores its fp32 partial sums, this kernel adds a row's `count`
// items of `NV` values each in fp64 -- thread t takes items t, t + 256, ... in order, then a fixed
// tree over the 256 threads.  partial (rows, count, NV) f32 -> out (rows, NV) f64.  grid: rows.
template <int NV>
static __global__ __launch_bounds__(256) void k_sum_rows(const float *__restrict__ partial, long long count,
                                                         double *__restrict__ out)
{
    __shared__ double red[256 * NV];
    const float *p = partial + (long long)blockIdx.x * count * NV;
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v)
        acc[v] = 0.0;
    for (long long i = threadIdx.x; i < count; i += 256)
#pragma unroll
        for (int v = 0; v < NV; ++v)
            acc[v] += (double)p[i * NV + v];
#pragma unroll
    for (int v = 0; v < NV; ++v)
        red[threadIdx.x * NV + v] = acc[v];
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int v = 0; v < NV; ++v)
                red[threadIdx.x * NV + v] += red[(threadIdx.x + s) * NV + v];
        __syncthreads();
    }
    if (threadIdx.x < NV)
        out[(long long)blockIdx.x * NV + threadIdx.x] = red[threadIdx.x];
}

template <int NV>
inline int sum_rows_async(const float *partial, long long rows, long long count, double *out, hipStream_t stream,
                          const char *fn)
{
    hipLaunchKernelGGL(k_sum_rows<NV>, dim3((unsigned)rows), dim3(256), 0, stream, partial, count, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail((int)e, fn, hipGetErrorString(e));
}

inline int absmax_reduce_async(const uint32_t *partial, long long per_image, uint32_t *words, int b, hipStream_t stream,
                               const char *fn)
{
    hipLaunchKernelGGL(k_absmax_reduce, dim3(b), dim3(256), 0, stream, partial, per_image, words);
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
    d.coop_stack = 1;
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
