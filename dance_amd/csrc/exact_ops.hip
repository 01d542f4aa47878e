// exact_ops.hip -- the integer / index side of the contour path for gfx950:
// the four `extreme_utils` ops and the fused polygon resampler.
//
// Built with -ffp-contract=off and correctly rounded fp32 divide/sqrt: every float
// operation here is a separately rounded IEEE op, so counts, indices and resampled
// coordinates are bit-identical to the reference's torch/CUDA op sequence
// (oracle: oracle/extreme_utils_ref.c).
#include <math.h>
#include <mutex>

#include "common.h"

namespace dance {
static thread_local char g_error[512] = "";
char *error_buffer() { return g_error; }
}  // namespace dance

using dance::ceil_div;
using dance::launch_check;

extern "C" int dance_abi_version(void) { return 16; }
extern "C" const char *dance_last_error(void) { return dance::error_buffer(); }

extern "C" int dance_device_info(int *cu_count_host, int *lds_bytes_host)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return dance::fail((int)e, __func__, hipGetErrorString(e));
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess)
        return dance::fail((int)e, __func__, hipGetErrorString(e));
    if (cu_count_host)
        *cu_count_host = p.multiProcessorCount;
    if (lds_bytes_host)
        *lds_bytes_host = (int)p.sharedMemPerBlock;
    return 0;
}

// ------------------------------------------------------------------------------------
// collect_extreme_point: one wavefront per (batch, box, side).  The reference walks the
// band around a box side with one thread (utils.cu:28-111); here the 64 lanes sweep the
// band in memory order, each keeping its first best pixel, and a butterfly reduction
// picks the highest score, ties going to the earliest position in the REFERENCE's scan
// order -- which is what the serial strict '<' update (utils.cu:59,101) produces.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float cosine_to_centre(float cx, float cy, long long x, long long y,
                                                  float u, float v)
{
    float dx = cx - (float)x;
    float dy = cy - (float)y;
    float n1 = sqrtf(u * u + v * v);
    float n2 = sqrtf(dx * dx + dy * dy);
    float dot = u * dx + v * dy;
    return dot / (n1 * n2);
}

__global__ __launch_bounds__(256) void k_collect_extreme_point(
    const float *__restrict__ ext_hm, const long long *__restrict__ bbox,
    const long long *__restrict__ radius, const float *__restrict__ vote,
    const float *__restrict__ ct, long long *__restrict__ out, int n, int h, int w)
{
    const int lane = threadIdx.x & 63;
    const int dir = threadIdx.x >> 6;           // 0 top, 1 left, 2 bottom, 3 right
    const long long bn = blockIdx.x;            // = ib * n + in
    const long long ib = bn / n;
    const float *hm = ext_hm + (ib * 4 + dir) * (long long)h * w;
    const float *vt = vote + ib * (long long)h * w * 2;
    const long long *bx = bbox + bn * 4;
    // the reference narrows the int64 box to int (utils.cu:136-139), then widens again
    const long long x_min = (int)bx[0], y_min = (int)bx[1];
    const long long x_max = (int)bx[2], y_max = (int)bx[3];
    const long long r = radius[bn];
    const float cx = ct[bn * 2], cy = ct[bn * 2 + 1];
    const bool horizontal = (dir & 1) == 0;

    long long init_x, init_y, along_lo, along_n, side;
    if (horizontal) {
        side = dir == 0 ? y_min : y_max;
        init_x = (x_min + x_max) / 2;
        init_y = side;
        along_lo = x_min;
        along_n = x_max - x_min + 1;
    } else {
        side = dir == 1 ? x_min : x_max;
        init_x = side;
        init_y = (y_min + y_max) / 2;
        along_lo = y_min;
        along_n = y_max - y_min + 1;
    }
    const long long band = 2 * r + 1;
    const long long total = (along_n > 0 && band > 0) ? along_n * band : 0;

    float best = -INFINITY;
    long long best_pos = -1;
    for (long long q = lane; q < total; q += 64) {
        long long x, y, pos;
        if (horizontal) {                        // memory order == scan order
            long long i = q / along_n, j = q - i * along_n;
            y = side - r + i;
            x = along_lo + j;
            pos = q;
        } else {                                 // memory order: x fastest
            long long j = q / band, i = q - j * band;
            x = side - r + i;
            y = along_lo + j;
            pos = i * along_n + j;               // reference order: column outer, row inner
        }
        if (x < 0 || x > w - 1 || y < 0 || y > h - 1)
            continue;
        const long long p = y * w + x;
        const float s = hm[p];
        const float c = cosine_to_centre(cx, cy, x, y, vt[p * 2], vt[p * 2 + 1]);
        // utils.cu:54 compares against the double literal 0.9, :96 against int 0
        const bool skip = horizontal ? ((double)c < 0.9) : (c < 0.0f);
        if (skip)
            continue;
        if (s > best || (s == best && (best_pos < 0 || pos < best_pos))) {
            best = s;
            best_pos = pos;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float o_s = __shfl_xor(best, m, 64);
        const long long o_p = __shfl_xor(best_pos, m, 64);
        const bool take = o_p >= 0 && (best_pos < 0 || o_s > best || (o_s == best && o_p < best_pos));
        if (take) {
            best = o_s;
            best_pos = o_p;
        }
    }
    if (lane == 0) {
        const bool init_in = init_x >= 0 && init_x <= w - 1 && init_y >= 0 && init_y <= h - 1;
        const float init_score = init_in ? hm[init_y * w + init_x] : -INFINITY;
        long long rx = init_x, ry = init_y;
        if (best_pos >= 0 && init_score < best) {
            if (horizontal) {
                ry = side - r + best_pos / along_n;
                rx = along_lo + best_pos % along_n;
            } else {
                rx = side - r + best_pos / along_n;
                ry = along_lo + best_pos % along_n;
            }
        }
        long long *o = out + (bn * 4 + dir) * 2;
        o[0] = rx;
        o[1] = ry;
    }
}

extern "C" int dance_collect_extreme_point(const float *ext_hm, const int64_t *bbox,
                                           const int64_t *radius, const float *vote,
                                           const float *ct, int64_t *extreme_point, int b, int n,
                                           int h, int w, void *stream)
{
    DANCE_REQUIRE(b >= 0 && n >= 0 && h > 0 && w > 0);
    if (b == 0 || n == 0)
        return 0;
    DANCE_REQUIRE(ext_hm && bbox && radius && vote && ct && extreme_point);
    DANCE_REQUIRE((long long)b * n < (1ll << 31));
    hipLaunchKernelGGL(k_collect_extreme_point, dim3(b * n), dim3(256), 0, (hipStream_t)stream,
                       ext_hm, (const long long *)bbox, (const long long *)radius, vote, ct,
                       (long long *)extreme_point, n, h, w);
    return launch_check(__func__);
}

// ------------------------------------------------------------------------------------
// calculate_edge_num: serial by nature (walks the sorted edges until the surplus is
// gone); one lane per polygon exactly as utils.cu:215-255, but the walk stops at the
// last edge instead of running off the array when the surplus cannot be absorbed.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void fix_edge_counts(long long *cnt, const long long *order, int E,
                                                long long sum, long long p_num)
{
    if (sum == p_num)
        return;
    if (sum < p_num) {
        cnt[order[0]] += p_num - sum;
        return;
    }
    long long surplus = sum - p_num;
    for (int k = 0; k < E && surplus > 0; ++k) {
        const long long e = order[k];
        if (cnt[e] > surplus) {
            cnt[e] -= surplus;
            surplus = 0;
        } else {
            surplus -= cnt[e] - 1;
            cnt[e] = 1;
        }
    }
}

__global__ __launch_bounds__(256) void k_calculate_edge_num(long long *edge_num,
                                                            const long long *edge_num_sum,
                                                            const long long *edge_idx_sort,
                                                            long long polys, int E, long long p_num)
{
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= polys)
        return;
    fix_edge_counts(edge_num + i * E, edge_idx_sort + i * E, E, edge_num_sum[i], p_num);
}

extern "C" int dance_calculate_edge_num(int64_t *edge_num, const int64_t *edge_num_sum,
                                        const int64_t *edge_idx_sort, int b, int n,
                                        int orig_p_num, int64_t p_num, void *stream)
{
    DANCE_REQUIRE(b >= 0 && n >= 0 && orig_p_num > 0);
    const long long polys = (long long)b * n;
    if (polys == 0)
        return 0;
    DANCE_REQUIRE(edge_num && edge_num_sum && edge_idx_sort);
    hipLaunchKernelGGL(k_calculate_edge_num, dim3((unsigned)((polys + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (long long *)edge_num, (const long long *)edge_num_sum,
                       (const long long *)edge_idx_sort, polys, orig_p_num, (long long)p_num);
    return launch_check(__func__);
}

// ------------------------------------------------------------------------------------
// calculate_wnp: the reference gives one thread a whole edge and lets it loop over that
// edge's points (utils.cu:303-331); here each output slot is one lane that looks up its
// edge, so stores are coalesced and every slot is written exactly once.  Where edges
// overlap (inconsistent input) the highest edge index wins, as in a serial e=0..E-1 walk.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_calculate_wnp(const long long *edge_num,
                                                       const long long *edge_start_idx,
                                                       float *weight, long long *ind,
                                                       long long polys, int E, int p_num)
{
    const long long t = blockIdx.x * 256ll + threadIdx.x;
    if (t >= polys * p_num)
        return;
    const long long poly = t / p_num;
    const int slot = (int)(t - poly * p_num);
    const long long *num = edge_num + poly * E;
    const long long *start = edge_start_idx + poly * E;
    float wv = 0.0f;
    long long i0 = 0, i1 = 0;
    for (int e = 0; e < E; ++e) {
        const int s = (int)start[e];
        const long long k = num[e];
        if (slot >= s && (long long)slot - s < k) {
            wv = (float)(slot - s) / (float)k;
            i0 = e;
            i1 = (e + 1) % E;
        }
    }
    weight[t] = wv;
    ind[t * 2] = i0;
    ind[t * 2 + 1] = i1;
}

extern "C" int dance_calculate_wnp(const int64_t *edge_num, const int64_t *edge_start_idx,
                                   float *weight, int64_t *ind, int b, int n, int orig_p_num,
                                   int p_num, void *stream)
{
    DANCE_REQUIRE(b >= 0 && n >= 0 && orig_p_num > 0 && p_num >= 0);
    const long long total = (long long)b * n * p_num;
    if (total == 0)
        return 0;
    DANCE_REQUIRE(edge_num && edge_start_idx && weight && ind);
    DANCE_REQUIRE(total < (1ll << 38));
    hipLaunchKernelGGL(k_calculate_wnp, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const long long *)edge_num,
                       (const long long *)edge_start_idx, weight, (long long *)ind,
                       (long long)b * n, orig_p_num, p_num);
    return launch_check(__func__);
}

// ------------------------------------------------------------------------------------
// roll_array: new[b][(i + step[b]) mod n][:] = old[b][i][:]  (utils.cu:377-401)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_roll_array(const float *__restrict__ array,
                                                    const long long *__restrict__ step,
                                                    float *__restrict__ out, long long total,
                                                    int n, int d)
{
    const long long t = blockIdx.x * 256ll + threadIdx.x;
    if (t >= total)
        return;
    const long long row = t / d;
    const int col = (int)(t - row * d);
    const long long ib = row / n;
    const int i = (int)(row - ib * n);
    const int s = (int)step[ib];                 // utils.cu:396 narrows to int
    const int dst = ((i + s) % n + n) % n;
    out[(ib * n + dst) * d + col] = array[t];
}

extern "C" int dance_roll_array(const float *array, const int64_t *step, float *new_array, int b,
                                int n, int d, void *stream)
{
    DANCE_REQUIRE(b >= 0 && n >= 0 && d >= 0);
    const long long total = (long long)b * n * d;
    if (total == 0)
        return 0;
    DANCE_REQUIRE(array && step && new_array);
    DANCE_REQUIRE(total < (1ll << 38));
    hipLaunchKernelGGL(k_roll_array, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, array, (const long long *)step, new_array, total, n, d);
    return launch_check(__func__);
}

// ------------------------------------------------------------------------------------
// nms (utils.h:36-43, nms.cu:13-128): IoU bit mask over 64 x 64 tiles of the score-sorted
// boxes, then the greedy sweep -- on the device (the reference copies the mask to the host).
// ------------------------------------------------------------------------------------
// one = 1: the reference kernel's "+1" pixel convention (nms.cu:13-21); one = 0: the continuous
// convention of torchvision.ops.nms, which the reference's ml_nms reaches through detectron2
template <int ONE>
__device__ __forceinline__ float nms_iou(const float *a, const float *b)
{
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float width = right - left, height = bottom - top, wa = a[2] - a[0], ha = a[3] - a[1], wb = b[2] - b[0],
          hb = b[3] - b[1];
    if (ONE) {
        width += 1; height += 1; wa += 1; ha += 1; wb += 1; hb += 1;
    }
    width = fmaxf(width, 0.f);
    height = fmaxf(height, 0.f);
    const float interS = width * height;
    const float Sa = wa * ha;
    const float Sb = wb * hb;
    return interS / (Sa + Sb - interS);
}

// mask[i][col] bit j = IoU(box i, box 64 col + j) > thresh for j later than i in the sorted
// order (nms.cu:23-67); one 64 x 64 tile of the pairs per workgroup of 64 threads
template <int ONE>
__device__ __forceinline__ void nms_mask_tile(int n, int col_blocks, int row_start, int col_start, float thresh,
                                              const float *__restrict__ boxes, unsigned long long *__restrict__ mask)
{
    if (col_start < row_start)                  // words left of the diagonal are never read by the sweep
        return;
    const int row_size = min(n - row_start * 64, 64), col_size = min(n - col_start * 64, 64);
    __shared__ float bb[64 * 5];
    if ((int)threadIdx.x < col_size)
        for (int k = 0; k < 5; ++k)
            bb[threadIdx.x * 5 + k] = boxes[(64ll * col_start + threadIdx.x) * 5 + k];
    __syncthreads();
    if ((int)threadIdx.x < row_size) {
        const int cur = 64 * row_start + threadIdx.x;
        const float *cb = boxes + 5ll * cur;
        unsigned long long t = 0;
        const int start = row_start == col_start ? threadIdx.x + 1 : 0;
        for (int i = start; i < col_size; ++i)
            if (nms_iou<ONE>(cb, bb + i * 5) > thresh)
                t |= 1ULL << i;
        mask[(long long)cur * col_blocks + col_start] = t;
    }
}

// grid (col_blocks, col_blocks)
template <int ONE>
__global__ __launch_bounds__(64) void k_nms_mask(int n, float thresh, const float *__restrict__ boxes,
                                                 unsigned long long *__restrict__ mask)
{
    nms_mask_tile<ONE>(n, gridDim.x, blockIdx.y, blockIdx.x, thresh, boxes, mask);
}

// Segmented form: grid (max col_blocks, max col_blocks, segments); seg (segments, 3) i64 = first
// box, number of boxes, first mask word of each segment (every segment an independent NMS)
template <int ONE>
__global__ __launch_bounds__(64) void k_nms_mask_seg(const long long *__restrict__ seg, float thresh,
                                                     const float *__restrict__ boxes,
                                                     unsigned long long *__restrict__ mask)
{
    const long long first = seg[3 * blockIdx.z];
    const int n = (int)seg[3 * blockIdx.z + 1];
    const int cb = (n + 63) / 64;
    if ((int)blockIdx.x >= cb || (int)blockIdx.y >= cb)
        return;
    nms_mask_tile<ONE>(n, cb, blockIdx.y, blockIdx.x, thresh, boxes + 5 * first, mask + seg[3 * blockIdx.z + 2]);
}

// one wavefront: blocks of 64 boxes in order.  Inside a block the 64 decisions are serial but
// touch only the block's diagonal mask words (one per lane); the kept rows are then OR-ed
// into the removal words of the later blocks, lanes across the columns (nms.cu:105-121).
__device__ __forceinline__ void nms_sweep_body(int n, int col_blocks, const unsigned long long *__restrict__ mask,
                                                  unsigned long long *__restrict__ remv, long long *__restrict__ keep,
                                                  int *__restrict__ count)
{
    const int lane = threadIdx.x;
    for (int j = lane; j < col_blocks; j += 64)
        remv[j] = 0;
    __syncthreads();
    int num = 0;
    for (int kb = 0; kb < col_blocks; ++kb) {
        const int row = kb * 64 + lane, size = min(n - kb * 64, 64);
        const unsigned long long diag = lane < size ? mask[(long long)row * col_blocks + kb] : 0ull;
        unsigned long long r = remv[kb], kept = 0;
        for (int i = 0; i < size; ++i) {
            const unsigned long long di = __shfl(diag, i, 64);
            if (!((r >> i) & 1ull)) {
                kept |= 1ull << i;
                r |= di;
            }
        }
        if (lane < size && ((kept >> lane) & 1ull))
            keep[num + __popcll(kept & ((1ull << lane) - 1ull))] = row;
        num += __popcll(kept);
        for (int j = kb + 1 + lane; j < col_blocks; j += 64) {
            unsigned long long acc = remv[j];
            // eight independent row loads in flight, masked by the kept bits afterwards (a load
            // per kept row behind a branch left this loop latency-bound)
            for (int i0 = 0; i0 < size; i0 += 8) {
                unsigned long long m[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    m[k] = mask[(long long)(kb * 64 + min(i0 + k, size - 1)) * col_blocks + j];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    acc |= (i0 + k < size && ((kept >> (i0 + k)) & 1ull)) ? m[k] : 0ull;
            }
            remv[j] = acc;
        }
        __syncthreads();
    }
    if (lane == 0)
        *count = num;
}

__global__ __launch_bounds__(64) void k_nms_sweep(int n, int col_blocks, const unsigned long long *__restrict__ mask,
                                                  unsigned long long *__restrict__ remv, long long *__restrict__ keep,
                                                  int *__restrict__ count)
{
    nms_sweep_body(n, col_blocks, mask, remv, keep, count);
}

// one workgroup (one wavefront) per segment; keep positions are relative to the segment
__global__ __launch_bounds__(64) void k_nms_sweep_seg(const long long *__restrict__ seg,
                                                      unsigned long long *__restrict__ mask,
                                                      long long *__restrict__ keep, int *__restrict__ count)
{
    const long long first = seg[3 * blockIdx.x];
    const int n = (int)seg[3 * blockIdx.x + 1];
    const int cb = (n + 63) / 64;
    unsigned long long *m = mask + seg[3 * blockIdx.x + 2];
    if (n == 0) {
        if (threadIdx.x == 0)
            count[blockIdx.x] = 0;
        return;
    }
    nms_sweep_body(n, cb, m, m + (long long)n * cb, keep + first, count + blockIdx.x);
}

static int nms_sorted_impl(const float *boxes_sorted, int n, float thresh, int plus_one, uint64_t *mask_ws,
                           int64_t *keep, int32_t *count, void *stream);

extern "C" int dance_nms_sorted(const float *boxes_sorted, int n, float thresh, uint64_t *mask_ws,
                                int64_t *keep, int32_t *count, void *stream)
{
    return nms_sorted_impl(boxes_sorted, n, thresh, 1, mask_ws, keep, count, stream);
}

extern "C" int dance_nms_sorted_plain(const float *boxes_sorted, int n, float thresh, uint64_t *mask_ws,
                                      int64_t *keep, int32_t *count, void *stream)
{
    return nms_sorted_impl(boxes_sorted, n, thresh, 0, mask_ws, keep, count, stream);
}

static int nms_sorted_impl(const float *boxes_sorted, int n, float thresh, int plus_one, uint64_t *mask_ws,
                           int64_t *keep, int32_t *count, void *stream)
{
    DANCE_REQUIRE(n >= 0 && n <= (1 << 20));      // 2^20 boxes would need a 137 GB mask: the Python
    DANCE_REQUIRE(count != nullptr);              // wrappers stop at 65,536 (dance_amd/ml_nms.py)
    if (n == 0)
        return dance::zero_async(count, sizeof(int32_t), (hipStream_t)stream, __func__);
    DANCE_REQUIRE(boxes_sorted && mask_ws && keep);
    const int cb = (n + 63) / 64;
    unsigned long long *mask = (unsigned long long *)mask_ws;
    if (plus_one)
        hipLaunchKernelGGL(k_nms_mask<1>, dim3(cb, cb), dim3(64), 0, (hipStream_t)stream, n, thresh, boxes_sorted, mask);
    else
        hipLaunchKernelGGL(k_nms_mask<0>, dim3(cb, cb), dim3(64), 0, (hipStream_t)stream, n, thresh, boxes_sorted, mask);
    hipLaunchKernelGGL(k_nms_sweep, dim3(1), dim3(64), 0, (hipStream_t)stream, n, cb, mask,
                       mask + (long long)n * cb, (long long *)keep, count);
    return launch_check(__func__);
}

// ------------------------------------------------------------------------------------
// dance_nms_sorted_plain for callers that cannot allocate (HIP graph capture, tight serving
// loops): the IoU bit mask and the removal words live in a file-scope device array instead of
// a caller workspace.  No counterpart in the reference.  Every GPU holds its own copy of the
// array at its own address; the address is looked up once per device and that pointer -- never
// the array's name, which on the host is only a stand-in -- is what the kernels receive.
// ------------------------------------------------------------------------------------
// DANCE_NMS_STATIC_MAX (include/dance_hip.h) = 4096 boxes: (4096 + 1) * 64 words = 2.0 MiB per GPU
#define DANCE_MAX_DEVICES 64
static __device__ __align__(256) unsigned long long scratch[(DANCE_NMS_STATIC_MAX + 1) * (DANCE_NMS_STATIC_MAX / 64)];
static unsigned long long *g_scratch_ptr[DANCE_MAX_DEVICES];     // per device, set once
static std::mutex g_scratch_mutex;

static int nms_static_scratch(unsigned long long **out, const char *fn)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return dance::fail((int)e, fn, hipGetErrorString(e));
    if (dev < 0 || dev >= DANCE_MAX_DEVICES)
        return dance::fail((int)hipErrorInvalidDevice, fn, "device ordinal beyond DANCE_MAX_DEVICES");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        unsigned long long *p = nullptr;
        e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));     // the current device's copy
        if (e != hipSuccess || !p)
            return dance::fail((int)(e != hipSuccess ? e : hipErrorInvalidSymbol), fn,
                               "hipGetSymbolAddress(scratch) failed");
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" int dance_nms_sorted_plain_static(const float *boxes_sorted, int n, float thresh, int64_t *keep,
                                             int32_t *count, void *stream)
{
    if (n > DANCE_NMS_STATIC_MAX)
        return dance::fail((int)hipErrorInvalidValue, __func__,
                           "n exceeds the static scratch (at most 4096 boxes; larger lists: "
                           "dance_nms_sorted_plain with a caller workspace)");
    DANCE_REQUIRE(n >= 0 && count != nullptr);
    DANCE_REQUIRE(n == 0 || (boxes_sorted && keep));
    unsigned long long *ws = nullptr;
    if (int rc = nms_static_scratch(&ws, __func__))
        return rc;
    if (stream != nullptr) {        // the array is the CURRENT device's: a stream of another device would get the wrong GPU's
        int cur = -1, sdev = -1;
        if (hipGetDevice(&cur) == hipSuccess && hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess &&
            sdev != cur)
            return dance::fail((int)hipErrorInvalidDevice, __func__,
                               "the stream belongs to another device than the current one");
    }
    // nothing in the scratch has to start at zero: k_nms_mask writes every word the sweep reads
    // (diagonal and right of it) and the sweep clears its removal words itself
    return nms_sorted_impl(boxes_sorted, n, thresh, 0, (uint64_t *)ws, keep, count, stream);
}

extern "C" int dance_nms_segments_plain(const float *boxes_sorted, const int64_t *seg, int n_seg, int n_max,
                                        float thresh, uint64_t *mask_ws, int64_t *keep, int32_t *count,
                                        void *stream)
{
    DANCE_REQUIRE(n_seg >= 0 && n_seg <= 65535 && n_max >= 0 && n_max <= (1 << 20));
    if (n_seg == 0)
        return 0;
    DANCE_REQUIRE(seg && count);
    DANCE_REQUIRE(n_max == 0 || (boxes_sorted && mask_ws && keep));
    const int cb = (n_max + 63) / 64;
    if (cb > 0)
        hipLaunchKernelGGL(k_nms_mask_seg<0>, dim3(cb, cb, n_seg), dim3(64), 0, (hipStream_t)stream,
                           (const long long *)seg, thresh, boxes_sorted, (unsigned long long *)mask_ws);
    hipLaunchKernelGGL(k_nms_sweep_seg, dim3(n_seg), dim3(64), 0, (hipStream_t)stream, (const long long *)seg,
                       (unsigned long long *)mask_ws, (long long *)keep, count);
    return launch_check(__func__);
}

// ------------------------------------------------------------------------------------
// contour_upsample_2x (snake_head.py:1752-1761): out[2i] = p[i], out[2i+1] = (p[i] + p[i+1 mod V]) / 2
// (one fp32 add, one exact halving -- this file is compiled with contraction off)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_contour_upsample_2x(const float *__restrict__ loc,
                                                             float *__restrict__ out, long long total, int v)
{
    const long long t = blockIdx.x * 256ll + threadIdx.x;       // one output coordinate pair
    if (t >= total)
        return;
    const long long nn = t / (2 * v);
    const int j = (int)(t - nn * 2 * v), i = j >> 1;
    const float2 a = reinterpret_cast<const float2 *>(loc)[nn * v + i];
    float2 o = a;
    if (j & 1) {
        const float2 b = reinterpret_cast<const float2 *>(loc)[nn * v + (i + 1 == v ? 0 : i + 1)];
        o.x = (a.x + b.x) / 2.0f;
        o.y = (a.y + b.y) / 2.0f;
    }
    reinterpret_cast<float2 *>(out)[t] = o;
}

extern "C" int dance_contour_upsample_2x(const float *loc, float *out, int n, int v, void *stream)
{
    DANCE_REQUIRE(n >= 0 && v >= 1);
    const long long total = 2ll * n * v;
    if (total == 0)
        return 0;
    DANCE_REQUIRE(loc && out && total < (1ll << 38));
    hipLaunchKernelGGL(k_contour_upsample_2x, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, loc, out, total, v);
    return launch_check(__func__);
}

// ------------------------------------------------------------------------------------
// Fused resampler (snake_head.py:764-794 + utils.cu:215-331 + the polygon builders):
// one wavefront per polygon.  Lanes build the polygon and its edge lengths in LDS; lane 0
// runs the short serial part (left-to-right length sum, round-half-even counts, stable
// descending sort, fix-up, prefix sum); then all lanes emit points, one slot per lane per
// step, each finding its edge by a scan of the (<= 64) start indices.
// ------------------------------------------------------------------------------------
#define RS_MAX_E 64

// kind OCTAGON only: what DsnakeHead.forward does to the init snake's extreme points before
// get_octagon (dsnake_head.py:800-817) -- clamp to the image, take the ground-truth points where
// they are given -- done while the points are loaded, and the points actually used written out.
struct RsExtremes {
    const float *override_pts;           // (n, 4, 2) or null
    const unsigned char *override_mask;  // (n) non-zero: take override_pts[poly]
    float *used;                         // (n, 4, 2) out or null
    float x_max, y_max;                  // clamp to [0, x_max] x [0, y_max] ...
    int clamp;                           // ... when non-zero
    long long poly_stride, point_stride; // floats between instances / between the 4 points in src
};

#ifdef DANCE_EXP_LERP
__attribute__((target("packed-fp32-ops")))          // the experiment's asm needs the feature the build switches off
#endif
__global__ __launch_bounds__(64) void k_uniform_upsample(int kind, const float *__restrict__ src,
                                                         int E, int p_num, float *__restrict__ out,
                                                         long long *__restrict__ start_out,
                                                         int *__restrict__ status, RsExtremes x)
{
    __shared__ float px[RS_MAX_E], py[RS_MAX_E], len[RS_MAX_E];
    __shared__ long long cnt[RS_MAX_E], start[RS_MAX_E];
    __shared__ int order[RS_MAX_E];
    __shared__ int ok;
    const int lane = threadIdx.x;
    const long long poly = blockIdx.x;

    if (kind == DANCE_POLY_RAW) {
        if (lane < E) {
            px[lane] = src[(poly * E + lane) * 2];
            py[lane] = src[(poly * E + lane) * 2 + 1];
        }
    } else if (kind == DANCE_POLY_BOX) {
        // corners (xmax,ymin) (xmin,ymin) (xmin,ymax) (xmax,ymax): snake_head.py:878.
        // Values are loaded first and picked with single-condition selects (a pointer
        // select under `lane == 0 || lane == 3` was miscompiled by hipcc 7.2).
        if (lane < 4) {
            const float *b = src + poly * 4;
            const float x0 = b[0], y0 = b[1], x1 = b[2], y1 = b[3];
            px[lane] = ((0x9 >> lane) & 1) ? x1 : x0;
            py[lane] = ((0x3 >> lane) & 1) ? y0 : y1;
        }
    } else if (kind == DANCE_POLY_QUAD) {
        // edge mid-points top, left, bottom, right: dsnake_head.py:506-515
        if (lane < 4) {
            const float *b = src + poly * 4;
            const float x0 = b[0], y0 = b[1], x1 = b[2], y1 = b[3];
            const float mx = (x0 + x1) / 2.0f, my = (y0 + y1) / 2.0f;
            float x = mx, y = my;
            x = lane == 1 ? x0 : x;
            x = lane == 3 ? x1 : x;
            y = lane == 0 ? y0 : y;
            y = lane == 2 ? y1 : y;
            px[lane] = x;
            py[lane] = y;
        }
    } else {
        // 12-gon from extreme points t,l,b,r: snake_head.py:800-830
        if (lane < 12) {
            const float *e = src + poly * x.poly_stride;
            float ev[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float c = e[(k >> 1) * x.point_stride + (k & 1)];
                if (x.clamp) {                              // tensor.clamp_(min=0, max=..): NaN stays NaN
                    const float hi = (k & 1) ? x.y_max : x.x_max;
                    c = c < 0.0f ? 0.0f : (c > hi ? hi : c);
                }
                ev[k] = c;
            }
            if (x.override_pts != nullptr && x.override_mask[poly]) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    ev[k] = x.override_pts[poly * 8 + k];
            }
            if (x.used != nullptr) {
#pragma unroll
                for (int k = 0; k < 8; ++k)                 // (no dynamic register-array index)
                    if (lane == k)
                        x.used[poly * 8 + k] = ev[k];
            }
            const float tx = ev[0], ty = ev[1], lx = ev[2], ly = ev[3];
            const float bx = ev[4], by = ev[5], rx = ev[6], ry = ev[7];
            const float w8 = (rx - lx) / 8.0f, h8 = (by - ty) / 8.0f;
            float x, y;
            switch (lane) {
            case 0: x = tx; y = ty; break;
            case 1: x = fmaxf(tx - w8, lx); y = ty; break;
            case 2: x = lx; y = fmaxf(ly - h8, ty); break;
            case 3: x = lx; y = ly; break;
            case 4: x = lx; y = fminf(ly + h8, by); break;
            case 5: x = fmaxf(bx - w8, lx); y = by; break;
            case 6: x = bx; y = by; break;
            case 7: x = fminf(bx + w8, rx); y = by; break;
            case 8: x = rx; y = fminf(ry + h8, by); break;
            case 9: x = rx; y = ry; break;
            case 10: x = rx; y = fmaxf(ry - h8, ty); break;
            default: x = fminf(tx + w8, rx); y = ty; break;
            }
            px[lane] = x;
            py[lane] = y;
        }
    }
    __syncthreads();
    if (lane < E) {
        const int nx = lane + 1 == E ? 0 : lane + 1;
        const float dx = px[nx] - px[lane], dy = py[nx] - py[lane];
        const float sx = dx * dx, sy = dy * dy;
        len[lane] = sqrtf(sx + sy);
    }
    __syncthreads();
    if (lane == 0) {
        float total = 0.0f;
        for (int e = 0; e < E; ++e)
            total = total + len[e];
        long long sum = 0;
        for (int e = 0; e < E; ++e) {
            const float q = (len[e] * (float)p_num) / total;
            const float r = rintf(q);                       // torch.round: half to even
            // NaN/inf -> INT64_MIN is what torch's .long() yields on the reference's CPU path
            long long c = (r == r && fabsf(r) < 9.0e18f) ? (long long)r : (long long)(-9223372036854775807ll - 1);
            if (c < 1)
                c = 1;
            cnt[e] = c;
            sum += c;
            order[e] = e;
        }
        // stable descending insertion sort == torch.argsort(descending=True), ties low index first
        for (int a = 1; a < E; ++a) {
            const int key = order[a];
            int j = a - 1;
            while (j >= 0 && cnt[order[j]] < cnt[key]) {
                order[j + 1] = order[j];
                --j;
            }
            order[j + 1] = key;
        }
        if (sum != (long long)p_num) {
            if (sum < (long long)p_num) {
                cnt[order[0]] += p_num - sum;
            } else {
                long long surplus = sum - p_num;
                for (int k = 0; k < E && surplus > 0; ++k) {
                    const int e = order[k];
                    if (cnt[e] > surplus) {
                        cnt[e] -= surplus;
                        surplus = 0;
                    } else {
                        surplus -= cnt[e] - 1;
                        cnt[e] = 1;
                    }
                }
            }
        }
        long long acc = 0;
        for (int e = 0; e < E; ++e) {
            start[e] = acc;
            acc += cnt[e];
        }
        ok = acc == (long long)p_num;
        status[poly] = ok ? 0 : 1;
    }
    __syncthreads();
    if (start_out != nullptr && lane < E)
        start_out[poly * E + lane] = start[lane];
    float *o = out + poly * p_num * 2;
    if (!ok) {
        for (int s = lane; s < p_num * 2; s += 64)
            o[s] = 0.0f;
        return;
    }
    for (int s = lane; s < p_num; s += 64) {
        int e = 0;
        for (int k = 1; k < E; ++k)
            e = (start[k] <= s) ? k : e;
        const int nx = e + 1 == E ? 0 : e + 1;
        const float wgt = (float)(s - (int)start[e]) / (float)cnt[e];
        const float om = 1.0f - wgt;
#if defined(DANCE_EXP_LERP) && DANCE_EXP_LERP != 5     // 5: plain source, the compiler's own packed forms
        // experiment build only (tools/build_variant.py): the lerp in the packed fp32 forms hipcc chose before the
        // library was built without them, written out so that one ingredient at a time can change.
        //   1: as compiled then -- second product with the {wgt, om} pair as BOTH source (halves swapped through
        //      op_sel) and destination;  2: the same with a destination of its own;  3: no op_sel (the pair
        //      swapped by moves first), destination = source pair;  4: as 1 with 8 idle cycles around each packed op
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 t2 = {wgt, om}, a2 = {px[nx], py[e]}, b2 = {px[e], py[nx]}, r2;
#if DANCE_EXP_LERP == 1
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(t2));
        asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(t2) : "v"(b2));
        asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(t2) : "v"(a2));
        r2 = t2;
#elif DANCE_EXP_LERP == 2
        f2 m2;
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(t2));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(m2) : "v"(b2), "v"(t2));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(r2) : "v"(a2), "v"(m2));
#elif DANCE_EXP_LERP == 6
        // op_sel on source 0 instead of source 1: b2 arrives swapped, {py[nx], px[e]}, and is swapped back by op_sel
        f2 bs = {b2.y, b2.x}, m2;
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(t2));
        f2 ts = {om, wgt};
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(m2) : "v"(bs), "v"(ts));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(r2) : "v"(a2), "v"(m2));
#elif DANCE_EXP_LERP == 7
        // only the HIGH result selects a non-default half (a broadcast of source 1's low half, the form of the first
        // sighting's GEMV): lo = px[e] * om, hi = py[nx] * wgt computed as {b2.x, b2.y} * {om, om} and {.., wgt} apart
        f2 m2, oo = {om, wgt}, ww = {wgt, om}, c2 = {px[e], py[e]}, d2 = {px[nx], py[nx]};
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(m2) : "v"(c2), "v"(oo));       // {px[e] om, py[e] om}
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(a2) : "v"(d2), "v"(ww));       // {px[nx] wgt, py[nx] wgt}
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(r2) : "v"(m2), "v"(a2));
#elif DANCE_EXP_LERP == 8
        // ONLY the low result re-selects: lo = c2.lo * t2.HI, hi = c2.hi * t2.hi (op_sel:[0,1], op_sel_hi default);
        // the other product broadcasts t2's LOW half to the high result (the form build 7 showed clean)
        f2 m2, c2 = {px[e], py[e]}, d2 = {px[nx], py[nx]};
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(m2) : "v"(c2), "v"(t2));          // {px[e] om, py[e] om}
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(a2) : "v"(d2), "v"(t2));       // {px[nx] wgt, py[nx] wgt}
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(r2) : "v"(m2), "v"(a2));
#elif DANCE_EXP_LERP == 3
        f2 sw = {om, wgt};
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(t2));
        asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(sw) : "v"(b2));
        asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(sw) : "v"(a2));
        r2 = sw;
#else
        asm volatile("s_nop 7\n\tv_pk_mul_f32 %0, %0, %1\n\ts_nop 7" : "+v"(a2) : "v"(t2));
        asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 7" : "+v"(t2) : "v"(b2));
        asm volatile("v_pk_add_f32 %0, %1, %0\n\ts_nop 7" : "+v"(t2) : "v"(a2));
        r2 = t2;
#endif
        o[s * 2] = r2.x;
        o[s * 2 + 1] = r2.y;
#else
        const float x = px[e] * om, x2 = px[nx] * wgt;
        const float y = py[e] * om, y2 = py[nx] * wgt;
        o[s * 2] = x + x2;
        o[s * 2 + 1] = y + y2;
#endif
    }
}

extern "C" int dance_uniform_upsample(int kind, const float *src, int n, int e, int p_num,
                                      float *out, int64_t *edge_start_idx, int32_t *status,
                                      void *stream)
{
    DANCE_REQUIRE(kind >= DANCE_POLY_RAW && kind <= DANCE_POLY_OCTAGON);
    DANCE_REQUIRE(n >= 0 && p_num >= 1 && p_num <= 4096);
    if (kind == DANCE_POLY_BOX || kind == DANCE_POLY_QUAD)
        DANCE_REQUIRE(e == 4);
    if (kind == DANCE_POLY_OCTAGON)
        DANCE_REQUIRE(e == 12);
    DANCE_REQUIRE(e >= 1 && e <= RS_MAX_E);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(src && out && status);
    RsExtremes x = {nullptr, nullptr, nullptr, 0.0f, 0.0f, 0, 8, 2};
    hipLaunchKernelGGL(k_uniform_upsample, dim3(n), dim3(64), 0, (hipStream_t)stream, kind, src, e,
                       p_num, out, (long long *)edge_start_idx, (int *)status, x);
    return launch_check(__func__);
}

extern "C" int dance_octagon_upsample(const float *ext_points, int64_t poly_stride, int64_t point_stride,
                                      const float *override_pts,
                                      const uint8_t *override_mask, int clamp, float x_max, float y_max,
                                      int n, int p_num, float *out, float *ext_used,
                                      int64_t *edge_start_idx, int32_t *status, void *stream)
{
    DANCE_REQUIRE(n >= 0 && p_num >= 1 && p_num <= 4096);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(ext_points && out && status);
    DANCE_REQUIRE((override_pts == nullptr) == (override_mask == nullptr));
    DANCE_REQUIRE(point_stride >= 2 && poly_stride >= 2);
    RsExtremes x = {override_pts, override_mask, ext_used, x_max, y_max, clamp, poly_stride, point_stride};
    hipLaunchKernelGGL(k_uniform_upsample, dim3(n), dim3(64), 0, (hipStream_t)stream, DANCE_POLY_OCTAGON,
                       ext_points, 12, p_num, out, (long long *)edge_start_idx, (int *)status, x);
    return launch_check(__func__);
}
