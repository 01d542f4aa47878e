// raster.hip -- polygon -> run-length-encoded mask, the step right after the contour path
// (SURVEY.md 8f row 2).  reference call sites: get_polygon_rles, one
// `mask_util.merge(mask_util.frPyObjects([p.tolist()], h, w))` per instance on the CPU after a
// device -> host copy (core/evaluation/coco_evaluation.py:311-318, 352-357;
// core/modeling/postprocessing.py:266-274; core/modeling/dsnake_baseline/postprocessing.py:13-21).
// The arithmetic is pycocotools' (cocoapi common/maskApi.c: rleFrPoly, rleToString, rleDecode),
// a dependency that is not part of the reference tree; results are bit-identical to the C
// restatement of that published algorithm (oracle/rle_ref.c).  Compiled with
// -ffp-contract=off: the library rounds every double operation separately.
//
// The library builds the dense 5x-upsampled boundary, collects the points where it crosses a
// pixel column, sorts their column-major positions x*h + y and folds equal positions away in
// pairs.  Sorting + pair folding = "a position toggles the mask iff it occurs an odd number of
// times", so here each workgroup (one polygon) XORs crossings into a bitmap of column-major
// positions held in LDS and reads the runs off the set bits in order:
//   1. vertices -> 5x integer grid (double arithmetic as the library), per-edge step counts,
//      slopes and a prefix sum of the steps in LDS;
//   2. the steps of all edges, flattened, are dealt to the 256 threads; a step that changes x
//      onto a pixel-column boundary toggles bit x*h + y (ds_xor);
//   3. threads scan contiguous slices of the bitmap: popcount prefix sum -> output slot,
//      running maximum -> previous toggle, and write run lengths straight to HBM.
// A polygon wider than the bitmap (40 KB = 327,680 positions) is processed in column bands;
// positions, hence runs, come out in order band after band.
#include "common.h"

using dance::launch_check;

namespace {

constexpr int RT_THREADS = 256;
constexpr int RT_MAX_V = 512;
constexpr int RT_BM_WORDS = 10240;                 // 40 KB of LDS: three workgroups per CU
constexpr long long RT_BM_BITS = RT_BM_WORDS * 32ll;
constexpr double RT_CLAMP = 1048576.0;             // 5x grid: +-209,715 px; 512 edges x 2^21 steps < 2^31

__device__ __forceinline__ int wave_incl_add(int x, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(x, o, 64);
        x += lane >= o ? t : 0;
    }
    return x;
}

__device__ __forceinline__ int wave_incl_max(int x, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(x, o, 64);
        x = lane >= o ? max(x, t) : x;
    }
    return x;
}

// Inclusive sum and inclusive max over the 256 threads in thread order (sc: 8 ints of LDS);
// tot / top = the block-wide results.
__device__ __forceinline__ void block_scan(int &sum, int &mx, int &tot, int &top, int *sc, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    sum = wave_incl_add(sum, lane);
    mx = wave_incl_max(mx, lane);
    __syncthreads();                               // sc free again
    if (lane == 63) {
        sc[wave] = sum;
        sc[4 + wave] = mx;
    }
    __syncthreads();
    int add = 0, pm = INT_MIN;
    tot = 0;
    top = INT_MIN;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < wave) {
            add += sc[k];
            pm = max(pm, sc[4 + k]);
        }
        tot += sc[k];
        top = max(top, sc[4 + k]);
    }
    sum += add;
    mx = max(mx, pm);
}

// Point t of an edge walked along its major axis from (xs, ys): the library's
// (int)(start + slope * t + .5) in double, one rounding per operation.
__device__ __forceinline__ void edge_point(bool xmajor, int xs, int ys, double slope, int t, int &u, int &v)
{
    if (xmajor) {
        u = t + xs;
        v = (int)((double)ys + slope * (double)t + .5);
    } else {
        v = t + ys;
        u = (int)((double)xs + slope * (double)t + .5);
    }
}

__global__ __launch_bounds__(RT_THREADS) void k_polygon_rle(const float *__restrict__ polys,
                                                            const int *__restrict__ sizes, int v, int h0,
                                                            int w0, unsigned *__restrict__ counts, int cap,
                                                            int *__restrict__ m_out,
                                                            unsigned *__restrict__ area_out,
                                                            int *__restrict__ status)
{
    __shared__ int vx[RT_MAX_V + 1], vy[RT_MAX_V + 1], pre[RT_MAX_V + 1];
    __shared__ double slope[RT_MAX_V];
    __shared__ unsigned bm[RT_BM_WORDS];
    __shared__ int sc[8], smax[RT_THREADS];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int h = sizes ? sizes[2 * n] : h0, w = sizes ? sizes[2 * n + 1] : w0;
    const long long hw = (long long)h * w;
    unsigned *out = counts + (long long)n * cap;

    // 1. vertices on the 5x grid
    int umin = INT_MAX, umax = INT_MIN;
    for (int j = tid; j < v; j += RT_THREADS) {
        const float2 p = reinterpret_cast<const float2 *>(polys)[(long long)n * v + j];
        // the library's (int)(5 * c + .5); magnitudes no image reaches (and NaN / inf, where the
        // library's conversion is undefined) are clamped so that the step count stays in int range
        const int x = (int)fmin(fmax(5.0 * (double)p.x + .5, -RT_CLAMP), RT_CLAMP);
        const int y = (int)fmin(fmax(5.0 * (double)p.y + .5, -RT_CLAMP), RT_CLAMP);
        vx[j] = x;
        vy[j] = y;
        if (j == 0) {
            vx[v] = x;
            vy[v] = y;
        }
        umin = min(umin, x);
        umax = max(umax, x);
    }
    {
        int a = -umin, b = umax, ta, tb;            // max-scan of -x gives the minimum
        int s0 = 0, s1 = 0, t0, t1;
        block_scan(s0, a, t0, ta, sc, tid);
        block_scan(s1, b, t1, tb, sc, tid);
        umin = -ta;
        umax = tb;
    }
    // 2. steps per edge, slopes, prefix sum of the steps
    int total_steps = 0;
    for (int base = 0; base < v; base += RT_THREADS) {
        const int j = base + tid;
        int len = 0;
        if (j < v) {
            int xs = vx[j], xe = vx[j + 1], ys = vy[j], ye = vy[j + 1];
            const int dx = abs(xe - xs), dy = abs(ys - ye);
            len = max(dx, dy);
            const bool back = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
            if (back) {
                int t = xs; xs = xe; xe = t;
                t = ys; ys = ye; ye = t;
            }
            slope[j] = len == 0 ? 0.0 : (dx >= dy ? (double)(ye - ys) / (double)dx : (double)(xe - xs) / (double)dy);
        }
        int incl = len, dummy = 0, tot, top;
        block_scan(incl, dummy, tot, top, sc, tid);
        if (j < v)
            pre[j] = total_steps + incl - len;
        total_steps += tot;
    }
    if (tid == 0)
        pre[v] = total_steps;
    __syncthreads();

    // columns that can hold a crossing: x' = 5 * col + 2 with x' in [umin, umax - 1]
    const int xlo = max(0, umin <= 2 ? 0 : (umin - 2 + 4) / 5);
    const int xhi = umax - 3 < 0 ? -1 : min(w - 1, (umax - 3) / 5);
    if (h < 1 || w < 1 || h > RT_BM_BITS || hw >= (1ll << 31)) {        // uniform: sizes the bitmap cannot hold
        if (tid == 0) {
            m_out[n] = 0;
            if (area_out)
                area_out[n] = 0;
            if (status)
                status[n] = 2;
        }
        return;
    }
    const int cpb = (int)(RT_BM_BITS / h);                              // columns per band
    // every possible position lies in [xlo*h, (xhi+1)*h] (y == h lands on the next column's first slot)
    const long long pend = xhi < xlo ? 0ll : min((long long)(xhi + 1) * h + 1, hw);
    int idx_base = 0;                // runs written so far
    int last_pos = -1;               // last toggle position so far (-1: none)
    unsigned area = 0;

    for (long long p0 = (long long)xlo * h; p0 < pend; p0 += (long long)cpb * h) {
        const long long p1 = min(p0 + (long long)cpb * h, pend);
        const int nbits = (int)(p1 - p0), nwords = (nbits + 31) >> 5;
        for (int i = tid; i < nwords; i += RT_THREADS)
            bm[i] = 0u;
        __syncthreads();
        int e = 0;
        for (int i = tid; i < total_steps; i += RT_THREADS) {
            while (pre[e + 1] <= i)
                ++e;
            int xs = vx[e], xe = vx[e + 1], ys = vy[e], ye = vy[e + 1];
            const int dx = abs(xe - xs), dy = abs(ys - ye), len = max(dx, dy);
            const bool xmajor = dx >= dy;
            const bool back = (xmajor && xs > xe) || (!xmajor && ys > ye);
            if (back) {
                xs = xe;
                ys = ye;
            }
            const int d = i - pre[e] + 1;                       // points d - 1 and d of the edge
            const int tp = back ? len - d + 1 : d - 1, tc = back ? len - d : d;
            const double s = slope[e];
            int up, vp, uc, vc;
            if (xmajor) {                                       // x is exact: test before the doubles
                up = tp + xs;
                uc = tc + xs;
                const int xq = uc < up ? uc : uc - 1;
                if (xq < 2 || (xq - 2) % 5 != 0)
                    continue;
            }
            edge_point(xmajor, xs, ys, s, tp, up, vp);
            edge_point(xmajor, xs, ys, s, tc, uc, vc);
            if (uc == up)
                continue;
            double xd = (double)(uc < up ? uc : uc - 1);
            xd = (xd + .5) / 5.0 - .5;
            if (floor(xd) != xd || xd < 0 || xd > (double)(w - 1))
                continue;
            double yd = (double)(vc < vp ? vc : vp);
            yd = (yd + .5) / 5.0 - .5;
            if (yd < 0)
                yd = 0;
            else if (yd > (double)h)
                yd = (double)h;
            yd = ceil(yd);
            const long long p = (long long)(int)xd * h + (int)yd;
            if (p >= p0 && p < p1)                              // p == h*w is the terminator's
                atomicXor(&bm[(p - p0) >> 5], 1u << ((p - p0) & 31));
        }
        __syncthreads();
        // 3. runs off the set bits, in order
        const int wpt = (nwords + RT_THREADS - 1) / RT_THREADS;
        const int wb = min(tid * wpt, nwords), we = min(wb + wpt, nwords);
        int cnt = 0, last = -1;
        for (int i = wb; i < we; ++i) {
            const unsigned x = bm[i];
            if (x) {
                cnt += __popc(x);
                last = i * 32 + 31 - __clz(x);
            }
        }
        int isum = cnt, imax = last, tot, top;
        block_scan(isum, imax, tot, top, sc, tid);
        // previous toggle before this thread's slice = inclusive running max of the thread before
        smax[tid] = imax;
        __syncthreads();
        const int prev_excl = tid > 0 ? smax[tid - 1] : -1;
        long long prev = prev_excl >= 0 ? p0 + prev_excl : (last_pos >= 0 ? (long long)last_pos : 0ll);
        int idx = idx_base + isum - cnt;
        for (int i = wb; i < we; ++i) {
            unsigned x = bm[i];
            while (x) {
                const int bit = __ffs(x) - 1;
                x &= x - 1;
                const long long pos = p0 + i * 32 + bit;
                const unsigned run = (unsigned)(pos - prev);
                if (idx < cap)
                    out[idx] = run;
                if (idx & 1)
                    area += run;
                prev = pos;
                ++idx;
            }
        }
        idx_base += tot;
        if (top >= 0)
            last_pos = (int)(p0 + top);
        __syncthreads();
    }
    // the terminator h*w always closes the last run
    {
        int asum = (int)area, dummy = 0, tot, top;
        block_scan(asum, dummy, tot, top, sc, tid);
        if (tid == 0) {
            const unsigned run = (unsigned)(hw - (last_pos >= 0 ? last_pos : 0));
            unsigned a = (unsigned)tot;
            if (idx_base < cap)
                out[idx_base] = run;
            if (idx_base & 1)
                a += run;
            m_out[n] = idx_base + 1;
            if (area_out)
                area_out[n] = a;
            if (status)
                status[n] = idx_base + 1 > cap ? 1 : 0;
        }
    }
}

// counts -> the library's ASCII form (rleToString): 5 data bits + continuation bit per
// character, offset 48; from the fourth count on the difference to the count two back.
__global__ __launch_bounds__(RT_THREADS) void k_rle_to_string(const unsigned *__restrict__ counts,
                                                              const int *__restrict__ m, int cap,
                                                              unsigned char *__restrict__ str,
                                                              const long long *__restrict__ offsets,
                                                              long long str_cap, int *__restrict__ len_out)
{
    __shared__ int sc[8];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int mm = min(m[n], cap);
    const unsigned *c = counts + (long long)n * cap;
    unsigned char *dst = str ? str + (offsets ? offsets[n] : (long long)n * str_cap) : nullptr;
    const long long room = offsets ? offsets[n + 1] - offsets[n] : str_cap;
    int written = 0;
    for (int base = 0; base < mm; base += RT_THREADS) {
        const int i = base + tid;
        unsigned char buf[8];
        int nch = 0;
        if (i < mm) {
            long long x = (long long)c[i];
            if (i > 2)
                x -= (long long)c[i - 2];
            bool more = true;
            while (more) {
                unsigned char ch = (unsigned char)(x & 0x1f);
                x >>= 5;
                more = (ch & 0x10) ? x != -1 : x != 0;
                if (more)
                    ch |= 0x20;
                buf[nch++] = (unsigned char)(ch + 48);
            }
        }
        int incl = nch, dummy = 0, tot, top;
        block_scan(incl, dummy, tot, top, sc, tid);
        const int off = written + incl - nch;
        if (dst && off + nch <= room)
            for (int k = 0; k < nch; ++k)
                dst[off + k] = buf[k];
        written += tot;
    }
    if (tid == 0)
        len_out[n] = written;
}

// counts -> byte mask in the library's memory order (column-major: mask[x * h + y]).
// One workgroup per (polygon, slice of runs); the mask must be zero-filled by the caller's
// stream-ordered memset: only the runs of ones are written.
__global__ __launch_bounds__(RT_THREADS) void k_rle_decode(const unsigned *__restrict__ counts,
                                                           const int *__restrict__ m, int cap,
                                                           unsigned char *__restrict__ mask, long long hw)
{
    __shared__ int sc[8];
    __shared__ unsigned rstart[RT_THREADS], rlen[RT_THREADS];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mm = min(m[n], cap);
    const unsigned *c = counts + (long long)n * cap;
    unsigned char *dst = mask + (long long)n * hw;
    long long carry = 0;
    for (int base = 0; base < mm; base += RT_THREADS) {
        const int i = base + tid;
        const int len = i < mm ? (int)c[i] : 0;
        int incl = len, dummy = 0, tot, top;
        block_scan(incl, dummy, tot, top, sc, tid);
        const long long start = carry + incl - len;
        __syncthreads();
        rstart[tid] = (unsigned)min(start, hw);
        rlen[tid] = (i & 1) ? (unsigned)max(0ll, min((long long)len, hw - min(start, hw))) : 0u;
        __syncthreads();
        // a wave per run of ones, lanes along the run
        for (int r = 1 + 2 * wave; r < RT_THREADS; r += 8) {       // base is even: odd slots are ones
            const unsigned s0 = rstart[r], l = rlen[r];
            for (unsigned k = lane; k < l; k += 64)
                dst[s0 + k] = 1;
        }
        carry += tot;
    }
}

}  // namespace

extern "C" int dance_polygon_rle(const float *polys, const int32_t *sizes, int n, int v, int h, int w,
                                 uint32_t *counts, int cap, int32_t *m, uint32_t *area, int32_t *status,
                                 void *stream)
{
    DANCE_REQUIRE(n >= 0 && v >= 1 && v <= RT_MAX_V && cap >= 1);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(polys && counts && m);
    DANCE_REQUIRE(sizes || (h >= 1 && w >= 1 && (long long)h * w < (1ll << 31)));
    hipLaunchKernelGGL(k_polygon_rle, dim3(n), dim3(RT_THREADS), 0, (hipStream_t)stream, polys, sizes, v, h, w,
                       counts, cap, m, area, status);
    return launch_check(__func__);
}

extern "C" int dance_rle_to_string(const uint32_t *counts, const int32_t *m, int n, int cap, uint8_t *str,
                                   const int64_t *offsets, int64_t str_cap, int32_t *len, void *stream)
{
    DANCE_REQUIRE(n >= 0 && cap >= 1 && str_cap >= 0);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(counts && m && len);
    hipLaunchKernelGGL(k_rle_to_string, dim3(n), dim3(RT_THREADS), 0, (hipStream_t)stream, counts, m, cap, str,
                       (const long long *)offsets, (long long)str_cap, len);
    return launch_check(__func__);
}

extern "C" int dance_rle_decode(const uint32_t *counts, const int32_t *m, int n, int cap, uint8_t *mask,
                                int h, int w, void *stream)
{
    DANCE_REQUIRE(n >= 0 && cap >= 1 && h >= 1 && w >= 1 && (long long)h * w < (1ll << 31));
    if (n == 0)
        return 0;
    DANCE_REQUIRE(counts && m && mask);
    const long long hw = (long long)h * w;
    if (int rc = dance::zero_async(mask, (size_t)(hw * n), (hipStream_t)stream, __func__))
        return rc;
    hipLaunchKernelGGL(k_rle_decode, dim3(n), dim3(RT_THREADS), 0, (hipStream_t)stream, counts, m, cap, mask, hw);
    return launch_check(__func__);
}
