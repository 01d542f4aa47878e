// targets.hip -- the NumPy half of the reference's training-side target matching (SURVEY 8f
// row 4) as device kernels, ONE WORKGROUP PER INSTANCE:
//   uniform_sample                  core/modeling/edge_snake/snake_head.py:661-725
//   dense targets                   snake_head.py:1134-1144 (uniform_sample + roll + stride)
//   uniform_sample_1d               snake_head.py:728-762
//   get_aux_extreme_points          core/modeling/fcose/utils.py:94-189
//   single_uniform_segment_matching snake_head.py:1265-1384 (+ reorder_perloss 1520-1540)
// The reference runs these per instance in Python / NumPy on the host.  Index work is integer
// exact; coordinates are computed in the precision NumPy 2.2 promotes the reference's lines to
// (float32 edge lengths summed in NumPy's pairwise order, float64 from the interpolation on) with
// one rounding per operation (this file is compiled with -ffp-contract=off), so points come out
// bit-identical to the reference's on the same inputs.  The data-dependent, inherently serial
// steps (fix-up walk, clustering, cumulative sums) run on one lane; the bulk (all points, all
// distance minima) on the whole workgroup.  Tiny, latency-bound work: no roofline to speak of.
#include "common.h"

using dance::launch_check;

namespace {

constexpr int MAXP = 2048;          // polygon vertices / poly-line nodes per instance
constexpr int NT = 256;

// ------------------------------------------------------------------ workgroup helpers
// first index of the smallest value over the workgroup (np.argmin: first minimum, a NaN wins)
struct MinIdx {
    double d;
    int i;
};

__device__ __forceinline__ bool better(double d, int i, double bd, int bi)
{
    const bool dn = d != d, bn = bd != bd;
    if (dn != bn)
        return dn;
    if (dn)
        return i < bi;
    return d < bd || (d == bd && i < bi);
}

__device__ MinIdx block_argmin(double d, int i, MinIdx *red)
{
    const int t = threadIdx.x;
    red[t].d = d;
    red[t].i = i;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (t < s && better(red[t + s].d, red[t + s].i, red[t].d, red[t].i))
            red[t] = red[t + s];
        __syncthreads();
    }
    MinIdx r = red[0];
    __syncthreads();
    return r;
}

// np.sum of a contiguous 1-D array: NumPy's pairwise summation, iteratively (one lane)
template <typename T>
__device__ T pairwise_sum(const T *a, int n)
{
    struct Range {
        int lo, n;
        int state;          // 0 = not started, 1 = left done
        T left;
    };
    Range st[16];
    int sp = 0;
    st[0] = {0, n, 0, (T)0};
    T ret = (T)0;
    while (sp >= 0) {
        Range &r = st[sp];
        if (r.n <= 128) {
            const T *p = a + r.lo;
            T res;
            if (r.n < 8) {
                res = (T)0;
                for (int i = 0; i < r.n; ++i)
                    res = res + p[i];
            } else {
                T q[8];
                for (int j = 0; j < 8; ++j)
                    q[j] = p[j];
                int i = 8;
                for (; i < r.n - (r.n % 8); i += 8)
                    for (int j = 0; j < 8; ++j)
                        q[j] = q[j] + p[i + j];
                res = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
                for (; i < r.n; ++i)
                    res = res + p[i];
            }
            ret = res;
            --sp;
            // hand the value to the parent
            while (sp >= 0) {
                Range &pr = st[sp];
                if (pr.state == 0) {
                    pr.left = ret;
                    pr.state = 1;
                    int n2 = pr.n / 2;
                    n2 -= n2 % 8;
                    st[sp + 1] = {pr.lo + n2, pr.n - n2, 0, (T)0};
                    ++sp;
                    break;
                }
                ret = pr.left + ret;
                --sp;
            }
        } else {
            int n2 = r.n / 2;
            n2 -= n2 % 8;
            st[sp + 1] = {r.lo, n2, 0, (T)0};
            ++sp;
        }
    }
    return ret;
}

// ------------------------------------------------------------------ uniform_sample
struct USample {
    float *el;          // [MAXP] edge lengths
    int *rank;          // [MAXP] stable ascending rank of the edge
    int *num;           // [MAXP] points per edge, then exclusive prefix sums in start[]
    int *start;         // [MAXP + 1]
    int *sh;            // [4] scratch scalars
};

// edge lengths, stable ranks and (add branch) the per-edge counts with prefix sums.
// Returns true in the `remove points` branch (p > newp).
__device__ bool usample_setup(const float *pts, int p, int newp, USample u)
{
    const int t = threadIdx.x;
    for (int i = t; i < p; i += NT) {
        const int j = i + 1 == p ? 0 : i + 1;
        const float dx = pts[2 * j] - pts[2 * i], dy = pts[2 * j + 1] - pts[2 * i + 1];
        u.el[i] = sqrtf(dx * dx + dy * dy);
    }
    __syncthreads();
    for (int i = t; i < p; i += NT) {
        const float e = u.el[i];
        int r = 0;
        for (int j = 0; j < p; ++j) {
            const float f = u.el[j];
            r += (f < e || (f == e && j < i)) ? 1 : 0;
        }
        u.rank[i] = r;
    }
    __syncthreads();
    if (p > newp)
        return true;
    if (t == 0) {
        const float total = pairwise_sum<float>(u.el, p);
        u.sh[0] = __float_as_int(total);
    }
    __syncthreads();
    const float total = __int_as_float(u.sh[0]);
    for (int i = t; i < p; i += NT) {
        int k = (int)rintf((u.el[i] * (float)newp) / total);
        u.num[i] = k == 0 ? 1 : k;
    }
    __syncthreads();
    if (t == 0) {
        long long s = 0;
        for (int i = 0; i < p; ++i)
            s += u.num[i];
        if (s != newp) {
            // order[k] = the edge of rank k: walk from the longest down
            if (s > newp) {
                long long rest = s - newp;
                for (int k = p - 1; rest > 0 && k >= 0; --k) {
                    int e = 0;
                    for (int i = 0; i < p; ++i)
                        if (u.rank[i] == k) {
                            e = i;
                            break;
                        }
                    if (u.num[e] > rest) {
                        u.num[e] -= (int)rest;
                        rest = 0;
                    } else {
                        rest -= u.num[e] - 1;
                        u.num[e] = 1;
                    }
                }
            } else {
                for (int i = 0; i < p; ++i)
                    if (u.rank[i] == p - 1)
                        u.num[i] += (int)(newp - s);
            }
        }
        int acc = 0;
        for (int i = 0; i < p; ++i) {
            u.start[i] = acc;
            acc += u.num[i];
        }
        u.start[p] = acc;
    }
    __syncthreads();
    return false;
}

// point o of the `add points` branch
__device__ __forceinline__ void usample_point(const float *pts, int p, const USample &u, int o, double &x, double &y)
{
    int lo = 0, hi = p;                                 // last edge with start <= o
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (u.start[mid] <= o)
            lo = mid;
        else
            hi = mid;
    }
    const int e = lo, j = e + 1 == p ? 0 : e + 1;
    const double w = (double)(float)(o - u.start[e]) / (double)u.num[e];
    x = (double)pts[2 * e] * (1.0 - w) + (double)pts[2 * j] * w;
    y = (double)pts[2 * e + 1] * (1.0 - w) + (double)pts[2 * j + 1] * w;
}

__global__ __launch_bounds__(NT) void k_uniform_sample(const float *polys, const int *poly_off, const int *newpnum,
                                                       const long long *out_off, double *out, int *status)
{
    __shared__ float el[MAXP];
    __shared__ int rank[MAXP], num[MAXP], start[MAXP + 1], sh[4];
    const int n = blockIdx.x, t = threadIdx.x;
    const int p0 = poly_off[n], p = poly_off[n + 1] - p0, newp = newpnum[n];
    const float *pts = polys + 2ll * p0;
    double *o = out + 2 * out_off[n];
    if (p < 1 || p > MAXP || newp < 1) {
        if (t == 0)
            status[n] = 2;
        return;
    }
    USample u = {el, rank, num, start, sh};
    if (usample_setup(pts, p, newp, u)) {
        // keep the newp longest edges' start points, in contour order
        if (t == 0) {
            int k = 0;
            for (int i = 0; i < p; ++i)
                if (rank[i] >= p - newp) {
                    o[2 * k] = (double)pts[2 * i];
                    o[2 * k + 1] = (double)pts[2 * i + 1];
                    ++k;
                }
            status[n] = 0;
        }
        return;
    }
    for (int i = t; i < newp; i += NT) {
        double x, y;
        usample_point(pts, p, u, i, x, y);
        o[2 * i] = x;
        o[2 * i + 1] = y;
    }
    if (t == 0)
        status[n] = start[p] == newp ? 0 : 1;
}

// uniform_sample(cnt, p * m) -> roll so that the point nearest `first` leads -> every p-th point
// (snake_head.py:1134-1144): m points per instance, the p * m points never leave the chip.
__global__ __launch_bounds__(NT) void k_dense_targets(const float *polys, const int *poly_off, const float *first,
                                                      int m, double *out, int *status)
{
    __shared__ float el[MAXP];
    __shared__ int rank[MAXP], num[MAXP], start[MAXP + 1], sh[4];
    __shared__ MinIdx red[NT];
    const int n = blockIdx.x, t = threadIdx.x;
    const int p0 = poly_off[n], p = poly_off[n + 1] - p0;
    const float *pts = polys + 2ll * p0;
    double *o = out + 2ll * n * m;
    const long long total = (long long)p * m;
    if (p < 1 || p > MAXP || total > (1ll << 30)) {
        if (t == 0)
            status[n] = 2;
        return;
    }
    const int newp = (int)total;
    USample u = {el, rank, num, start, sh};
    usample_setup(pts, p, newp, u);                     // p <= p * m: always the add branch
    const float fx = first[2 * n], fy = first[2 * n + 1];
    double bd = INFINITY;
    int bi = 0x7fffffff;
    for (int i = t; i < newp; i += NT) {
        double x, y;
        usample_point(pts, p, u, i, x, y);
        const double dx = x - fx, dy = y - fy;          // float64 - float32 -> float64
        const double d = dx * dx + dy * dy;
        if (better(d, i, bd, bi)) {
            bd = d;
            bi = i;
        }
    }
    const MinIdx best = block_argmin(bd, bi, red);
    for (int k = t; k < m; k += NT) {
        int i = (int)((best.i + (long long)k * p) % newp);
        double x, y;
        usample_point(pts, p, u, i, x, y);
        o[2 * k] = x;
        o[2 * k + 1] = y;
    }
    if (t == 0)
        status[n] = start[p] == newp ? 0 : 1;
}

// ------------------------------------------------------------------ uniform_sample_1d
// nn nodes (x, y): pts[0 .. nn - 2] and, as node nn - 1, `last` when given (the arc that closes
// the contour ends at its first point) else pts[nn - 1]; out: new_n points.  start / seg: nn
// doubles of LDS.  Returns false where the reference raises.  All threads must call.
__device__ bool sample_1d(const double *pts_, int nn, const double *last, int new_n, double *out, double *start,
                          double *seg, int *flag)
{
    const int t = threadIdx.x;
    struct Nodes {
        const double *p, *last;
        int nn;
        __device__ double operator[](int k) const
        {
            const int node = k >> 1;
            return (last != nullptr && node == nn - 1) ? last[k & 1] : p[k];
        }
    } pts = {pts_, last, nn};
    if (new_n == 1) {
        if (t == 0) {
            out[0] = pts[0];
            out[1] = pts[1];
        }
        return true;
    }
    if (nn == new_n + 1) {
        for (int i = t; i < new_n; i += NT) {
            out[2 * i] = pts[2 * i];
            out[2 * i + 1] = pts[2 * i + 1];
        }
        return true;
    }
    if (nn < 2)
        return false;
    for (int i = t; i < nn - 1; i += NT) {
        const double dx = pts[2 * i + 2] - pts[2 * i], dy = pts[2 * i + 3] - pts[2 * i + 1];
        seg[i] = sqrt(dx * dx + dy * dy);
    }
    __syncthreads();
    if (t == 0) {
        double acc = 0.0;
        start[0] = acc;
        for (int i = 0; i < nn - 1; ++i) {             // np.cumsum: sequential
            acc = acc + seg[i];
            start[i + 1] = acc;
        }
        seg[nn - 1] = pairwise_sum<double>(seg, nn - 1) / (double)new_n;   // np.sum: pairwise
        *flag = 0;
    }
    __syncthreads();
    const double per = seg[nn - 1];
    for (int i = t; i < new_n; i += NT) {
        if (i == 0) {
            out[0] = pts[0];
            out[1] = pts[1];
            continue;
        }
        const double mark = (double)i * per;
        // first node with start - mark > 0 (start is non-decreasing)
        int lo = 0, hi = nn;                            // start[lo] - mark <= 0 < start[hi] - mark
        if (!(start[nn - 1] - mark > 0)) {
            *flag = 1;
            continue;
        }
        hi = nn - 1;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (start[mid] - mark > 0)
                hi = mid;
            else
                lo = mid;
        }
        const int a = hi, b = hi - 1;
        const double before = start[b] - mark, after = start[a] - mark;
        const double w = -before / (after - before);
        out[2 * i] = (1.0 - w) * pts[2 * b] + w * pts[2 * a];
        out[2 * i + 1] = (1.0 - w) * pts[2 * b + 1] + w * pts[2 * a + 1];
    }
    __syncthreads();
    return *flag == 0;
}

__global__ __launch_bounds__(NT) void k_uniform_sample_1d(const double *pts, const int *off, const int *new_n,
                                                          const long long *out_off, double *out, int *status)
{
    __shared__ double start[MAXP], seg[MAXP];
    __shared__ int flag;
    const int n = blockIdx.x;
    const int nn = off[n + 1] - off[n], m = new_n[n];
    if (nn < 1 || nn > MAXP || m < 1) {
        if (threadIdx.x == 0)
            status[n] = 2;
        return;
    }
    const bool ok = sample_1d(pts + 2ll * off[n], nn, nullptr, m, out + 2 * out_off[n], start, seg, &flag);
    if (threadIdx.x == 0)
        status[n] = ok ? 0 : 1;
}

// ------------------------------------------------------------------ get_aux_extreme_points
__global__ __launch_bounds__(NT) void k_aux_extreme_points(const double *pts_all, double *re_all, long long *ext_all,
                                                           int *count, int *status, int np_, int cap)
{
    constexpr int MAXAUX = 64;
    __shared__ unsigned char band[MAXP];
    __shared__ double aux[2 * MAXAUX];
    __shared__ int ext[MAXAUX];
    __shared__ int naux, over;
    __shared__ double box[4];
    __shared__ MinIdx red[NT];
    const int n = blockIdx.x, t = threadIdx.x;
    const double *pts = pts_all + 2ll * n * np_;
    double *re = re_all + 2ll * n * np_;
    // extent (python min / max over the columns)
    double lo_x = INFINITY, lo_y = INFINITY, hi_x = -INFINITY, hi_y = -INFINITY;
    for (int i = t; i < np_; i += NT) {
        lo_x = fmin(lo_x, pts[2 * i]);
        hi_x = fmax(hi_x, pts[2 * i]);
        lo_y = fmin(lo_y, pts[2 * i + 1]);
        hi_y = fmax(hi_y, pts[2 * i + 1]);
    }
    {
        MinIdx r = block_argmin(lo_x, t, red);
        if (t == 0) box[0] = r.d;
        r = block_argmin(lo_y, t, red);
        if (t == 0) box[1] = r.d;
        r = block_argmin(-hi_x, t, red);
        if (t == 0) box[2] = -r.d;
        r = block_argmin(-hi_y, t, red);
        if (t == 0) box[3] = -r.d;
    }
    if (t == 0) {
        naux = 0;
        over = 0;
    }
    __syncthreads();
    const double l = box[0], tp = box[1], r_ = box[2], b = box[3];
    const double thresh = 0.02, w = r_ - l + 1, h = b - tp + 1;
    // the four sides in the reference's order: top, bottom, left, right
    for (int side = 0; side < 4; ++side) {
        const int axis = side < 2 ? 1 : 0, other = 1 - axis;
        const bool lo_side = (side & 1) == 0;
        const double edge = side == 0 ? tp : (side == 1 ? b : (side == 2 ? l : r_));
        const double lim = thresh * (axis == 1 ? h : w);
        for (int i = t; i < np_; i += NT) {
            const double v = pts[2 * i + axis];
            band[i] = (lo_side ? (v - edge) : (edge - v)) <= lim ? 1 : 0;
        }
        __syncthreads();
        if (t == 0) {
            for (;;) {
                // the band's extreme point, first in index order
                int k = -1;
                for (int i = 0; i < np_; ++i)
                    if (band[i] && (k < 0 || (lo_side ? pts[2 * i + axis] < pts[2 * k + axis]
                                                      : pts[2 * i + axis] > pts[2 * k + axis])))
                        k = i;
                if (k < 0)
                    break;
                double mx = pts[2 * k + other], mn = mx;
                band[k] = 0;
                for (int step = 1; step >= -1; step -= 2) {
                    int tmp = k + step;
                    tmp = tmp < 0 ? tmp + np_ : (tmp >= np_ ? tmp - np_ : tmp);
                    while (tmp != k && (lo_side ? (pts[2 * tmp + axis] - pts[2 * k + axis])
                                                : (pts[2 * k + axis] - pts[2 * tmp + axis])) <= lim) {
                        mx = fmax(mx, pts[2 * tmp + other]);
                        mn = fmin(mn, pts[2 * tmp + other]);
                        band[tmp] = 0;
                        tmp += step;
                        tmp = tmp < 0 ? tmp + np_ : (tmp >= np_ ? tmp - np_ : tmp);
                    }
                }
                const double mid = (mx + mn) / 2;
                if (naux < MAXAUX) {
                    aux[2 * naux + axis] = edge;
                    aux[2 * naux + other] = mid;
                    ++naux;
                } else {
                    over = 1;
                }
            }
        }
        __syncthreads();
    }
    // roll so that the point nearest the first auxiliary point leads
    const int na = naux;
    double bd = INFINITY;
    int bi = 0x7fffffff;
    for (int i = t; i < np_; i += NT) {
        const double dx = pts[2 * i] - aux[0], dy = pts[2 * i + 1] - aux[1];
        const double d = dx * dx + dy * dy;
        if (better(d, i, bd, bi)) {
            bd = d;
            bi = i;
        }
    }
    const int shift = block_argmin(bd, bi, red).i;
    for (int i = t; i < np_; i += NT) {
        int s = i + shift;
        s = s >= np_ ? s - np_ : s;
        re[2 * i] = pts[2 * s];
        re[2 * i + 1] = pts[2 * s + 1];
    }
    for (int k = 0; k < na; ++k) {
        bd = INFINITY;
        bi = 0x7fffffff;
        for (int i = t; i < np_; i += NT) {
            int s = i + shift;
            s = s >= np_ ? s - np_ : s;
            const double dx = aux[2 * k] - pts[2 * s], dy = aux[2 * k + 1] - pts[2 * s + 1];
            const double d = dx * dx + dy * dy;
            if (better(d, i, bd, bi)) {
                bd = d;
                bi = i;
            }
        }
        const int e = block_argmin(bd, bi, red).i;
        if (t == 0)
            ext[k] = k == 0 ? 0 : e;
    }
    __syncthreads();
    if (t == 0) {
        // np.unique: sorted, duplicates dropped
        for (int i = 1; i < na; ++i) {
            const int v = ext[i];
            int j = i - 1;
            while (j >= 0 && ext[j] > v) {
                ext[j + 1] = ext[j];
                --j;
            }
            ext[j + 1] = v;
        }
        int c = 0;
        for (int i = 0; i < na; ++i)
            if (i == 0 || ext[i] != ext[i - 1]) {
                if (c < cap)
                    ext_all[(long long)n * cap + c] = ext[i];
                ++c;
            }
        count[n] = c < cap ? c : cap;
        status[n] = (over || c > cap) ? 1 : 0;
    }
}

// the reference's fall-back of both matchings: the cyclic shift of dense[::up] nearest the contour
// (summed squared distance, first minimum), snake_head.py:1378-1382 / 1492-1496, 1520-1540
__device__ void fallback_reorder(const double *dense, const float *sampled, double *out, int v, int up, MinIdx *red)
{
    const int t = threadIdx.x;
    double bd = INFINITY;
    int bi = 0x7fffffff;
    for (int s = t; s < v; s += NT) {
        double tot = 0.0;
        for (int i = 0; i < v; ++i) {
            int k = s + i;
            k = k >= v ? k - v : k;
            const double dx = dense[2ll * k * up] - (double)sampled[2 * i];
            const double dy = dense[2ll * k * up + 1] - (double)sampled[2 * i + 1];
            tot = tot + (dx * dx + dy * dy);
        }
        if (better(tot, s, bd, bi)) {
            bd = tot;
            bi = s;
        }
    }
    const int sft = block_argmin(bd, bi, red).i;
    for (int i = t; i < v; i += NT) {
        int k = sft + i;
        k = k >= v ? k - v : k;
        out[2 * i] = dense[2ll * k * up];
        out[2 * i + 1] = dense[2ll * k * up + 1];
    }
}

// ------------------------------------------------------------------ segment matching
__global__ __launch_bounds__(NT) void k_segment_matching(const double *dense_all, const float *sampled_all,
                                                         const long long *edge_idx_all, double *out_all, int *status,
                                                         int v, int up, int e_num)
{
    __shared__ double start[MAXP], seg[MAXP];
    __shared__ MinIdx red[NT];
    __shared__ int mn[5], aext[5], flag, okflag;
    const int n = blockIdx.x, t = threadIdx.x;
    const int big = v * up;
    const double *dense = dense_all + 2ll * n * big;
    const float *sampled = sampled_all + 2ll * n * v;
    const long long *edge_idx = edge_idx_all + (long long)n * e_num;
    double *out = out_all + 2ll * n * v;
    // nearest dense point of each of the four extreme vertices (every third edge start)
    for (int k = 0; k < 4; ++k) {
        const int e = (int)edge_idx[3 * k];
        const double cx = (double)sampled[2 * e], cy = (double)sampled[2 * e + 1];
        double bd = INFINITY;
        int bi = 0x7fffffff;
        for (int i = t; i < big; i += NT) {
            const double dx = cx - dense[2 * i], dy = cy - dense[2 * i + 1];
            const double d = dx * dx + dy * dy;
            if (better(d, i, bd, bi)) {
                bd = d;
                bi = i;
            }
        }
        const int m = block_argmin(bd, bi, red).i;
        if (t == 0) {
            mn[k] = m;
            aext[k] = e;
        }
    }
    if (t == 0) {
        mn[4] = big;
        aext[4] = v;
        if (mn[3] < mn[1])
            mn[3] = big - 2;
        if (mn[0] != 0)
            mn[0] = 0;
        for (int a = 1; a < 5; ++a)                     // np.sort of five values, both lists
            for (int c = a; c > 0; --c) {
                if (mn[c - 1] > mn[c]) {
                    const int x = mn[c - 1];
                    mn[c - 1] = mn[c];
                    mn[c] = x;
                }
                if (aext[c - 1] > aext[c]) {
                    const int x = aext[c - 1];
                    aext[c - 1] = aext[c];
                    aext[c] = x;
                }
            }
        if (mn[2] == big - 1) {
            mn[2] = big - 3;
            mn[3] = big - 2;
        }
        if (mn[3] == big - 1) {
            mn[3] -= 1;
            mn[2] -= 1;
        }
        okflag = 1;
    }
    __syncthreads();
    int written = 0;
    for (int i = 0; i < 4; ++i) {
        const int cnt = aext[i + 1] - aext[i];
        if (cnt == 0)
            continue;
        if (t == 0 && mn[i + 1] - mn[i] <= 0)
            mn[i + 1] = mn[i] + 1;
        __syncthreads();
        const int a0 = mn[i];
        int len = i == 3 ? big - a0 + 1 : mn[i + 1] + 1 - a0;
        if (i != 3 && a0 + len > big)
            len = big - a0;                             // numpy slicing clips at the array's end
        if (len < 0)
            len = 0;
        bool ok = len >= 1 && written + cnt <= v && len <= MAXP;
        if (ok)     // arcs are slices of the dense contour; the last one ends at its first point
            ok = sample_1d(dense + 2ll * a0, len, i == 3 ? dense : nullptr, cnt, out + 2 * written, start, seg, &flag);
        if (!ok) {
            if (t == 0)
                okflag = 0;
            break;                                      // uniform for the workgroup
        }
        written += cnt;
        __syncthreads();
    }
    __syncthreads();
    if (okflag && written == v) {
        if (t == 0)
            status[n] = 0;
        return;
    }
    fallback_reorder(dense, sampled, out, v, up, red);
    if (t == 0)
        status[n] = 1;
}

// RefineNet.single_uniform_multisegment_matching (snake_head.py:1386-1498), the form the shipped
// configuration (INITIAL = box, NEW_MATCHING) runs after get_aux_extreme_points, together with the
// roll of the initial contour that precedes it (1150-1155): the contour is rolled to start at the
// vertex nearest dense[0]; the auxiliary extreme points (indices into dense, sorted, first = 0)
// are matched to their nearest contour vertices; every arc between consecutive ones is resampled
// to the contour's vertex count on that arc.  k (ext_count) >= 2.
__global__ __launch_bounds__(NT) void k_multisegment_matching(const double *dense_all, const float *sampled_all,
                                                              const long long *ext_all, const int *ext_count,
                                                              float *rolled_all, double *out_all, int *status,
                                                              int v, int up, int cap)
{
    constexpr int MAXK = 65;
    __shared__ double start[MAXP], seg[MAXP];
    __shared__ MinIdx red[NT];
    __shared__ float roll[2 * MAXP];
    __shared__ int mn[MAXK + 1], aext[MAXK + 1], flag, okflag;
    const int n = blockIdx.x, t = threadIdx.x;
    const int big = v * up, k = ext_count[n];
    const double *dense = dense_all + 2ll * n * big;
    const float *sampled = sampled_all + 2ll * n * v;
    const long long *ext = ext_all + (long long)n * cap;
    double *out = out_all + 2ll * n * v;
    float *rolled = rolled_all + 2ll * n * v;
    if (k < 2 || k > MAXK - 1 || k > cap) {
        if (t == 0)
            status[n] = 2;
        return;
    }
    {   // roll the contour: vertex nearest dense[0] first (float32 - float64 -> float64)
        double bd = INFINITY;
        int bi = 0x7fffffff;
        for (int i = t; i < v; i += NT) {
            const double dx = (double)sampled[2 * i] - dense[0], dy = (double)sampled[2 * i + 1] - dense[1];
            const double d = dx * dx + dy * dy;
            if (better(d, i, bd, bi)) {
                bd = d;
                bi = i;
            }
        }
        const int sft = block_argmin(bd, bi, red).i;
        for (int i = t; i < v; i += NT) {
            int s = i + sft;
            s = s >= v ? s - v : s;
            roll[2 * i] = sampled[2 * s];
            roll[2 * i + 1] = sampled[2 * s + 1];
            rolled[2 * i] = sampled[2 * s];
            rolled[2 * i + 1] = sampled[2 * s + 1];
        }
        __syncthreads();
    }
    for (int j = 0; j < k; ++j) {
        const int m = (int)ext[j];
        const double cx = dense[2 * m], cy = dense[2 * m + 1];
        double bd = INFINITY;
        int bi = 0x7fffffff;
        for (int i = t; i < v; i += NT) {
            const double dx = cx - (double)roll[2 * i], dy = cy - (double)roll[2 * i + 1];
            const double d = dx * dx + dy * dy;
            if (better(d, i, bd, bi)) {
                bd = d;
                bi = i;
            }
        }
        const int e = block_argmin(bd, bi, red).i;
        if (t == 0) {
            aext[j] = e;
            mn[j] = m;
        }
    }
    if (t == 0) {
        if (aext[0] != 0)
            aext[0] = 0;
        if (aext[k - 1] < aext[1])
            aext[k - 1] = v - 2;
        // np.sort(ext_idx), then the bound appended and both lists sorted again (1406-1434)
        for (int a = 1; a < k; ++a)
            for (int c = a; c > 0 && aext[c - 1] > aext[c]; --c) {
                const int x = aext[c - 1];
                aext[c - 1] = aext[c];
                aext[c] = x;
            }
        aext[k] = v;
        mn[k] = big;
        if (mn[k] < mn[1])
            mn[k] = big - 2;
        if (mn[0] != 0)
            mn[0] = 0;
        for (int a = 1; a <= k; ++a) {
            for (int c = a; c > 0 && aext[c - 1] > aext[c]; --c) {
                const int x = aext[c - 1];
                aext[c - 1] = aext[c];
                aext[c] = x;
            }
            for (int c = a; c > 0 && mn[c - 1] > mn[c]; --c) {
                const int x = mn[c - 1];
                mn[c - 1] = mn[c];
                mn[c] = x;
            }
        }
        if (mn[k - 1] == big - 1) {
            mn[k - 1] = big - 3;
            mn[k] = big - 2;
        }
        if (mn[k] == big - 1) {
            mn[k] -= 1;
            mn[k - 1] -= 1;
        }
        okflag = 1;
    }
    __syncthreads();
    int written = 0;
    for (int i = 0; i < k; ++i) {
        const int cnt = aext[i + 1] - aext[i];
        if (cnt == 0)
            continue;
        if (t == 0 && mn[i + 1] - mn[i] <= 0)
            mn[i + 1] = mn[i] + 1;
        __syncthreads();
        const int a0 = mn[i];
        const bool closing = i == k - 1;
        int len = closing ? big - a0 + 1 : mn[i + 1] + 1 - a0;
        if (!closing && a0 + len > big)
            len = big - a0;
        if (len < 0)
            len = 0;
        bool ok = len >= 1 && written + cnt <= v && len <= MAXP;
        if (ok)
            ok = sample_1d(dense + 2ll * a0, len, closing ? dense : nullptr, cnt, out + 2 * written, start, seg, &flag);
        if (!ok) {
            if (t == 0)
                okflag = 0;
            break;
        }
        written += cnt;
        __syncthreads();
    }
    __syncthreads();
    if (okflag && written == v) {
        if (t == 0)
            status[n] = 0;
        return;
    }
    fallback_reorder(dense, roll, out, v, up, red);
    if (t == 0)
        status[n] = 1;
}

}  // namespace

extern "C" int dance_uniform_sample(const float *polys, const int32_t *poly_off, const int32_t *newpnum,
                                    const int64_t *out_off, double *out, int32_t *status, int n, void *stream)
{
    DANCE_REQUIRE(n >= 0);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(polys && poly_off && newpnum && out_off && out && status);
    hipLaunchKernelGGL(k_uniform_sample, dim3(n), dim3(NT), 0, (hipStream_t)stream, polys, poly_off, newpnum,
                       (const long long *)out_off, out, status);
    return launch_check(__func__);
}

extern "C" int dance_dense_targets(const float *polys, const int32_t *poly_off, const float *first, int m,
                                   double *out, int32_t *status, int n, void *stream)
{
    DANCE_REQUIRE(n >= 0 && m >= 1);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(polys && poly_off && first && out && status);
    hipLaunchKernelGGL(k_dense_targets, dim3(n), dim3(NT), 0, (hipStream_t)stream, polys, poly_off, first, m, out,
                       status);
    return launch_check(__func__);
}

extern "C" int dance_uniform_sample_1d(const double *pts, const int32_t *off, const int32_t *new_n,
                                       const int64_t *out_off, double *out, int32_t *status, int n, void *stream)
{
    DANCE_REQUIRE(n >= 0);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(pts && off && new_n && out_off && out && status);
    hipLaunchKernelGGL(k_uniform_sample_1d, dim3(n), dim3(NT), 0, (hipStream_t)stream, pts, off, new_n,
                       (const long long *)out_off, out, status);
    return launch_check(__func__);
}

extern "C" int dance_aux_extreme_points(const double *pts, double *reordered, int64_t *ext_idx, int32_t *count,
                                        int32_t *status, int n, int num_points, int cap, void *stream)
{
    DANCE_REQUIRE(n >= 0 && num_points >= 1 && num_points <= MAXP && cap >= 1);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(pts && reordered && ext_idx && count && status);
    hipLaunchKernelGGL(k_aux_extreme_points, dim3(n), dim3(NT), 0, (hipStream_t)stream, pts, reordered,
                       (long long *)ext_idx, count, status, num_points, cap);
    return launch_check(__func__);
}

extern "C" int dance_segment_matching(const double *dense, const float *sampled, const int64_t *edge_idx,
                                      double *out, int32_t *status, int n, int v, int up_rate, int e_num,
                                      void *stream)
{
    DANCE_REQUIRE(n >= 0 && v >= 4 && up_rate >= 1 && (long long)v * up_rate + 1 <= MAXP && e_num >= 10);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(dense && sampled && edge_idx && out && status);
    hipLaunchKernelGGL(k_segment_matching, dim3(n), dim3(NT), 0, (hipStream_t)stream, dense, sampled,
                       (const long long *)edge_idx, out, status, v, up_rate, e_num);
    return launch_check(__func__);
}

extern "C" int dance_multisegment_matching(const double *dense, const float *sampled, const int64_t *ext_idx,
                                           const int32_t *ext_count, float *rolled, double *out, int32_t *status,
                                           int n, int v, int up_rate, int cap, void *stream)
{
    DANCE_REQUIRE(n >= 0 && v >= 4 && up_rate >= 1 && (long long)v * up_rate + 1 <= MAXP && cap >= 2);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(dense && sampled && ext_idx && ext_count && rolled && out && status);
    hipLaunchKernelGGL(k_multisegment_matching, dim3(n), dim3(NT), 0, (hipStream_t)stream, dense, sampled,
                       (const long long *)ext_idx, ext_count, rolled, out, status, v, up_rate, cap);
    return launch_check(__func__);
}
