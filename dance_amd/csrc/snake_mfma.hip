// snake_mfma.hip -- the snake blocks of DANCE as fp32 MFMA implicit GEMMs for gfx950.
//
// One workgroup (4 wavefronts, one per SIMD) owns one contour and 128 output channels.
// The circular dilated Conv1d
//     y[o][v] = b[o] + sum_t sum_c W[o][c][t] * x[c][(v + (t - taps/2) d) mod V]
// is the GEMM  Y(o x v) = W(o x K) * X(K x v)  with K = (tap, channel):
//   * A operand = weights, read straight from L2 in the packed layout [tap][c][o]
//     (lane -> 32 consecutive o: two 128-byte segments per wave-load, prefetched one
//     16-channel block ahead);
//   * B operand = the contour's activations, staged once per channel chunk in LDS as
//     [c][V]; the ring shift of a tap is an index rotation, so the 9 taps re-read the
//     same LDS image and no halo copy or im2col buffer exists anywhere;
//   * v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation (an fmaf chain), which
//     is what keeps vertex coordinates within 1e-4 of the fp32 reference; each wave holds
//     one 32-channel row tile for ALL vertex tiles of the ring (RT x 16 accumulators).
//   * C/D tile: lane -> vertex (col), registers -> channels (rows), so every global store
//     is two 128-byte row segments of the channel-major (C, V) activation layout.
// The epilogue fuses bias, ReLU, the eval-BatchNorm affine, the residual add, the
// per-contour row constant of prediction[0] and the max over the ring of the fusion layer.
#include <stdlib.h>

#include "common.h"
#include "p16.h"
#include "sampling.h"

using dance::launch_check;


namespace {

struct ConvArgs {
    const float *x;
    long long x_stride;
    const float *w, *bias, *scale, *shift, *residual, *rowinit;
    float *y;
    long long y_stride;
    int v, cin, cin_pad, cout, taps, dilation, flags, ck;
};

struct GatherArgs {
    const float *feat, *att, *loc;
    const int *image_idx;
    float *att_out;
    int b, hf, wf, c, reflection, deloc;
};

// ring index of vertex (vv + off); vv may be a padded lane (>= v), off in [-16, 16]
__device__ __forceinline__ int ring(int vv, int off, int v)
{
    int t = vv + off;
    t = t < 0 ? t + v : t;
    t = t >= v ? t - v : t;
    t = t >= v ? t - v : t;
    return t;
}

// Accumulate one LDS-resident channel chunk: acc[rt] += W[:, chunk] * X[chunk, :].
//   xs      : LDS image [ckn][v] of the chunk
//   wchunk  : &w[(c0 + hi) * cout + o0 + lo]   (tap 0 of this chunk, this lane's row pair)
template <int RT>
__device__ __forceinline__ void mfma_chunk(f32x16 (&acc)[RT], const float *xs, int v, int ckn,
                                           const float *__restrict__ wchunk, long long tap_stride,
                                           int cout, int taps, int dilation, int lo, int hi)
{
    const int ncb = ckn >> 4;                    // blocks of 16 channels = 8 k-pairs
    const int steps = taps * ncb;
    const long long row2 = 2ll * cout;
    float a_nxt[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
        a_nxt[u] = wchunk[u * row2];
    int tap = 0, cb = 0;
    int idx[RT];
    for (int step = 0; step < steps; ++step) {
        float a_cur[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            a_cur[u] = a_nxt[u];
        int ntap = tap, nblk = cb + 1;
        if (nblk == ncb) {
            nblk = 0;
            ++ntap;
        }
        if (step + 1 < steps) {
            const float *p = wchunk + ntap * tap_stride + (long long)nblk * 16 * cout;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                a_nxt[u] = p[u * row2];
        }
        if (cb == 0) {
            const int off = (tap - (taps >> 1)) * dilation;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                idx[rt] = ring(rt * 32 + lo, off, v) + hi * v;
        }
        const float *xb = xs + cb * 16 * v;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float bval = xb[2 * u * v + idx[rt]];
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[u], bval, acc[rt], 0, 0, 0);
            }
        }
        tap = ntap;
        cb = nblk;
    }
}

// accumulator start: bias[o] or the per-contour row constant
template <int RT>
__device__ __forceinline__ void init_acc(f32x16 (&acc)[RT], const float *row, int o0, int hi)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float b = row[o0 + (r & 3) + 8 * (r >> 2) + 4 * hi];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            acc[rt][r] = b;
    }
}

template <int RT>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[RT], const ConvArgs &a, long long nn, int o0,
                                         int lo, int hi)
{
    const int v = a.v;
    const bool relu = a.flags & DANCE_EP_RELU, affine = a.flags & DANCE_EP_AFFINE;
    const bool resid = a.flags & DANCE_EP_RESIDUAL, maxv = a.flags & DANCE_EP_MAXV;
    const bool p16 = a.flags & DANCE_EP_OUT_P16;
    // register group g holds channels o0 + 8g + 4*hi + (0..3) for the lane's vertex column
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float sc[4], sh[4], m[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = o0 + 8 * g + 4 * hi + i;
            sc[i] = affine ? a.scale[o] : 1.0f;
            sh[i] = affine ? a.shift[o] : 0.0f;
            m[i] = -INFINITY;
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int vv = rt * 32 + lo;
            if (vv >= v)
                continue;
            float val[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = o0 + 8 * g + 4 * hi + i;
                float x = acc[rt][4 * g + i];
                if (relu)
                    x = x > 0.0f ? x : (x != x ? x : 0.0f);
                if (affine)
                    x = x * sc[i] + sh[i];
                if (resid)
                    x += a.residual[nn * a.y_stride + (long long)o * v + vv];
                val[i] = x;
                m[i] = dance::nan_max(x, m[i]);
            }
            if (p16) {
                // y is a P16 block (p16.h): contour stride y_stride halfs, 128-channel slots
                _Float16 *slot = reinterpret_cast<_Float16 *>(a.y) + nn * a.y_stride +
                                 (long long)(o0 >> 7) * 256 * v;
                dance::store_p16(slot, v, ((o0 & 127) >> 3) + g, vv, hi, val);
            } else if (!maxv) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a.y[nn * a.y_stride + (long long)(o0 + 8 * g + 4 * hi + i) * v + vv] = val[i];
            }
        }
        if (maxv) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float mm = m[i];
#pragma unroll
                for (int s = 16; s >= 1; s >>= 1)
                    mm = dance::nan_max(__shfl_xor(mm, s, 64), mm);
                if (lo == 0)
                    a.y[nn * a.cout + o0 + 8 * g + 4 * hi + i] = mm;
            }
        }
    }
}

template <int RT>
__global__ __launch_bounds__(256, 2) void k_circconv(ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const long long nn = blockIdx.x;
    const int o0 = blockIdx.y * 128 + wave * 32;
    const bool active = o0 < a.cout;
    const int v = a.v;
    const float *xn = a.x + nn * a.x_stride;

    f32x16 acc[RT];
    if (active)
        init_acc<RT>(acc, (a.flags & DANCE_EP_ROWINIT) ? a.rowinit + nn * a.cout : a.bias, o0, hi);

    for (int c0 = 0; c0 < a.cin_pad; c0 += a.ck) {
        const int ckn = min(a.ck, a.cin_pad - c0);
        const int real = max(0, min(ckn, a.cin - c0)) * v;   // floats that exist in global
        const int total = ckn * v;
        __syncthreads();
        const float4 *src = reinterpret_cast<const float4 *>(xn + (long long)c0 * v);
        float4 *dst = reinterpret_cast<float4 *>(xs);
        for (int i = t; i < (total >> 2); i += 256)
            dst[i] = (4 * i < real) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        if (active)
            mfma_chunk<RT>(acc, xs, v, ckn, a.w + (long long)(c0 + hi) * a.cout + o0 + lo,
                           (long long)a.cin_pad * a.cout, a.cout, a.taps, a.dilation, lo, hi);
    }
    if (active)
        epilogue<RT>(acc, a, nn, o0, lo, hi);
}

// Fused gather + de_location + head SnakeBlock.  Chunks of CK feature channels are
// gathered (4 taps x 16 B per lane, CK/4 lanes per vertex) and blended directly into the
// LDS image the MFMA loop reads; the last chunk holds the two de_location channels.
template <int RT>
__global__ __launch_bounds__(256, 2) void k_gather_head(ConvArgs a, GatherArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int v = a.v, ck = a.ck;
    float *xs = smem;                                          // [ck][v]
    float *tw = smem + (size_t)ck * v;                         // [v][4]
    int *toff = reinterpret_cast<int *>(tw + 4 * v);           // [v][4]
    float *dl = reinterpret_cast<float *>(toff + 4 * v);       // [2][v] de_location
    float *red = dl + 2 * v;                                   // [16]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const long long nn = blockIdx.x;
    const int o0 = wave * 32;
    const float *loc_n = g.loc + nn * v * 2;
    int img = g.image_idx ? g.image_idx[nn] : g.b - 1;
    const bool img_ok = img >= 0 && img < g.b;
    img = img_ok ? img : 0;
    const long long plane = (long long)g.hf * g.wf;

    float xmin = 0.f, xmax = 0.f, ymin = 0.f, ymax = 0.f;
    if (g.deloc != DANCE_DELOC_NONE)
        dance::contour_bounds(loc_n, v, red, xmin, xmax, ymin, ymax);

    for (int vi = t; vi < v; vi += 256) {
        const float lx = loc_n[2 * vi], ly = loc_n[2 * vi + 1];
        dance::Taps tp = dance::bilinear_taps(lx, ly, g.hf, g.wf, g.reflection);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tw[vi * 4 + k] = img_ok ? tp.w[k] : 0.0f;
            toff[vi * 4 + k] = tp.off[k];
        }
        if (g.att_out != nullptr) {
            float s = 0.0f;
            if (g.att != nullptr) {
#pragma clang fp contract(off)
                const float *ap = g.att + img * plane;
                s = ap[tp.off[0]] * tp.w[0] + ap[tp.off[1]] * tp.w[1] + ap[tp.off[2]] * tp.w[2] +
                    ap[tp.off[3]] * tp.w[3];
                s = img_ok ? s : 0.0f;
            }
            g.att_out[nn * v + vi] = s;
        }
        if (g.deloc != DANCE_DELOC_NONE) {
            dl[vi] = dance::deloc_value(lx, xmin, xmax, g.deloc);
            dl[v + vi] = dance::deloc_value(ly, ymin, ymax, g.deloc);
        }
    }

    f32x16 acc[RT];
    init_acc<RT>(acc, a.bias, o0, hi);

    const float *fimg = g.feat + img * plane * g.c;
    const int lpv = ck >> 2;                 // lanes per vertex (ck/4 floats4 per tap row)
    const int vpi = 256 / lpv;               // vertices gathered per workgroup pass
    const int lc = t % lpv, vs = t / lpv;
    // chunks of ck feature channels, then one 16-row chunk holding the two de_location
    // channels (rows 2..15 are zero and meet zero weight rows)
    const int c_end = g.c + (g.deloc != DANCE_DELOC_NONE ? 16 : 0);
    for (int c0 = 0; c0 < c_end; c0 += ck) {
        __syncthreads();                     // taps visible / previous chunk consumed
        int ckn = ck;
        if (c0 < g.c) {
#pragma unroll 2
            for (int s = vs; s < v; s += vpi) {
#pragma clang fp contract(off)
                const float w0 = tw[s * 4 + 0], w1 = tw[s * 4 + 1], w2 = tw[s * 4 + 2], w3 = tw[s * 4 + 3];
                const int ch = c0 + 4 * lc;
                const float4 a0 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 0] * g.c + ch);
                const float4 a1 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 1] * g.c + ch);
                const float4 a2 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 2] * g.c + ch);
                const float4 a3 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 3] * g.c + ch);
                float *d = xs + (4 * lc) * v + s;
                d[0] = a0.x * w0 + a1.x * w1 + a2.x * w2 + a3.x * w3;
                d[v] = a0.y * w0 + a1.y * w1 + a2.y * w2 + a3.y * w3;
                d[2 * v] = a0.z * w0 + a1.z * w1 + a2.z * w2 + a3.z * w3;
                d[3 * v] = a0.w * w0 + a1.w * w1 + a2.w * w2 + a3.w * w3;
            }
        } else {
            ckn = 16;
            for (int i = t; i < 16 * v; i += 256)
                xs[i] = i < 2 * v ? dl[i] : 0.0f;
        }
        __syncthreads();
        mfma_chunk<RT>(acc, xs, v, ckn, a.w + (long long)(c0 + hi) * a.cout + o0 + lo,
                       (long long)a.cin_pad * a.cout, a.cout, a.taps, a.dilation, lo, hi);
    }
    epilogue<RT>(acc, a, nn, o0, lo, hi);
}

// h[n][o] = b[o] + sum_g wg[g][o] * gmax[n][g]: the vertex-invariant part of
// prediction[0] (the 256 global-state input columns, snake_head.py:278-282), hoisted.
__global__ __launch_bounds__(256) void k_rowinit(const float *__restrict__ gmax,
                                                 const float *__restrict__ wg,
                                                 const float *__restrict__ b, float *__restrict__ h,
                                                 int g_dim, int cout)
{
    __shared__ float gs[1024];
    const long long nn = blockIdx.x;
    for (int i = threadIdx.x; i < g_dim; i += blockDim.x)
        gs[i] = gmax[nn * g_dim + i];
    __syncthreads();
    for (int o = threadIdx.x; o < cout; o += blockDim.x) {
        float s = b[o];
        for (int k = 0; k < g_dim; ++k)
            s = fmaf(wg[(long long)k * cout + o], gs[k], s);
        h[nn * cout + o] = s;
    }
}

// prediction[4] (64 -> 2) + tanh * wh + attention + residual add + clip (one lane per vertex)
__global__ __launch_bounds__(256) void k_offset_update(const float *__restrict__ p2,
                                                       long long p2_stride,
                                                       const float *__restrict__ w3,
                                                       const float *__restrict__ b3,
                                                       const float *__restrict__ loc,
                                                       const float *__restrict__ whs,
                                                       const float *__restrict__ att,
                                                       float *__restrict__ out, long long total,
                                                       int v, int cmid, int clip_first, float clip_x,
                                                       float clip_y)
{
    const long long t = blockIdx.x * 256ll + threadIdx.x;
    if (t >= total)
        return;
    const long long nn = t / v;
    const int vi = (int)(t - nn * v);
    const float *p = p2 + nn * p2_stride + vi;
    float z0 = b3[0], z1 = b3[1];
    for (int k = 0; k < cmid; ++k) {
        const float x = p[(long long)k * v];
        z0 = fmaf(w3[2 * k], x, z0);
        z1 = fmaf(w3[2 * k + 1], x, z1);
    }
    if (whs != nullptr) {
        z0 = tanhf(z0) * whs[nn * 2];
        z1 = tanhf(z1) * whs[nn * 2 + 1];
    }
    if (att != nullptr) {
        const float s = att[t];
        z0 *= s;
        z1 *= s;
    }
    float x = loc[t * 2] + z0, y = loc[t * 2 + 1] + z1;
    if (clip_first && nn == 0) {
        x = x < 0.0f ? 0.0f : (x > clip_x ? clip_x : x);    // NaN stays NaN, like clamp_
        y = y < 0.0f ? 0.0f : (y > clip_y ? clip_y : y);
    }
    out[t * 2] = x;
    out[t * 2 + 1] = y;
}

int pick_rt(int v) { return v <= 64 ? 2 : (v <= 128 ? 4 : (v <= 224 ? 7 : 8)); }

// channels per LDS chunk: as many as fit 64 KiB (two workgroups per CU), multiple of 16
int pick_ck(int v, int cin_pad)
{
    int ck = (16384 / v) & ~15;
    ck = ck > 256 ? 256 : ck;
    return ck < cin_pad ? ck : cin_pad;
}

template <int RT>
int launch_head(dim3 grid, size_t lds, hipStream_t s, const ConvArgs &a, const GatherArgs &g)
{
    if (int rc = dance::allow_lds((const void *)k_gather_head<RT>, lds, "dance_gather_circconv_head"))
        return rc;
    hipLaunchKernelGGL(k_gather_head<RT>, grid, dim3(256), lds, s, a, g);
    return launch_check("dance_gather_circconv_head");
}

}  // namespace

extern "C" int dance_circconv(const float *x, int64_t x_stride, const float *w_packed,
                              const float *bias, const float *scale, const float *shift,
                              const float *residual, const float *rowinit, float *y,
                              int64_t y_stride, int n, int v, int cin, int cout, int taps,
                              int dilation, int flags, void *stream)
{
    DANCE_REQUIRE(n >= 0 && v >= 16 && v <= 256 && v % 4 == 0);
    DANCE_REQUIRE(cin >= 1 && cout >= 32 && cout % 32 == 0);
    DANCE_REQUIRE(taps == 1 || taps == 9);
    DANCE_REQUIRE(dilation >= 1 && dilation * (taps / 2) <= 16);
    DANCE_REQUIRE(x_stride >= (int64_t)cin * v && x_stride % 4 == 0);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(x && w_packed && y);
    DANCE_REQUIRE((flags & DANCE_EP_ROWINIT) ? rowinit != nullptr : bias != nullptr);
    DANCE_REQUIRE(!(flags & DANCE_EP_AFFINE) || (scale && shift));
    DANCE_REQUIRE(!(flags & DANCE_EP_RESIDUAL) || residual);
    DANCE_REQUIRE((flags & (DANCE_EP_MAXV | DANCE_EP_OUT_P16)) || y_stride >= (int64_t)cout * v);
    DANCE_REQUIRE(!(flags & DANCE_EP_OUT_P16) || !(flags & (DANCE_EP_MAXV | DANCE_EP_RESIDUAL)));
    ConvArgs a;
    a.x = x; a.x_stride = x_stride; a.w = w_packed; a.bias = bias; a.scale = scale;
    a.shift = shift; a.residual = residual; a.rowinit = rowinit; a.y = y; a.y_stride = y_stride;
    a.v = v; a.cin = cin; a.cin_pad = (cin + 15) & ~15; a.cout = cout; a.taps = taps;
    a.dilation = dilation; a.flags = flags; a.ck = pick_ck(v, a.cin_pad);
    const size_t lds = (size_t)a.ck * v * sizeof(float);
    dim3 grid(n, dance::ceil_div(cout, 128));
    hipStream_t s = (hipStream_t)stream;
    switch (pick_rt(v)) {
    case 2: hipLaunchKernelGGL(k_circconv<2>, grid, dim3(256), lds, s, a); break;
    case 4: hipLaunchKernelGGL(k_circconv<4>, grid, dim3(256), lds, s, a); break;
    case 7: hipLaunchKernelGGL(k_circconv<7>, grid, dim3(256), lds, s, a); break;
    default: hipLaunchKernelGGL(k_circconv<8>, grid, dim3(256), lds, s, a); break;
    }
    return launch_check(__func__);
}

extern "C" int dance_gather_circconv_head(const float *feat, const float *att, const float *loc,
                                          const int32_t *image_idx, const float *w_packed,
                                          const float *bias, const float *scale, const float *shift,
                                          float *y, int64_t y_stride, float *att_out, int n, int v,
                                          int b, int hf, int wf, int c, int cout, int padding_mode,
                                          int deloc_type, void *stream)
{
    DANCE_REQUIRE(n >= 0 && v >= 16 && v <= 256 && v % 4 == 0);
    DANCE_REQUIRE(b >= 1 && hf >= 1 && wf >= 1 && (long long)hf * wf < (1ll << 30));
    DANCE_REQUIRE(cout == 128);
    DANCE_REQUIRE(padding_mode == DANCE_PAD_ZEROS || padding_mode == DANCE_PAD_REFLECTION);
    DANCE_REQUIRE(deloc_type >= DANCE_DELOC_DERANGE && deloc_type <= DANCE_DELOC_NONE);
    DANCE_REQUIRE(y_stride >= (int64_t)cout * v);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(feat && loc && w_packed && bias && scale && shift && y);
    ConvArgs a;
    a.x = nullptr; a.x_stride = 0; a.w = w_packed; a.bias = bias; a.scale = scale; a.shift = shift;
    a.residual = nullptr; a.rowinit = nullptr; a.y = y; a.y_stride = y_stride; a.v = v;
    a.cin = c + (deloc_type != DANCE_DELOC_NONE ? 2 : 0);
    a.cin_pad = (a.cin + 15) & ~15; a.cout = cout; a.taps = 9; a.dilation = 1;
    a.flags = DANCE_EP_RELU | DANCE_EP_AFFINE;
    // chunk: 128 channels while chunk + tap tables stay under 80 KiB (2 workgroups/CU), else 64
    a.ck = (v <= 144 && c % 128 == 0) ? 128 : 64;
    DANCE_REQUIRE(c % a.ck == 0);
    GatherArgs g;
    g.feat = feat; g.att = att; g.loc = loc; g.image_idx = image_idx; g.att_out = att_out;
    g.b = b; g.hf = hf; g.wf = wf; g.c = c; g.reflection = padding_mode == DANCE_PAD_REFLECTION;
    g.deloc = deloc_type;
    const size_t lds = ((size_t)a.ck * v + 4 * v + 4 * v + 2 * v + 16) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(n);
    switch (pick_rt(v)) {
    case 2: return launch_head<2>(grid, lds, s, a, g);
    case 4: return launch_head<4>(grid, lds, s, a, g);
    case 7: return launch_head<7>(grid, lds, s, a, g);
    default: return launch_head<8>(grid, lds, s, a, g);
    }
}

extern "C" int dance_row_constant(const float *g, const float *wg, const float *bias, float *h, int n,
                                  int g_dim, int cout, void *stream)
{
    DANCE_REQUIRE(n >= 0 && g_dim >= 1 && g_dim <= 1024 && cout >= 1);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(g && wg && bias && h);
    hipLaunchKernelGGL(k_rowinit, dim3(n), dim3(256), 0, (hipStream_t)stream, g, wg, bias, h, g_dim,
                       cout);
    return launch_check(__func__);
}

extern "C" int dance_offset_update(const float *p2, int64_t p2_stride, const float *w3,
                                   const float *b3, const float *loc, const float *whs,
                                   const float *att, float *out, int n, int v, int cmid,
                                   int clip_first, float clip_x_max, float clip_y_max, void *stream)
{
    DANCE_REQUIRE(n >= 0 && v >= 1 && cmid >= 1);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(p2 && w3 && b3 && loc && out);
    const long long total = (long long)n * v;
    hipLaunchKernelGGL(k_offset_update, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, p2, (long long)p2_stride, w3, b3, loc, whs, att, out,
                       total, v, cmid, clip_first, clip_x_max, clip_y_max);
    return launch_check(__func__);
}

// ------------------------------------------------------------------------------------
// One evolve step = the kernel sequence below on one stream; every intermediate lives in
// the caller's workspace.
// ------------------------------------------------------------------------------------
namespace {
struct Workspace {
    float *x0, *state, *gmax, *h, *p1, *p2, *att_s;
    long long bytes;
};

Workspace carve(void *base, int n, int v, int cin0)
{
    Workspace w;
    auto al = [](long long x) { return (x + 63) & ~63ll; };   // 256-byte aligned pieces
    float *p = (float *)base;
    long long off = 0;
    w.x0 = p + off;    off += al((long long)n * cin0 * v);
    w.state = p + off; off += al((long long)n * 1024 * v);
    w.gmax = p + off;  off += al((long long)n * 256);
    w.h = p + off;     off += al((long long)n * 256);
    w.p1 = p + off;    off += al((long long)n * 256 * v);
    w.p2 = p + off;    off += al((long long)n * 64 * v);
    w.att_s = p + off; off += al((long long)n * v);
    w.bytes = off * (long long)sizeof(float);
    return w;
}
}  // namespace

extern "C" int64_t dance_snake_workspace_bytes(int n, int v)
{
    if (n <= 0 || v <= 0)
        return 0;
    return carve(nullptr, n, v, 272).bytes;
}

#define DANCE_TRY(call)      \
    do {                     \
        int rc_ = (call);    \
        if (rc_)             \
            return rc_;      \
    } while (0)

extern "C" int dance_snake_evolve(const dance_snake_net *net, const float *feat, const float *att,
                                  const float *loc, const int32_t *image_idx, const float *whs,
                                  const float *x_in, float *out, void *workspace,
                                  int64_t workspace_bytes, int n, int v, int b, int hf, int wf,
                                  int padding_mode, int deloc_type, int individual_scale,
                                  int use_att, int clip_first, float clip_x_max, float clip_y_max,
                                  int fused_head, void **stage_events_host, void *stream)
{
    DANCE_REQUIRE(net != nullptr && n >= 0);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(workspace != nullptr && out != nullptr && loc != nullptr);
    DANCE_REQUIRE(v >= 16 && v <= 256 && v % 4 == 0);
    const int c = net->feat_channels;
    DANCE_REQUIRE(c >= 4 && c + 2 <= 272);
    DANCE_REQUIRE(!individual_scale || whs != nullptr);
    DANCE_REQUIRE(!use_att || att != nullptr);
    DANCE_REQUIRE(x_in != nullptr || feat != nullptr);
    const int cin0 = c + 2;
    Workspace w = carve(workspace, n, v, 272);
    DANCE_REQUIRE(workspace_bytes >= w.bytes);
    const long long sstride = 1024ll * v;
    const int bn = DANCE_EP_RELU | DANCE_EP_AFFINE;
    int stage = 0;
    // optional instrumentation: an event before the first kernel and after every stage
    auto mark = [&]() -> int {
        if (stage_events_host == nullptr)
            return 0;
        void *ev = stage_events_host[stage++];
        if (ev == nullptr)                       // sparse instrumentation: this boundary is not timed
            return 0;
        hipError_t e = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
        return e == hipSuccess ? 0 : dance::fail((int)e, "dance_snake_evolve", hipGetErrorString(e));
    };
    DANCE_TRY(mark());

    if (x_in != nullptr) {
        DANCE_TRY(dance_circconv(x_in, (long long)cin0 * v, net->head_w, net->head_b,
                                 net->head_scale, net->head_shift, nullptr, nullptr, w.state,
                                 sstride, n, v, cin0, 128, 9, 1, bn, stream));
    } else if (fused_head) {
        DANCE_TRY(dance_gather_circconv_head(feat, att, loc, image_idx, net->head_w, net->head_b,
                                             net->head_scale, net->head_shift, w.state, sstride,
                                             use_att ? w.att_s : nullptr, n, v, b, hf, wf, c, 128,
                                             padding_mode, deloc_type, stream));
    } else {
        DANCE_TRY(dance_gather_features(feat, att, loc, image_idx, w.x0,
                                        use_att ? w.att_s : nullptr, n, v, b, hf, wf, c,
                                        padding_mode, deloc_type, stream));
        DANCE_TRY(dance_circconv(w.x0, (long long)cin0 * v, net->head_w, net->head_b,
                                 net->head_scale, net->head_shift, nullptr, nullptr, w.state,
                                 sstride, n, v, cin0, 128, 9, 1, bn, stream));
    }
    DANCE_TRY(mark());
    for (int i = 0; i < 7; ++i) {
        const float *xi = w.state + (long long)i * 128 * v;
        float *yi = w.state + (long long)(i + 1) * 128 * v;
        DANCE_TRY(dance_circconv(xi, sstride, net->res_w[i], net->res_b[i], net->res_scale[i],
                                 net->res_shift[i], xi, nullptr, yi, sstride, n, v, 128, 128, 9,
                                 net->dilation[i], bn | DANCE_EP_RESIDUAL, stream));
        DANCE_TRY(mark());
    }
    DANCE_TRY(dance_circconv(w.state, sstride, net->fusion_w, net->fusion_b, nullptr, nullptr,
                             nullptr, nullptr, w.gmax, 0, n, v, 1024, 256, 1, 1, DANCE_EP_MAXV,
                             stream));
    DANCE_TRY(mark());
    DANCE_TRY(dance_row_constant(w.gmax, net->pred0_wg, net->pred0_b, w.h, n, 256, 256, stream));
    DANCE_TRY(mark());
    DANCE_TRY(dance_circconv(w.state, sstride, net->pred0_ws, nullptr, nullptr, nullptr, nullptr,
                             w.h, w.p1, 256ll * v, n, v, 1024, 256, 1, 1,
                             DANCE_EP_ROWINIT | DANCE_EP_RELU, stream));
    DANCE_TRY(mark());
    DANCE_TRY(dance_circconv(w.p1, 256ll * v, net->pred1_w, net->pred1_b, nullptr, nullptr, nullptr,
                             nullptr, w.p2, 64ll * v, n, v, 256, 64, 1, 1, DANCE_EP_RELU, stream));
    DANCE_TRY(mark());
    DANCE_TRY(dance_offset_update(w.p2, 64ll * v, net->pred2_w, net->pred2_b, loc,
                                  individual_scale ? whs : nullptr, use_att ? w.att_s : nullptr, out,
                                  n, v, 64, clip_first, clip_x_max, clip_y_max, stream));
    DANCE_TRY(mark());
    return 0;
}

namespace {

// stage-boundary events of one evolve call (bench instrumentation; null: nothing recorded)
struct StageMarks {
    void **events;
    void *stream;
    const char *fn;
    int stage = 0;
    int operator()()
    {
        if (events == nullptr)
            return 0;
        void *ev = events[stage++];
        if (ev == nullptr)                       // sparse instrumentation: this boundary is not timed
            return 0;
        hipError_t e = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
        return e == hipSuccess ? 0 : dance::fail((int)e, fn, hipGetErrorString(e));
    }
};

// The fast mode after the head block (state slot 0 written): residual stack, fusion, tail.
int evolve16_after_head(const dance_snake_net16 *net, const Workspace &w, const float *loc,
                        const float *whs, const float *att_s, float *out, int n, int v, int clip_first,
                        float clip_x_max, float clip_y_max, int mfma_terms, const dance_policy *policy,
                        StageMarks &mark, void *stream)
{
    _Float16 *state = reinterpret_cast<_Float16 *>(w.state);     // (n, 8 slots) P16
    _Float16 *p1 = reinterpret_cast<_Float16 *>(w.p1);           // (n, 2 slots) P16
    const long long slot = 256ll * v, sstride = 8 * slot, pstride = 2 * slot;
    // one launch for the 7 residual blocks; its time is booked on stage res0
    // (the exact-fp32 path's gathered-input piece of the workspace is free here: the pair flags of the
    // few-contour stack live in its first bytes)
    DANCE_TRY(dance_res_stack16(net, state, sstride, slot, n, v, mfma_terms, policy, reinterpret_cast<int32_t *>(w.x0),
                                stream));
    for (int i = 0; i < 7; ++i)
        DANCE_TRY(mark());
    // the fusion kernel (any ring length, rings above 128 vertices as two parts) + k_tail16 (parts' max,
    // row constant, ReLU, prediction[2], prediction[4], offset update)
    const int parts = dance_fuse16_parts(n, v, policy);
    int tail_parts = parts;
    float *gpart = w.gmax;                  // (n, parts <= 2, 256): gmax and h are adjacent pieces
    DANCE_REQUIRE(w.h == w.gmax + (long long)n * 256);
    // one part per contour and a last round of workgroups that is mostly empty: its contours as
    // half rings (fuse16w.hip, k_fuse16w_split); the tail kernel then reads two part slots
    const int n_split = parts == 1 ? dance_fuse16w_tail_split(n, v, policy) : 0;
    if (n_split > 0) {
        DANCE_TRY(dance_fuse16w_raw_split(state, sstride, slot, net->fuse_w, net->fuse_bs, net->fuse_winv,
                                          net->pred0_winv, p1, pstride, slot, gpart, n, v, n_split, mfma_terms,
                                          stream));
        tail_parts = 2;
    } else {
        DANCE_TRY(dance_fuse16w_raw(state, sstride, slot, net->fuse_w, net->fuse_bs, net->fuse_winv,
                                    net->pred0_winv, p1, pstride, slot, gpart, n, v, parts, mfma_terms, stream));
    }
    DANCE_TRY(mark());
    DANCE_TRY(mark());      // row constant: inside the tail kernel
    DANCE_TRY(mark());      // prediction[0]'s ReLU: inside the tail kernel
    DANCE_TRY(mark());      // prediction[2]: the tail kernel, booked on the last stage
    DANCE_TRY(dance_tail16(p1, pstride, slot, gpart, tail_parts, net->pred0_wg, net->pred0_b, net->pred1_w,
                           net->pred1_bs, net->pred1_winv, net->pred2_w, net->pred2_b, loc, whs, att_s, out, n, v,
                           clip_first, clip_x_max, clip_y_max, mfma_terms, policy, stream));
    DANCE_TRY(mark());
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------
// One evolve step in the fast precision mode (P16 activations, 3 x f16 MFMA): same stages,
// fusion + row constant + prediction[0] fused into one kernel that reads the state once.
// ------------------------------------------------------------------------------------
extern "C" int dance_snake_evolve16(const dance_snake_net16 *net, const float *feat, const float *att,
                                    const uint32_t *feat_absmax, const float *loc, const int32_t *image_idx,
                                    const float *whs, float *out, void *workspace, int64_t workspace_bytes,
                                    int n, int v, int b, int hf, int wf, int padding_mode, int deloc_type,
                                    int individual_scale, int use_att, int clip_first, float clip_x_max,
                                    float clip_y_max, int mfma_terms, const dance_policy *policy,
                                    void **stage_events_host, void *stream)
{
    DANCE_REQUIRE(net != nullptr && n >= 0);
    DANCE_REQUIRE(mfma_terms == 3 || mfma_terms == 1);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(workspace != nullptr && out != nullptr && loc != nullptr && feat != nullptr);
    DANCE_REQUIRE(v >= 16 && v <= 256 && v % 4 == 0);
    const int c = net->feat_channels;
    DANCE_REQUIRE(c >= 64 && c % 64 == 0 && c + 2 <= 272);
    DANCE_REQUIRE(!individual_scale || whs != nullptr);
    DANCE_REQUIRE(!use_att || att != nullptr);
    DANCE_REQUIRE(net->head_wd != nullptr || deloc_type == DANCE_DELOC_NONE);
    Workspace w = carve(workspace, n, v, 272);
    DANCE_REQUIRE(workspace_bytes >= w.bytes);
    _Float16 *state = reinterpret_cast<_Float16 *>(w.state);     // (n, 8 slots) P16
    const long long slot = 256ll * v, sstride = 8 * slot;
    StageMarks mark{stage_events_host, stream, "dance_snake_evolve16"};
    DANCE_TRY(mark());
    DANCE_TRY(dance_gather_head16w(feat, att, loc, image_idx, net->head_w, net->head_wd, net->head_bs,
                                   net->head_winv, net->head_wd_winv, net->head_scale, net->head_shift, state, sstride,
                                   use_att ? w.att_s : nullptr, n, v, b, hf, wf, c, padding_mode, deloc_type,
                                   mfma_terms, feat_absmax, policy, stream));
    DANCE_TRY(mark());
    return evolve16_after_head(net, w, loc, individual_scale ? whs : nullptr, use_att ? w.att_s : nullptr, out,
                               n, v, clip_first, clip_x_max, clip_y_max, mfma_terms, policy, mark, stream);
}

// ------------------------------------------------------------------------------------
// Dsnake's init stage in the fast precision mode (DsnakeHead.init, dsnake_head.py:718-744):
// gather + centre feature + fuse Conv1d + de_location + the init snake's head block in one
// launch (init16.hip), then the same residual stack / fusion / tail as an evolve step.
// No attention, no individual scale, no clipping (the caller clamps the extreme points).
// ------------------------------------------------------------------------------------
extern "C" int dance_snake_init16(const dance_snake_net16 *net, const dance_init_fuse16 *fuse,
                                  const float *feat, const float *loc, const int32_t *image_idx, float *out,
                                  void *workspace, int64_t workspace_bytes, int n, int v, int b, int hf,
                                  int wf, int padding_mode, int deloc_type, int mfma_terms,
                                  const uint32_t *feat_absmax, const dance_policy *policy,
                                  void **stage_events_host, void *stream)
{
    DANCE_REQUIRE(net != nullptr && fuse != nullptr && n >= 0);
    DANCE_REQUIRE(mfma_terms == 3 || mfma_terms == 1);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(workspace != nullptr && out != nullptr && loc != nullptr && feat != nullptr);
    DANCE_REQUIRE(v >= 16 && v <= 64 && v % 4 == 0);
    DANCE_REQUIRE(net->feat_channels == 256 && net->head_wd != nullptr);
    Workspace w = carve(workspace, n, v, 272);
    DANCE_REQUIRE(workspace_bytes >= w.bytes);
    StageMarks mark{stage_events_host, stream, "dance_snake_init16"};
    DANCE_TRY(mark());
    DANCE_TRY(dance_init_head16(feat, loc, image_idx, fuse->w16, fuse->w_inv_scale, fuse->wg_center, fuse->bias,
                                net->head_w, net->head_wd, net->head_bs, net->head_winv, net->head_wd_winv, net->head_scale,
                                net->head_shift, w.state, 2048ll * v, n, v, b, hf, wf, 256, padding_mode, deloc_type, mfma_terms,
                                feat_absmax, stream));
    DANCE_TRY(mark());
    return evolve16_after_head(net, w, loc, nullptr, nullptr, out, n, v, 0, 0.0f, 0.0f, mfma_terms, policy,
                               mark, stream);
}
