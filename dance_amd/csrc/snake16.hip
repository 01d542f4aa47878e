// snake16.hip -- the snake blocks in the fast precision mode: split-half (P16) activations and
// weights, three f16 MFMAs per product term (see p16.h), fp32 accumulation.
//
// Same decomposition as snake_mfma.hip (one workgroup = one contour, a wave = 32-channel row
// tiles x all vertex tiles, weights from L2 as the A operand, the contour's LDS image as the
// B operand, ring taps as index rotations), with v_mfma_f32_32x32x16_f16:
//   A lane (r = lane & 31, h = lane >> 5): W[o0 + r][channels 8h .. 8h+7]  = one 16-B word of
//     the packed weights [plane][tap][c/8][o][8];
//   B lane: X[channels 8h .. 8h+7][vertex r] = one 16-B word of the P16 LDS image
//     [plane][c/8][v][8]  ->  ds_read_b128, consecutive lanes on consecutive words.
// Kernels: k_conv16 (residual SnakeBlock / k=1 Conv1d), k_fuse16 (fusion + max over the ring
// + row constant + prediction[0], the 1024-channel state read once), k_gather_head16 (gather
// + de_location + head block).
#include "common.h"
#include "p16.h"
#include "sampling.h"

using dance::launch_check;

namespace {

struct Conv16Args {
    const _Float16 *x;          // P16 input
    long long x_stride;         // halfs per contour
    long long x_slot_stride;    // halfs per 128-channel slot
    const h8 *w;                // [plane][tap][cin/8][cout] words
    const float *bias_s;        // bias * 2^s (accumulator start)
    const float *scale, *shift; // folded BatchNorm (or null)
    float winv;                 // 2^-s
    _Float16 *y;                // P16 output (or null)
    long long y_stride, y_slot_stride;
    float *yf;                  // fp32 channel-major output (or null)
    long long yf_stride;
    const float *rowinit;       // per-contour accumulator start (n, cout), unscaled, or null
    int v, cin, cout, taps, dilation, ck, relu, residual, maxv;
};

__device__ __forceinline__ int ring16(int vv, int off, int v)
{
    int t = vv + off;
    t = t < 0 ? t + v : t;
    t = t >= v ? t - v : t;
    t = t >= v ? t - v : t;
    return t;
}

// acc[ot][rt] += W[o-tiles, chunk] * X[chunk, :] for one LDS-resident chunk of `cbc` channel
// blocks.  xs: LDS image [2][cbc][vp] words (16 B);  wl: this lane's weight word for tap 0,
// block (chunk start + h), o-tile 0;  wps / wts / wbs: plane / tap / channel-block strides and
// wos the o-tile stride of the packed weights, in words.
template <int RT, int OTW>
__device__ __forceinline__ void mfma16_chunk(f32x16 (&acc)[OTW][RT], const char *xs, int v, int vp,
                                             int cbc, const h8 *__restrict__ wl, long long wps,
                                             long long wts, long long wbs, int wos, int taps,
                                             int dilation, int r, int h)
{
    const int nst = cbc >> 1;                 // steps of 16 channels
    const int steps = taps * nst;
    const int lds_plane = cbc * vp * 16;      // bytes
    h8 ah_n[OTW], al_n[OTW];
#pragma unroll
    for (int ot = 0; ot < OTW; ++ot) {
        ah_n[ot] = wl[ot * wos];
        al_n[ot] = wl[wps + ot * wos];
    }
    int tap = 0, s = 0;
    int idx[RT];
    for (int step = 0; step < steps; ++step) {
        h8 ah[OTW], al[OTW];
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot) {
            ah[ot] = ah_n[ot];
            al[ot] = al_n[ot];
        }
        int ntap = tap, ns = s + 1;
        if (ns == nst) {
            ns = 0;
            ++ntap;
        }
        if (step + 1 < steps) {
            const h8 *p = wl + ntap * wts + ns * 2 * wbs;
#pragma unroll
            for (int ot = 0; ot < OTW; ++ot) {
                ah_n[ot] = p[ot * wos];
                al_n[ot] = p[wps + ot * wos];
            }
        }
        if (s == 0) {
            const int off = (tap - (taps >> 1)) * dilation;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                idx[rt] = (h * vp + ring16(rt * 32 + r, off, v)) * 16;
        }
        const char *xb = xs + s * 2 * vp * 16;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const h8 bh = *reinterpret_cast<const h8 *>(xb + idx[rt]);
            const h8 bl = *reinterpret_cast<const h8 *>(xb + idx[rt] + lds_plane);
#pragma unroll
            for (int ot = 0; ot < OTW; ++ot) {
                acc[ot][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ot], bh, acc[ot][rt], 0, 0, 0);
                acc[ot][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ot], bl, acc[ot][rt], 0, 0, 0);
                acc[ot][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ot], bh, acc[ot][rt], 0, 0, 0);
            }
        }
        tap = ntap;
        s = ns;
    }
}

// copy `cbc` channel blocks (both planes) of a P16 slot into the LDS image [2][cbc][vp];
// one row (plane, block) = v words, rows dealt to waves, lanes along the row.
__device__ __forceinline__ void load_chunk_p16(char *xs, const _Float16 *slot, int v, int vp, int cb0,
                                               int cbc, int wave, int nwaves, int lane)
{
    for (int row = wave; row < 2 * cbc; row += nwaves) {
        const int plane = row >= cbc ? 1 : 0;
        const int cbl = row - plane * cbc;
        const uint4 *src = reinterpret_cast<const uint4 *>(slot + plane * 128ll * v +
                                                            (long long)(cb0 + cbl) * v * 8);
        uint4 *dst = reinterpret_cast<uint4 *>(xs + (long long)row * vp * 16);
        for (int i = lane; i < v; i += 64)
            dst[i] = src[i];
    }
}

template <int RT, int OTW>
__device__ __forceinline__ void init_acc16(f32x16 (&acc)[OTW][RT], const float *bias_s, int o0, int h)
{
#pragma unroll
    for (int ot = 0; ot < OTW; ++ot)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float b = bias_s ? bias_s[o0 + ot * 32 + (q & 3) + 8 * (q >> 2) + 4 * h] : 0.0f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                acc[ot][rt][q] = b;
        }
}

// Residual SnakeBlock / k=1 Conv1d.  4 waves, one 32-channel tile each (128 outputs per
// workgroup, grid.y tiles of 128).
template <int RT>
__global__ __launch_bounds__(256, 2) void k_conv16(Conv16Args a)
{
    extern __shared__ __attribute__((aligned(16))) char xs[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long nn = blockIdx.x;
    const int o0 = blockIdx.y * 128 + wave * 32;
    const bool active = o0 < a.cout;
    const int v = a.v, vp = v + 1;
    const _Float16 *xn = a.x + nn * a.x_stride;
    const int cbt = a.cin >> 3;
    const long long wbs = a.cout, wts = (long long)cbt * a.cout, wps = (long long)a.taps * wts;

    f32x16 acc[1][RT];
    if (active) {
        init_acc16<RT, 1>(acc, a.rowinit ? a.rowinit + nn * a.cout : a.bias_s, o0, h);
        if (a.rowinit) {
            const float ws = 1.0f / a.winv;               // exact: a power of two
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                acc[0][rt] *= ws;
        }
    }

    for (int c0 = 0; c0 < a.cin; c0 += a.ck) {
        const int ckn = min(a.ck, a.cin - c0);
        const int cbc = ckn >> 3;
        __syncthreads();
        load_chunk_p16(xs, xn + (c0 >> 7) * a.x_slot_stride, v, vp, (c0 & 127) >> 3, cbc, wave, 4, lane);
        __syncthreads();
        if (active)
            mfma16_chunk<RT, 1>(acc, xs, v, vp, cbc, a.w + ((long long)(c0 >> 3) + h) * wbs + o0 + r,
                                wps, wts, wbs, 0, a.taps, a.dilation, r, h);
    }
    if (!active)
        return;
    // epilogue: lane = vertex column, register group g = channels 8g + 4h .. + 3 of the tile
    const int cbc_last = min(a.ck, a.cin) >> 3;
    _Float16 *yslot = a.y ? a.y + nn * a.y_stride + (o0 >> 7) * a.y_slot_stride : nullptr;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float sc[4], sh[4], mx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = o0 + 8 * g + 4 * h + i;
            sc[i] = a.scale ? a.scale[o] : 1.0f;
            sh[i] = a.scale ? a.shift[o] : 0.0f;
            mx[i] = -INFINITY;
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int vv = rt * 32 + r;
            if (vv >= v)
                continue;
            float val[4];
            h4 rh, rl;
            if (a.residual) {
                // x itself (single-chunk image still in LDS): rows (o0/8 + g), halfs 4h..4h+3
                const char *p = xs + ((long long)((o0 >> 3) + g) * vp + vv) * 16 + 8 * h;
                rh = *reinterpret_cast<const h4 *>(p);
                rl = *reinterpret_cast<const h4 *>(p + (long long)cbc_last * vp * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = acc[0][rt][4 * g + i] * a.winv;
                if (a.relu)
                    x = x > 0.0f ? x : (x != x ? x : 0.0f);
                if (a.scale)
                    x = x * sc[i] + sh[i];
                if (a.residual)
                    x += (float)rh[i] + (float)rl[i];
                val[i] = x;
                mx[i] = dance::nan_max(x, mx[i]);
            }
            if (a.maxv)
                continue;
            if (yslot) {
                dance::store_p16(yslot, v, ((o0 & 127) >> 3) + g, vv, h, val);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a.yf[nn * a.yf_stride + (long long)(o0 + 8 * g + 4 * h + i) * v + vv] = val[i];
            }
        }
        if (a.maxv) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float m = mx[i];
#pragma unroll
                for (int s = 16; s >= 1; s >>= 1)
                    m = dance::nan_max(__shfl_xor(m, s, 64), m);
                if (r == 0)
                    a.yf[nn * a.cout + o0 + 8 * g + 4 * h + i] = m;
            }
        }
    }
}

// fusion (1024 -> 256, max over the ring) + row constant + prediction[0] (1280 -> 256, ReLU)
// in one pass over the 1024-channel state: 8 waves, two 32-channel tiles each; waves 0-3 own
// the fusion outputs, waves 4-7 the prediction outputs (weights packed as one 512-output GEMM).
struct Fuse16Args {
    const _Float16 *x;
    long long x_stride, x_slot_stride;
    const h8 *w;                  // [plane][1][128][512]
    const float *bias_s;          // (512): fusion bias * 2^s, then zeros
    const float *wg;              // (256, 256) [g][o] fp32: global-state columns of prediction[0]
    const float *pb;              // (256) prediction[0] bias
    float winv;
    _Float16 *y;                  // P16, 2 slots
    long long y_stride, y_slot_stride;
    int v, ck;
};

template <int RT>
__global__ __launch_bounds__(512, 2) void k_fuse16(Fuse16Args a)
{
    extern __shared__ __attribute__((aligned(16))) char xs[];
    __shared__ float gs[256];
    __shared__ float hp[512];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long nn = blockIdx.x;
    const int o0 = wave * 64;
    const int v = a.v, vp = v + 1;
    const _Float16 *xn = a.x + nn * a.x_stride;
    const long long wbs = 512, wts = 128ll * 512, wps = wts;

    f32x16 acc[2][RT];
    init_acc16<RT, 2>(acc, a.bias_s, o0, h);
    for (int c0 = 0; c0 < 1024; c0 += a.ck) {
        const int cbc = a.ck >> 3;
        __syncthreads();
        load_chunk_p16(xs, xn + (c0 >> 7) * a.x_slot_stride, v, vp, (c0 & 127) >> 3, cbc, wave, 8, lane);
        __syncthreads();
        mfma16_chunk<RT, 2>(acc, xs, v, vp, cbc, a.w + ((long long)(c0 >> 3) + h) * wbs + o0 + r, wps,
                            wts, wbs, 32, 1, 1, r, h);
    }
    if (wave < 4) {
        // global state = max over the ring of the fusion output (snake_head.py:271-272)
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float m = -INFINITY;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    if (rt * 32 + r < v)
                        m = dance::nan_max(acc[ot][rt][q] * a.winv, m);
#pragma unroll
                for (int s = 16; s >= 1; s >>= 1)
                    m = dance::nan_max(__shfl_xor(m, s, 64), m);
                if (r == 0)
                    gs[o0 + ot * 32 + (q & 3) + 8 * (q >> 2) + 4 * h] = m;
            }
    }
    __syncthreads();
    {   // row constant h[o] = pb[o] + sum_g wg[g][o] * G[g], two half sums per output
        const int o = t & 255, half = t >> 8;
        float s = 0.0f;
        const float *wgp = a.wg + (long long)half * 128 * 256 + o;
#pragma unroll 8
        for (int k = 0; k < 128; ++k)
            s = fmaf(wgp[(long long)k * 256], gs[half * 128 + k], s);
        hp[t] = s;
    }
    __syncthreads();
    if (wave >= 4) {
        const int p0 = o0 - 256;                        // prediction output channel of tile 0
        _Float16 *yn = a.y + nn * a.y_stride;
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float hc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = p0 + ot * 32 + 8 * g + 4 * h + i;
                    hc[i] = a.pb[o] + hp[o] + hp[256 + o];
                }
                const int ob = p0 + ot * 32 + 8 * g;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int vv = rt * 32 + r;
                    if (vv >= v)
                        continue;
                    float val[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float x = acc[ot][rt][4 * g + i] * a.winv + hc[i];
                        val[i] = x > 0.0f ? x : (x != x ? x : 0.0f);
                    }
                    dance::store_p16(yn + (ob >> 7) * a.y_slot_stride, v, (ob & 127) >> 3, vv, h, val);
                }
            }
    }
}

// Fused gather + de_location + head SnakeBlock, P16 flavour: chunks of 64 feature channels
// are gathered (4 taps x 16 B per lane, 16 lanes per vertex), blended in fp32, split into
// hi/lo halves and written straight into the LDS image the MFMA loop reads.
struct Gather16Args {
    const float *feat, *att, *loc;
    const int *image_idx;
    float *att_out;
    int b, hf, wf, c, reflection, deloc;
};

template <int RT>
__global__ __launch_bounds__(256, 2) void k_gather_head16(Conv16Args a, Gather16Args g)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int v = a.v, vp = v + 1;
    constexpr int CBC = 8;                                      // 64 channels per chunk
    char *xs = smem;                                            // [2][8][vp] words
    float *tw = reinterpret_cast<float *>(smem + 2 * CBC * vp * 16);   // [v][4]
    int *toff = reinterpret_cast<int *>(tw + 4 * v);            // [v][4]
    float *dl = reinterpret_cast<float *>(toff + 4 * v);        // [2][v]
    float *red = dl + 2 * v;                                    // [16]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
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

    f32x16 acc[1][RT];
    init_acc16<RT, 1>(acc, a.bias_s, o0, h);
    const int cbt = a.cin >> 3;                                 // padded input channels / 8
    const long long wbs = a.cout, wts = (long long)cbt * a.cout, wps = (long long)a.taps * wts;
    const float *fimg = g.feat + img * plane * g.c;
    const int lc = t & 15, vs = t >> 4;                         // 16 lanes x float4 = 64 channels
    const int c_end = g.c + (g.deloc != DANCE_DELOC_NONE ? 16 : 0);
    for (int c0 = 0; c0 < c_end; c0 += 64) {
        __syncthreads();
        int cbc = CBC;
        if (c0 < g.c) {
#pragma unroll 2
            for (int s = vs; s < v; s += 16) {
#pragma clang fp contract(off)
                const float w0 = tw[s * 4 + 0], w1 = tw[s * 4 + 1], w2 = tw[s * 4 + 2], w3 = tw[s * 4 + 3];
                const int ch = c0 + 4 * lc;
                const float4 a0 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 0] * g.c + ch);
                const float4 a1 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 1] * g.c + ch);
                const float4 a2 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 2] * g.c + ch);
                const float4 a3 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 3] * g.c + ch);
                float val[4];
                val[0] = a0.x * w0 + a1.x * w1 + a2.x * w2 + a3.x * w3;
                val[1] = a0.y * w0 + a1.y * w1 + a2.y * w2 + a3.y * w3;
                val[2] = a0.z * w0 + a1.z * w1 + a2.z * w2 + a3.z * w3;
                val[3] = a0.w * w0 + a1.w * w1 + a2.w * w2 + a3.w * w3;
                h4 hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    _Float16 p, q;
                    dance::split_half(val[i], p, q);
                    hi[i] = p;
                    lo[i] = q;
                }
                char *d = xs + ((long long)(lc >> 1) * vp + s) * 16 + 8 * (lc & 1);
                *reinterpret_cast<h4 *>(d) = hi;
                *reinterpret_cast<h4 *>(d + CBC * vp * 16) = lo;
            }
        } else {
            // de_location chunk: 2 blocks; block 0 = {dx, dy, 0 x 6}, block 1 = zeros
            cbc = 2;
            for (int i = t; i < 2 * v; i += 256) {
                const int blk = i >= v ? 1 : 0, vi = i - blk * v;
                h8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
                if (blk == 0) {
                    _Float16 p, q;
                    dance::split_half(dl[vi], p, q);
                    hi[0] = p; lo[0] = q;
                    dance::split_half(dl[v + vi], p, q);
                    hi[1] = p; lo[1] = q;
                }
                *reinterpret_cast<h8 *>(xs + ((long long)blk * vp + vi) * 16) = hi;
                *reinterpret_cast<h8 *>(xs + ((long long)(2 + blk) * vp + vi) * 16) = lo;
            }
        }
        __syncthreads();
        mfma16_chunk<RT, 1>(acc, xs, v, vp, cbc, a.w + ((long long)(c0 >> 3) + h) * wbs + o0 + r, wps,
                            wts, wbs, 0, a.taps, a.dilation, r, h);
    }
    _Float16 *yslot = a.y + nn * a.y_stride;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        float sc[4], sh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = o0 + 8 * gq + 4 * h + i;
            sc[i] = a.scale[o];
            sh[i] = a.shift[o];
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int vv = rt * 32 + r;
            if (vv >= v)
                continue;
            float val[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = acc[0][rt][4 * gq + i] * a.winv;
                x = x > 0.0f ? x : (x != x ? x : 0.0f);
                val[i] = x * sc[i] + sh[i];
            }
            dance::store_p16(yslot, v, (o0 >> 3) + gq, vv, h, val);
        }
    }
}

int pick_rt16(int v) { return v <= 64 ? 2 : (v <= 128 ? 4 : (v <= 224 ? 7 : 8)); }

template <typename K>
int big_lds(K kernel, size_t bytes, const char *fn)
{
    if (bytes <= 64 * 1024)
        return 0;
    hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    return e == hipSuccess ? 0 : dance::fail((int)e, fn, hipGetErrorString(e));
}

#define LAUNCH_RT(KERNEL, rt, grid, block, lds, s, fn, ...)                                  \
    do {                                                                                      \
        int rc_ = 0;                                                                          \
        switch (rt) {                                                                         \
        case 2: rc_ = big_lds(KERNEL<2>, lds, fn); if (!rc_) hipLaunchKernelGGL(KERNEL<2>, grid, block, lds, s, __VA_ARGS__); break; \
        case 4: rc_ = big_lds(KERNEL<4>, lds, fn); if (!rc_) hipLaunchKernelGGL(KERNEL<4>, grid, block, lds, s, __VA_ARGS__); break; \
        case 7: rc_ = big_lds(KERNEL<7>, lds, fn); if (!rc_) hipLaunchKernelGGL(KERNEL<7>, grid, block, lds, s, __VA_ARGS__); break; \
        default: rc_ = big_lds(KERNEL<8>, lds, fn); if (!rc_) hipLaunchKernelGGL(KERNEL<8>, grid, block, lds, s, __VA_ARGS__); break; \
        }                                                                                     \
        if (rc_)                                                                              \
            return rc_;                                                                       \
    } while (0)

}  // namespace

extern "C" int dance_conv16(const void *x, int64_t x_stride, int64_t x_slot_stride, const void *w16,
                            const float *bias_scaled, float w_inv_scale, const float *scale,
                            const float *shift, const float *rowinit, int residual, int relu,
                            int maxv, void *y16, int64_t y_stride, int64_t y_slot_stride, float *y_f32,
                            int64_t yf_stride, int n, int v, int cin, int cout, int taps, int dilation,
                            void *stream)
{
    DANCE_REQUIRE(n >= 0 && v >= 16 && v <= 256);
    DANCE_REQUIRE(cin >= 16 && cin % 16 == 0 && (cin <= 128 || cin % 128 == 0));
    DANCE_REQUIRE(cout >= 32 && cout % 32 == 0);
    DANCE_REQUIRE(taps == 1 || taps == 9);
    DANCE_REQUIRE(dilation >= 1 && dilation * (taps / 2) <= 16);
    DANCE_REQUIRE((y16 != nullptr) != (y_f32 != nullptr));
    DANCE_REQUIRE(!y16 || cout <= 128 || cout % 128 == 0);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(x && w16);
    Conv16Args a;
    a.x = (const _Float16 *)x; a.x_stride = x_stride; a.x_slot_stride = x_slot_stride;
    a.w = (const h8 *)w16; a.bias_s = bias_scaled; a.scale = scale; a.shift = scale ? shift : nullptr;
    a.winv = w_inv_scale; a.y = (_Float16 *)y16; a.y_stride = y_stride; a.y_slot_stride = y_slot_stride;
    a.yf = y_f32; a.yf_stride = yf_stride; a.v = v; a.cin = cin; a.cout = cout; a.taps = taps;
    a.dilation = dilation; a.relu = relu; a.residual = residual; a.rowinit = rowinit; a.maxv = maxv;
    DANCE_REQUIRE(!maxv || (y_f32 && !residual));
    DANCE_REQUIRE(rowinit || bias_scaled);
    DANCE_REQUIRE(!scale || shift);
    // whole input in LDS when it is a single slot (needed for the LDS residual); else 64-ch chunks
    a.ck = cin <= 128 ? cin : 64;
    DANCE_REQUIRE(!residual || (cin == cout && cin <= 128));
    const size_t lds = (size_t)2 * (a.ck / 8) * (v + 1) * 16;
    dim3 grid(n, dance::ceil_div(cout, 128));
    LAUNCH_RT(k_conv16, pick_rt16(v), grid, dim3(256), lds, (hipStream_t)stream, "dance_conv16", a);
    return launch_check(__func__);
}

extern "C" int dance_fuse16(const void *x, int64_t x_stride, int64_t x_slot_stride, const void *w16,
                            const float *bias_scaled, float w_inv_scale, const float *wg,
                            const float *pred_bias, void *y16, int64_t y_stride, int64_t y_slot_stride,
                            int n, int v, void *stream)
{
    DANCE_REQUIRE(n >= 0 && v >= 16 && v <= 256);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(x && w16 && bias_scaled && wg && pred_bias && y16);
    Fuse16Args a;
    a.x = (const _Float16 *)x; a.x_stride = x_stride; a.x_slot_stride = x_slot_stride;
    a.w = (const h8 *)w16; a.bias_s = bias_scaled; a.wg = wg; a.pb = pred_bias; a.winv = w_inv_scale;
    a.y = (_Float16 *)y16; a.y_stride = y_stride; a.y_slot_stride = y_slot_stride; a.v = v;
    a.ck = 64;
    const size_t lds = (size_t)2 * (a.ck / 8) * (v + 1) * 16;
    DANCE_REQUIRE(v <= 128);        // larger rings: unfused dance_conv16 sequence (accumulators)
    if (v <= 64)
        hipLaunchKernelGGL(k_fuse16<2>, dim3(n), dim3(512), lds, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(k_fuse16<4>, dim3(n), dim3(512), lds, (hipStream_t)stream, a);
    return launch_check(__func__);
}

extern "C" int dance_gather_head16(const float *feat, const float *att, const float *loc,
                                   const int32_t *image_idx, const void *w16, const float *bias_scaled,
                                   float w_inv_scale, const float *scale, const float *shift, void *y16,
                                   int64_t y_stride, float *att_out, int n, int v, int b, int hf, int wf,
                                   int c, int padding_mode, int deloc_type, void *stream)
{
    DANCE_REQUIRE(n >= 0 && v >= 16 && v <= 256);
    DANCE_REQUIRE(b >= 1 && hf >= 1 && wf >= 1 && (long long)hf * wf < (1ll << 30));
    DANCE_REQUIRE(c >= 64 && c % 64 == 0);
    DANCE_REQUIRE(padding_mode == DANCE_PAD_ZEROS || padding_mode == DANCE_PAD_REFLECTION);
    DANCE_REQUIRE(deloc_type >= DANCE_DELOC_DERANGE && deloc_type <= DANCE_DELOC_NONE);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(feat && loc && w16 && bias_scaled && scale && shift && y16);
    Conv16Args a;
    a.x = nullptr; a.x_stride = 0; a.x_slot_stride = 0; a.w = (const h8 *)w16; a.bias_s = bias_scaled;
    a.scale = scale; a.shift = shift; a.winv = w_inv_scale; a.y = (_Float16 *)y16; a.y_stride = y_stride;
    a.y_slot_stride = 0; a.yf = nullptr; a.yf_stride = 0; a.v = v;
    a.cin = (c + (deloc_type != DANCE_DELOC_NONE ? 2 : 0) + 15) & ~15; a.cout = 128; a.taps = 9;
    a.dilation = 1; a.ck = 64; a.relu = 1; a.residual = 0; a.rowinit = nullptr; a.maxv = 0;
    Gather16Args g;
    g.feat = feat; g.att = att; g.loc = loc; g.image_idx = image_idx; g.att_out = att_out; g.b = b;
    g.hf = hf; g.wf = wf; g.c = c; g.reflection = padding_mode == DANCE_PAD_REFLECTION;
    g.deloc = deloc_type;
    const size_t lds = (size_t)2 * 8 * (v + 1) * 16 + ((size_t)10 * v + 16) * sizeof(float);
    LAUNCH_RT(k_gather_head16, pick_rt16(v), dim3(n), dim3(256), lds, (hipStream_t)stream,
              "dance_gather_head16", a, g);
    return launch_check(__func__);
}
