// gather.hip -- HBM-bound kernels of the contour path for gfx950:
//   k_nchw_to_nhwc : (B, [att+]C, H*W) -> NHWC features + attention plane (LDS-tiled transpose)
//   k_gather       : bilinear gather at contour vertices + de_location, written channel-major
//                    (n, C+2, V) for the snake blocks, plus the sampled attention (n, V)
//
// Gather mapping: one workgroup per (contour, block of 32 vertices).  A vertex's four taps
// are four 16-byte-per-lane loads of C contiguous floats (1 KiB per wave-instruction at
// C = 256: the widest coalesced shape), blended in registers and transposed through a
// padded LDS tile so that the global stores are 128-byte row segments.
#include "common.h"
#include "sampling.h"

using dance::launch_check;

// ------------------------------------------------------------------------------------
typedef float f4v __attribute__((ext_vector_type(4)));
#define TP 64   // tile: 64 channels x 64 pixels

__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float *__restrict__ src,
                                                      float *__restrict__ feat,
                                                      float *__restrict__ att, int c_total,
                                                      int att_ch, int hw, uint32_t *absmax)
{
    __shared__ float tile[TP][TP + 1];
    __shared__ unsigned wm[16];
    float amax = 0.0f;
    const int c = c_total - att_ch;
    const int px0 = blockIdx.x * TP;
    const int ch0 = blockIdx.y * TP;
    const long long ib = blockIdx.z;
    const float *s = src + ib * (long long)c_total * hw;
    const int t = threadIdx.x;
    const bool vec_ok = (hw % 4 == 0) && (c % 4 == 0);

    if (att_ch && blockIdx.y == 0 && t < TP && px0 + t < hw)
        att[ib * hw + px0 + t] = s[px0 + t];

    {   // read: rows = channels, 16 lanes x float4 along pixels
        const int q = t & 15, r = t >> 4;
#pragma unroll
        for (int i = 0; i < TP / 16; ++i) {
            const int ch = ch0 + r + 16 * i;
            const int px = px0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ch < c) {
                const float *p = s + (long long)(ch + att_ch) * hw + px;
                if (vec_ok && px + 3 < hw) {
                    // read once, never again: a non-temporal load keeps the 1 GB of NCHW input from
                    // displacing the NHWC output in L2 (0.383 -> 0.353 ms at 16 x 257 x 200 x 304)
                    const f4v t_ = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
                    v = make_float4(t_.x, t_.y, t_.z, t_.w);
                } else {
                    if (px + 0 < hw) v.x = p[0];
                    if (px + 1 < hw) v.y = p[1];
                    if (px + 2 < hw) v.z = p[2];
                    if (px + 3 < hw) v.w = p[3];
                }
            }
            tile[4 * q + 0][r + 16 * i] = v.x;
            tile[4 * q + 1][r + 16 * i] = v.y;
            tile[4 * q + 2][r + 16 * i] = v.z;
            tile[4 * q + 3][r + 16 * i] = v.w;
            // (a NaN must win the max: its bit pattern does, fmaxf would drop it)
            amax = dance::absmax_acc(dance::absmax_acc(dance::absmax_acc(dance::absmax_acc(amax, v.x), v.y), v.z), v.w);
        }
    }
    // per-image max |feature| for the head block's input scale (p16.h): staged in front of the tile's
    // barrier, the workgroup's partial stored by one thread after the write phase
    if (absmax != nullptr)
        dance::absmax_stage(amax, wm);
    __syncthreads();
    {   // write: rows = pixels, 16 lanes x float4 along channels
        const int q = t & 15, r = t >> 4;
        float *d = feat + ib * (long long)hw * c;
#pragma unroll
        for (int i = 0; i < TP / 16; ++i) {
            const int px = px0 + r + 16 * i;
            const int ch = ch0 + 4 * q;
            if (px >= hw || ch >= c)
                continue;
            float4 v;
            v.x = tile[r + 16 * i][4 * q + 0];
            v.y = tile[r + 16 * i][4 * q + 1];
            v.z = tile[r + 16 * i][4 * q + 2];
            v.w = tile[r + 16 * i][4 * q + 3];
            float *p = d + (long long)px * c + ch;
            if (vec_ok) {
                *reinterpret_cast<float4 *>(p) = v;
            } else {
                p[0] = v.x;
                if (ch + 1 < c) p[1] = v.y;
                if (ch + 2 < c) p[2] = v.z;
                if (ch + 3 < c) p[3] = v.w;
            }
        }
    }
    if (absmax != nullptr)                      // absmax: the partials (b, gridDim.y * gridDim.x)
        dance::absmax_partial(wm, absmax + (ib * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
}

// max |x| over each image of an NHWC feature tensor that did not come through dance_nchw_to_nhwc
// or dance_gn_relu (which measure it on the way): one more read of the features
__global__ __launch_bounds__(256) void k_feature_absmax(const float *__restrict__ feat, long long per_image,
                                                        uint32_t *partial)
{
    __shared__ unsigned wm[16];
    const long long ib = blockIdx.y;
    const float *f = feat + ib * per_image;
    float amax = 0.0f;
    const long long n4 = per_image >> 2;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(f) + i);
        amax = dance::absmax_acc(dance::absmax_acc(dance::absmax_acc(dance::absmax_acc(amax, v.x), v.y), v.z), v.w);
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < per_image; i += 256)
            amax = dance::absmax_acc(amax, f[i]);
    dance::absmax_stage(amax, wm);
    __syncthreads();
    dance::absmax_partial(wm, partial + ib * gridDim.x + blockIdx.x);
}

namespace {
int feature_absmax_blocks(long long per_image)
{
    const long long blocks = (per_image / 4 + 255) / 256;
    return (int)(blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks));
}
}  // namespace

extern "C" int64_t dance_absmax_partials(int producer, int c, int64_t hw)
{
    if (c < 1 || hw < 1)
        return 0;
    switch (producer) {
    case DANCE_ABSMAX_OF_LAYOUT: return (int64_t)dance::ceil_div((long long)hw, (long long)TP) * dance::ceil_div(c, TP);
    case DANCE_ABSMAX_OF_GN_RELU: return (int64_t)((hw * 4 + 255) / 256) * (c / 16);
    case DANCE_ABSMAX_OF_FEATURES: return feature_absmax_blocks((long long)c * hw);
    default: return 0;
    }
}

extern "C" int dance_feature_absmax(const float *feat, int b, int64_t per_image, uint32_t *absmax,
                                    uint32_t *absmax_partial, void *stream)
{
    DANCE_REQUIRE(b >= 0 && b <= 65535 && per_image >= 1);
    if (b == 0)
        return 0;
    DANCE_REQUIRE(feat && absmax && absmax_partial && ((uintptr_t)feat & 15) == 0 && (per_image % 4 == 0 || b == 1));
    const int gx = feature_absmax_blocks(per_image);
    hipLaunchKernelGGL(k_feature_absmax, dim3(gx, b), dim3(256), 0, (hipStream_t)stream, feat, (long long)per_image,
                       absmax_partial);
    if (int rc = launch_check(__func__))
        return rc;
    return dance::absmax_reduce_async(absmax_partial, gx, absmax, b, (hipStream_t)stream, __func__);
}

extern "C" int dance_nchw_to_nhwc(const float *src, float *feat, float *att, uint32_t *absmax,
                                  uint32_t *absmax_partial, int b, int c_total, int att_channels, int hw, void *stream)
{
    DANCE_REQUIRE(b >= 0 && hw > 0 && (att_channels == 0 || att_channels == 1));
    DANCE_REQUIRE(c_total > att_channels);
    if (b == 0)
        return 0;
    DANCE_REQUIRE(src && feat && (att_channels == 0 || att));
    DANCE_REQUIRE(b <= 65535);
    DANCE_REQUIRE(absmax == nullptr || absmax_partial != nullptr);
    const int c = c_total - att_channels;
    dim3 grid(dance::ceil_div(hw, TP), dance::ceil_div(c, TP), b);
    hipLaunchKernelGGL(k_nchw_to_nhwc, grid, dim3(256), 0, (hipStream_t)stream, src, feat, att,
                       c_total, att_channels, hw, absmax ? absmax_partial : nullptr);
    if (int rc = launch_check(__func__))
        return rc;
    if (absmax != nullptr)
        return dance::absmax_reduce_async(absmax_partial, (long long)grid.x * grid.y, absmax, b, (hipStream_t)stream, __func__);
    return 0;
}

// ------------------------------------------------------------------------------------
#define GV 32                 // vertices per workgroup
#define GV_PAD (GV + 1)

__global__ __launch_bounds__(256) void k_gather(const float *__restrict__ feat,
                                                const float *__restrict__ att,
                                                const float *__restrict__ loc,
                                                const int *__restrict__ image_idx,
                                                float *__restrict__ x_out,
                                                float *__restrict__ att_out, int v, int b, int hf,
                                                int wf, int c, int reflection, int deloc)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tile = smem;                                   // [c][GV_PAD]
    float *tw = smem + (size_t)c * GV_PAD;                // [GV][4]
    int *toff = reinterpret_cast<int *>(tw + GV * 4);     // [GV][4]
    float *red = reinterpret_cast<float *>(toff + GV * 4);  // [16]

    const int t = threadIdx.x;
    const long long nn = blockIdx.y;
    const int v0 = blockIdx.x * GV;
    const float *loc_n = loc + nn * v * 2;
    int img = image_idx ? image_idx[nn] : b - 1;
    const bool img_ok = img >= 0 && img < b;
    img = img_ok ? img : 0;
    const long long plane = (long long)hf * wf;
    const int c_out = c + (deloc != DANCE_DELOC_NONE ? 2 : 0);

    float xmin = 0.f, xmax = 0.f, ymin = 0.f, ymax = 0.f;
    if (deloc != DANCE_DELOC_NONE)
        dance::contour_bounds(loc_n, v, red, xmin, xmax, ymin, ymax);

    if (t < GV) {
        const int vi = v0 + t;
        if (vi < v) {
            const float lx = loc_n[2 * vi], ly = loc_n[2 * vi + 1];
            dance::Taps tp = dance::bilinear_taps(lx, ly, hf, wf, reflection);
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!img_ok)
                    tp.w[k] = 0.0f;
                tw[t * 4 + k] = tp.w[k];
                toff[t * 4 + k] = tp.off[k];
            }
            if (att_out != nullptr) {
                if (att != nullptr) {
#pragma clang fp contract(off)
                    const float *ap = att + img * plane;
                    a = ap[tp.off[0]] * tp.w[0] + ap[tp.off[1]] * tp.w[1] +
                        ap[tp.off[2]] * tp.w[2] + ap[tp.off[3]] * tp.w[3];
                }
                att_out[nn * v + vi] = a;
            }
            if (deloc != DANCE_DELOC_NONE) {
                float *xo = x_out + (nn * c_out + c) * v + vi;
                xo[0] = dance::deloc_value(lx, xmin, xmax, deloc);
                xo[v] = dance::deloc_value(ly, ymin, ymax, deloc);
            }
        }
    }
    __syncthreads();

    const int lane = t & 63, wave = t >> 6;
    const float *fimg = feat + img * plane * c;
    for (int s = wave; s < GV; s += 4) {
        if (v0 + s >= v)
            break;
        const float w0 = tw[s * 4 + 0], w1 = tw[s * 4 + 1], w2 = tw[s * 4 + 2], w3 = tw[s * 4 + 3];
        const float *p0 = fimg + (long long)toff[s * 4 + 0] * c;
        const float *p1 = fimg + (long long)toff[s * 4 + 1] * c;
        const float *p2 = fimg + (long long)toff[s * 4 + 2] * c;
        const float *p3 = fimg + (long long)toff[s * 4 + 3] * c;
        for (int ch = lane * 4; ch < c; ch += 256) {
#pragma clang fp contract(off)
            const float4 a0 = *reinterpret_cast<const float4 *>(p0 + ch);
            const float4 a1 = *reinterpret_cast<const float4 *>(p1 + ch);
            const float4 a2 = *reinterpret_cast<const float4 *>(p2 + ch);
            const float4 a3 = *reinterpret_cast<const float4 *>(p3 + ch);
            tile[(ch + 0) * GV_PAD + s] = a0.x * w0 + a1.x * w1 + a2.x * w2 + a3.x * w3;
            tile[(ch + 1) * GV_PAD + s] = a0.y * w0 + a1.y * w1 + a2.y * w2 + a3.y * w3;
            tile[(ch + 2) * GV_PAD + s] = a0.z * w0 + a1.z * w1 + a2.z * w2 + a3.z * w3;
            tile[(ch + 3) * GV_PAD + s] = a0.w * w0 + a1.w * w1 + a2.w * w2 + a3.w * w3;
        }
    }
    __syncthreads();

    const int col = t & 31;
    if (v0 + col < v) {
        float *xo = x_out + nn * c_out * (long long)v + v0 + col;
        for (int ch = t >> 5; ch < c; ch += 8)
            xo[(long long)ch * v] = tile[ch * GV_PAD + col];
    }
}

extern "C" int dance_gather_features(const float *feat, const float *att, const float *loc,
                                     const int32_t *image_idx, float *x_out, float *att_out, int n,
                                     int v, int b, int hf, int wf, int c, int padding_mode,
                                     int deloc_type, void *stream)
{
    DANCE_REQUIRE(n >= 0 && v >= 1 && b >= 1 && hf >= 1 && wf >= 1);
    DANCE_REQUIRE(c >= 4 && c % 4 == 0 && c <= 1024);
    DANCE_REQUIRE(padding_mode == DANCE_PAD_ZEROS || padding_mode == DANCE_PAD_REFLECTION);
    DANCE_REQUIRE(deloc_type >= DANCE_DELOC_DERANGE && deloc_type <= DANCE_DELOC_NONE);
    DANCE_REQUIRE((long long)hf * wf < (1ll << 30));
    if (n == 0)
        return 0;
    DANCE_REQUIRE(feat && loc && x_out);
    DANCE_REQUIRE(n <= 65535 * 16);
    const size_t lds = ((size_t)c * GV_PAD + GV * 4 + GV * 4 + 16) * sizeof(float);
    if (int rc = dance::allow_lds((const void *)k_gather, lds, __func__))
        return rc;
    // contours go on grid.y (<= 65535): split very large n over several launches
    for (int n0 = 0; n0 < n; n0 += 65535) {
        const int nb = n - n0 < 65535 ? n - n0 : 65535;
        const int c_out = c + (deloc_type != DANCE_DELOC_NONE ? 2 : 0);
        hipLaunchKernelGGL(k_gather, dim3(dance::ceil_div(v, GV), nb), dim3(256), lds,
                           (hipStream_t)stream, feat, att, loc + (long long)n0 * v * 2,
                           image_idx ? image_idx + n0 : nullptr,
                           x_out + (long long)n0 * c_out * v,
                           att_out ? att_out + (long long)n0 * v : nullptr, v, b, hf, wf, c,
                           padding_mode == DANCE_PAD_REFLECTION, deloc_type);
        int rc = launch_check(__func__);
        if (rc)
            return rc;
    }
    return 0;
}
