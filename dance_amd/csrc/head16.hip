// head16.hip -- fused gather + de_location + head SnakeBlock of the fast precision mode on
// v_mfma_f32_16x16x32_f16 with 16-wide vertex tiles (RefineNet.evolve snake_head.py:1639-1652,
// get_locations_feature :1542-1572, de_location :1587-1610, _SnakeNet.head :233).
//
// Same work as k_gather_head16 (snake16.hip: 32 x 32 x 16 tiles, the ring padded to a multiple of
// 32 columns) with two differences:
//   * a wave's 32 outputs x the whole ring are 2 row tiles x CT column tiles of 16, CT =
//     ceil(v / 16) rounded up to an instantiated count: 196 vertices run 13 tiles (208 columns)
//     instead of 224, 40 vertices 3 (48) instead of 64;
//   * the two de_location channels do not ride a zero-padded 32-channel chunk through all 9 ring
//     taps; their 2 x 9 (channel, tap) pairs are laid out as ONE 32-deep K step (an im2col column
//     per vertex: dx at the 9 ring offsets, dy at the 9 ring offsets, 14 zeros) against the
//     matching 128 x 32 slice of the head weights (`head_wd`, packed with the head's scale).
// Results differ from k_gather_head16's only in summation order (~1e-7 relative).
#include "common.h"
#include "mfma16.h"
#include "p16.h"
#include "sampling.h"

using dance::launch_check;

#ifdef DANCE_EXP_STAMPS
// experiment build only (tools/build_variant.py): phase time stamps of k_gather_head16w, 100 MHz real-time clock
__device__ unsigned long long g_head_stamps[2048][20];
extern "C" int dance_exp_head_stamps(void *dst)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_head_stamps), sizeof(g_head_stamps));
}
#define STAMP(i)                                                                                      \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x < 2048)                                                    \
            g_head_stamps[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime();                          \
    } while (0)
#else
#define STAMP(i)
#endif

namespace {

struct Head16Args {
    const float *feat, *att, *loc;
    const int *image_idx;
    float *att_out;
    int b, hf, wf, c, reflection, deloc;
    const h8 *w;                // head weights [plane][9][cin/8][128] words (feature blocks used)
    int cin;                    // padded input channels of w
    const h8 *wd;               // de_location im2col weights [plane][4][128] words
    const uint32_t *absmax;     // (b, DANCE_ABSMAX_WORDS) max |feature| per image, or null (p16.h)
    const float *bias_s, *scale, *shift;
    float winv, wdinv;          // inverse weight scales of w / wd (powers of two)
    _Float16 *y;                // state slot 0 (P16)
    long long y_stride;
    int v;
};

// MINB: workgroups per CU the register budget is cut for (3: 168 VGPRs, two B buffers instead of four)
template <int RTG, int CT, int TERMS, int MINB = 2>
__global__ __launch_bounds__(256, MINB) void k_gather_head16w(Head16Args a)
{
    constexpr int NST = 2, AD = 2, NB = (NST * CT) % 4 == 0 && CT >= 3 && MINB == 2 ? 4 : 2;
    static_assert(CT * 16 <= RTG * 32, "the column tiles fit the image");
    using G = Geo<RTG, 8>;
    using GD = Geo<RTG, 4>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int v = a.v;
    char *xs = smem;                                            // one chunk image
    float *tw = reinterpret_cast<float *>(smem + G::BYTES);     // [v][4]
    int *toff = reinterpret_cast<int *>(tw + 4 * v);            // [v][4]
    float *dl = reinterpret_cast<float *>(toff + 4 * v);        // [2][v]
    float *red = dl + 2 * v;                                    // [16]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c16 = lane & 15, q4 = lane >> 4;
    const long long nn = blockIdx.x;
    const int o0 = wave * 32;
    const float *loc_n = a.loc + nn * v * 2;
    int img = a.image_idx ? a.image_idx[nn] : a.b - 1;
    const bool img_ok = img >= 0 && img < a.b;
    img = img_ok ? img : 0;
    const long long plane = (long long)a.hf * a.wf;

    STAMP(0);
#ifdef DANCE_EXP_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 2048) {
        g_head_stamps[blockIdx.x][17] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_ID
        g_head_stamps[blockIdx.x][18] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // XCC_ID
    }
#endif
    float xmin = 0.f, xmax = 0.f, ymin = 0.f, ymax = 0.f;
    if (a.deloc != DANCE_DELOC_NONE)
        dance::contour_bounds(loc_n, v, red, xmin, xmax, ymin, ymax);
    // the block's input (features | de_location) is split at per-contour power-of-two scales (p16.h):
    // folded into the tap weights and the de_location values, undone in the epilogue
    const float dmax = a.deloc == DANCE_DELOC_DEMIN ? fmaxf(xmax - xmin, ymax - ymin) : 1.0f;
    const dance::HeadInputScales hs = dance::head_input_scales(
        a.winv, a.wdinv, a.absmax != nullptr,
        a.absmax ? dance::feature_scale_exp(a.absmax + __builtin_amdgcn_readfirstlane(img) * DANCE_ABSMAX_WORDS) : 0,
        a.deloc != DANCE_DELOC_NONE, dance::bound_scale_exp(dmax));
    const float fs = hs.fs, ds = hs.ds;
    for (int vi = t; vi < v; vi += 256) {
        const float lx = loc_n[2 * vi], ly = loc_n[2 * vi + 1];
        dance::Taps tp = dance::bilinear_taps(lx, ly, a.hf, a.wf, a.reflection);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tw[vi * 4 + k] = img_ok ? tp.w[k] * fs : 0.0f;
            toff[vi * 4 + k] = tp.off[k];
        }
        if (a.att_out != nullptr) {
            float s = 0.0f;
            if (a.att != nullptr) {
#pragma clang fp contract(off)
                const float *ap = a.att + img * plane;
                s = ap[tp.off[0]] * tp.w[0] + ap[tp.off[1]] * tp.w[1] + ap[tp.off[2]] * tp.w[2] +
                    ap[tp.off[3]] * tp.w[3];
                s = img_ok ? s : 0.0f;
            }
            a.att_out[nn * v + vi] = s;
        }
        if (a.deloc != DANCE_DELOC_NONE) {
            dl[vi] = dance::deloc_value(lx, xmin, xmax, a.deloc) * ds;
            dl[v + vi] = dance::deloc_value(ly, ymin, ymax, a.deloc) * ds;
        }
    }

    f32x4 acc[2][CT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float4 b4 = *reinterpret_cast<const float4 *>(a.bias_s + o0 + 16 * i + 4 * q4);
        const f32x4 bv = {b4.x * fs, b4.y * fs, b4.z * fs, b4.w * fs};
#pragma unroll
        for (int j = 0; j < CT; ++j)
            acc[i][j] = bv;
    }

    if (a.deloc != DANCE_DELOC_NONE) {
        // de_location as one K step: word (block kb, vertex vi) = im2col entries 8 kb .. 8 kb + 7 of
        // [dx(vi - 4 .. vi + 4), dy(vi - 4 .. vi + 4), 0 x 14]
        __syncthreads();
        for (int i = t; i < 4 * v; i += 256) {
            const int kb = i / v, vi = i - kb * v;
            h8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kb * 8 + e;
                if (k < 18) {
                    const int ch = k >= 9, tap = k - 9 * ch;
                    _Float16 p, q;
                    dance::split_half(dl[ch * v + ring16(vi, tap - 4, v)], p, q);
                    hi[e] = p;
                    lo[e] = q;
                }
            }
            *reinterpret_cast<h8 *>(xs + (kb * GD::VP + vi) * 16) = hi;
            *reinterpret_cast<h8 *>(xs + GD::PLANE + (kb * GD::VP + vi) * 16) = lo;
        }
        __syncthreads();
        h8 ah[2], al[2];
        const char *wdp = reinterpret_cast<const char *>(a.wd) + (unsigned)(q4 * 128 + o0 + c16) * 16u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const h8 *>(wdp + i * 256);
            if constexpr (TERMS == 3)
                al[i] = *reinterpret_cast<const h8 *>(wdp + 4 * 128 * 16 + i * 256);
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const char *bq = xs + (unsigned)(q4 * GD::VP + min(j * 16 + c16, v - 1)) * 16u;
            const h8 bh = *reinterpret_cast<const h8 *>(bq);
            if constexpr (TERMS == 3) {
                const h8 bl = *reinterpret_cast<const h8 *>(bq + GD::PLANE);
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[0], bh, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[1], bh, acc[1][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], bl, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], bl, acc[1][j], 0, 0, 0);
            }
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], bh, acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], bh, acc[1][j], 0, 0, 0);
        }
    }

    const long long wbs_b = 128ll * 16, wts_b = (long long)(a.cin >> 3) * wbs_b, wps_b = 9 * wts_b;
    const unsigned wlane = (unsigned)(q4 * 128 + c16) * 16u;
    const float *fimg = a.feat + img * plane * a.c;
    const int lc = t & 15, vs = t >> 4;                         // 16 lanes x float4 = 64 channels
    auto nohook = [](int) __attribute__((always_inline)) {};
    STAMP(1);
    for (int c0 = 0; c0 < a.c; c0 += 64) {
        __syncthreads();
        STAMP(2 + (c0 >> 6) * 3);
#pragma unroll 4
        for (int s = vs; s < v; s += 16) {
#pragma clang fp contract(off)
            const float w0 = tw[s * 4 + 0], w1 = tw[s * 4 + 1], w2 = tw[s * 4 + 2], w3 = tw[s * 4 + 3];
            const int ch = c0 + 4 * lc;
            const float4 a0 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 0] * a.c + ch);
            const float4 a1 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 1] * a.c + ch);
            const float4 a2 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 2] * a.c + ch);
            const float4 a3 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 3] * a.c + ch);
            float val[4];
            val[0] = a0.x * w0 + a1.x * w1 + a2.x * w2 + a3.x * w3;
            val[1] = a0.y * w0 + a1.y * w1 + a2.y * w2 + a3.y * w3;
            val[2] = a0.z * w0 + a1.z * w1 + a2.z * w2 + a3.z * w3;
            val[3] = a0.w * w0 + a1.w * w1 + a2.w * w2 + a3.w * w3;
            h4 hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                _Float16 p, q;
                dance::split_half_sum(val[i], p, q);
                hi[i] = p;
                lo[i] = q;
            }
            char *d = xs + ((lc >> 1) * G::VP + s) * 16 + 8 * (lc & 1);
            *reinterpret_cast<h4 *>(d) = hi;
            *reinterpret_cast<h4 *>(d + G::PLANE) = lo;
        }
        STAMP(3 + (c0 >> 6) * 3);
        __syncthreads();
        STAMP(4 + (c0 >> 6) * 3);
        mfma16_core_sw<RTG, CT, NST, AD, NB, TERMS>(acc, xs, 0, v,
                                                    reinterpret_cast<const char *>(a.w) + ((long long)(c0 >> 3) * 128 + o0) * 16,
                                                    wlane, wps_b, wts_b, 4 * wbs_b, 9, 1, c16, q4, 0, nohook);
    }

    // epilogue: lane = vertex column 16 j + c16, row tile i = channels 16 i + 4 q4 .. + 3
    _Float16 *yslot = a.y + nn * a.y_stride;
    const float winv = hs.inv;
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
        const int ob = o0 + 16 * i2 + 4 * q4;
        const float4 sc4 = *reinterpret_cast<const float4 *>(a.scale + ob);
        const float4 sh4 = *reinterpret_cast<const float4 *>(a.shift + ob);
        const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int vv = j * 16 + c16;
            if (vv >= v)
                continue;
            float val[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                val[i] = relu_nan(acc[i2][j][i] * winv) * sc[i] + sh[i];
            dance::store_p16(yslot, v, ob >> 3, vv, (ob >> 2) & 1, val);
        }
    }
    STAMP(16);
}

template <int RTG, int CT>
int launch_head(const Head16Args &a, int n, int terms, size_t lds, hipStream_t s)
{
    // 65..128 vertices: 38.5 KB of LDS per workgroup, so three fit a CU when the kernel is held to
    // 168 VGPRs (a few spilled outside the MFMA loop): a third workgroup's gather phases fill the
    // other two's barrier waits (head 0.356 -> 0.335 ms at 1600 x 128, same box)
    constexpr int MINB = RTG == 4 ? 3 : 2;
    int rc;
    if (terms == 1) {
        if ((rc = dance::allow_lds((const void *)k_gather_head16w<RTG, CT, 1, MINB>, lds, "dance_gather_head16w")))
            return rc;
        hipLaunchKernelGGL((k_gather_head16w<RTG, CT, 1, MINB>), dim3(n), dim3(256), lds, s, a);
    } else {
        if ((rc = dance::allow_lds((const void *)k_gather_head16w<RTG, CT, 3, MINB>, lds, "dance_gather_head16w")))
            return rc;
        hipLaunchKernelGGL((k_gather_head16w<RTG, CT, 3, MINB>), dim3(n), dim3(256), lds, s, a);
    }
    return launch_check("dance_gather_head16w");
}

// Rings above 208 vertices: two workgroups per contour, each owning half of the ring's vertex
// tiles (the head block has one ring convolution, so a half needs only 4 halo vertices either
// side -- gathered twice, 8 of <= 128 columns).  A half is then the (4, 8) shape of the kernel above: 4
// waves x 8 column tiles, three workgroups per CU.  LDS image columns: [0, len + 4) = vertices
// h0 .. h0 + len + 3, the last 4 columns of the image's ring period (VC = 16 CT + 8) = vertices
// h0 - 4 .. h0 - 1, so the core's ring walk with period VC reads the right neighbours unchanged.
// Also used for 65..128 vertices when there are fewer contours than CUs (one image's ~100
// detections): twice the workgroups, each half as long.
template <int RTG, int CT, int TERMS>
__global__ __launch_bounds__(256, 3) void k_gather_head16h(Head16Args a, int h1)
{
    constexpr int NST = 2, AD = 2, NB = 2, VC = 16 * CT + 8;
    static_assert(VC <= RTG * 32, "the half ring and its halo fit the image");
    using G = Geo<RTG, 8>;
    using GD = Geo<RTG, 4>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int v = a.v;
    const int h0 = blockIdx.y ? h1 : 0, len = blockIdx.y ? v - h1 : h1, ns = len + 8;
    char *xs = smem;
    float *tw = reinterpret_cast<float *>(smem + G::BYTES);     // [VC][4]
    int *toff = reinterpret_cast<int *>(tw + 4 * VC);           // [VC][4]
    float *dl = reinterpret_cast<float *>(toff + 4 * VC);       // [2][v]
    float *red = dl + 2 * v;                                    // [16]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c16 = lane & 15, q4 = lane >> 4;
    const long long nn = blockIdx.x;
    const int o0 = wave * 32;
    const float *loc_n = a.loc + nn * v * 2;
    int img = a.image_idx ? a.image_idx[nn] : a.b - 1;
    const bool img_ok = img >= 0 && img < a.b;
    img = img_ok ? img : 0;
    const long long plane = (long long)a.hf * a.wf;

    float xmin = 0.f, xmax = 0.f, ymin = 0.f, ymax = 0.f;
    if (a.deloc != DANCE_DELOC_NONE)
        dance::contour_bounds(loc_n, v, red, xmin, xmax, ymin, ymax);
    // per-contour input scales, as in k_gather_head16w (both halves of a ring compute the same ones)
    const float dmax = a.deloc == DANCE_DELOC_DEMIN ? fmaxf(xmax - xmin, ymax - ymin) : 1.0f;
    const dance::HeadInputScales hs = dance::head_input_scales(
        a.winv, a.wdinv, a.absmax != nullptr,
        a.absmax ? dance::feature_scale_exp(a.absmax + __builtin_amdgcn_readfirstlane(img) * DANCE_ABSMAX_WORDS) : 0,
        a.deloc != DANCE_DELOC_NONE, dance::bound_scale_exp(dmax));
    const float fs = hs.fs, ds = hs.ds;
    if (a.deloc != DANCE_DELOC_NONE) {
        for (int vi = t; vi < v; vi += 256) {
            dl[vi] = dance::deloc_value(loc_n[2 * vi], xmin, xmax, a.deloc) * ds;
            dl[v + vi] = dance::deloc_value(loc_n[2 * vi + 1], ymin, ymax, a.deloc) * ds;
        }
    }
    // gather slot s = ring offset s - 4 from the half's first vertex
    for (int s = t; s < ns; s += 256) {
        const int vi = ring16(h0, s - 4, v);
        const float lx = loc_n[2 * vi], ly = loc_n[2 * vi + 1];
        dance::Taps tp = dance::bilinear_taps(lx, ly, a.hf, a.wf, a.reflection);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tw[s * 4 + k] = img_ok ? tp.w[k] * fs : 0.0f;
            toff[s * 4 + k] = tp.off[k];
        }
        if (a.att_out != nullptr && s >= 4 && s < len + 4) {
            float sa = 0.0f;
            if (a.att != nullptr) {
#pragma clang fp contract(off)
                const float *ap = a.att + img * plane;
                sa = ap[tp.off[0]] * tp.w[0] + ap[tp.off[1]] * tp.w[1] + ap[tp.off[2]] * tp.w[2] +
                     ap[tp.off[3]] * tp.w[3];
                sa = img_ok ? sa : 0.0f;
            }
            a.att_out[nn * v + vi] = sa;
        }
    }

    f32x4 acc[2][CT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float4 b4 = *reinterpret_cast<const float4 *>(a.bias_s + o0 + 16 * i + 4 * q4);
        const f32x4 bv = {b4.x * fs, b4.y * fs, b4.z * fs, b4.w * fs};
#pragma unroll
        for (int j = 0; j < CT; ++j)
            acc[i][j] = bv;
    }

    if (a.deloc != DANCE_DELOC_NONE) {
        __syncthreads();
        for (int i = t; i < 4 * len; i += 256) {
            const int kb = i / len, m = i - kb * len;
            h8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kb * 8 + e;
                if (k < 18) {
                    const int ch = k >= 9, tap = k - 9 * ch;
                    _Float16 p, q;
                    dance::split_half(dl[ch * v + ring16(h0 + m, tap - 4, v)], p, q);
                    hi[e] = p;
                    lo[e] = q;
                }
            }
            *reinterpret_cast<h8 *>(xs + (kb * GD::VP + m) * 16) = hi;
            *reinterpret_cast<h8 *>(xs + GD::PLANE + (kb * GD::VP + m) * 16) = lo;
        }
        __syncthreads();
        h8 ah[2], al[2];
        const char *wdp = reinterpret_cast<const char *>(a.wd) + (unsigned)(q4 * 128 + o0 + c16) * 16u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const h8 *>(wdp + i * 256);
            if constexpr (TERMS == 3)
                al[i] = *reinterpret_cast<const h8 *>(wdp + 4 * 128 * 16 + i * 256);
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const char *bq = xs + (unsigned)(q4 * GD::VP + min(j * 16 + c16, len - 1)) * 16u;
            const h8 bh = *reinterpret_cast<const h8 *>(bq);
            if constexpr (TERMS == 3) {
                const h8 bl = *reinterpret_cast<const h8 *>(bq + GD::PLANE);
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[0], bh, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[1], bh, acc[1][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], bl, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], bl, acc[1][j], 0, 0, 0);
            }
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], bh, acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], bh, acc[1][j], 0, 0, 0);
        }
    }

    const long long wbs_b = 128ll * 16, wts_b = (long long)(a.cin >> 3) * wbs_b, wps_b = 9 * wts_b;
    const unsigned wlane = (unsigned)(q4 * 128 + c16) * 16u;
    const float *fimg = a.feat + img * plane * a.c;
    const int lc = t & 15, vs = t >> 4;
    auto nohook = [](int) __attribute__((always_inline)) {};
    // image columns never gathered (a short half's tail tiles) feed only MFMA columns the epilogue drops
    for (int c0 = 0; c0 < a.c; c0 += 64) {
        __syncthreads();
#pragma unroll 2
        for (int s = vs; s < ns; s += 16) {
#pragma clang fp contract(off)
            const float w0 = tw[s * 4 + 0], w1 = tw[s * 4 + 1], w2 = tw[s * 4 + 2], w3 = tw[s * 4 + 3];
            const int ch = c0 + 4 * lc;
            const float4 a0 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 0] * a.c + ch);
            const float4 a1 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 1] * a.c + ch);
            const float4 a2 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 2] * a.c + ch);
            const float4 a3 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 3] * a.c + ch);
            float val[4];
            val[0] = a0.x * w0 + a1.x * w1 + a2.x * w2 + a3.x * w3;
            val[1] = a0.y * w0 + a1.y * w1 + a2.y * w2 + a3.y * w3;
            val[2] = a0.z * w0 + a1.z * w1 + a2.z * w2 + a3.z * w3;
            val[3] = a0.w * w0 + a1.w * w1 + a2.w * w2 + a3.w * w3;
            h4 hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                _Float16 p, q;
                dance::split_half_sum(val[i], p, q);
                hi[i] = p;
                lo[i] = q;
            }
            const int x = s >= 4 ? s - 4 : VC - 4 + s;
            char *d = xs + ((lc >> 1) * G::VP + x) * 16 + 8 * (lc & 1);
            *reinterpret_cast<h4 *>(d) = hi;
            *reinterpret_cast<h4 *>(d + G::PLANE) = lo;
        }
        __syncthreads();
        mfma16_core_sw<RTG, CT, NST, AD, NB, TERMS>(acc, xs, 0, VC,
                                                    reinterpret_cast<const char *>(a.w) + ((long long)(c0 >> 3) * 128 + o0) * 16,
                                                    wlane, wps_b, wts_b, 4 * wbs_b, 9, 1, c16, q4, 0, nohook);
    }

    _Float16 *yslot = a.y + nn * a.y_stride;
    const float winv = hs.inv;
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
        const int ob = o0 + 16 * i2 + 4 * q4;
        const float4 sc4 = *reinterpret_cast<const float4 *>(a.scale + ob);
        const float4 sh4 = *reinterpret_cast<const float4 *>(a.shift + ob);
        const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int m = j * 16 + c16;
            if (m >= len)
                continue;
            float val[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                val[i] = relu_nan(acc[i2][j][i] * winv) * sc[i] + sh[i];
            dance::store_p16(yslot, v, ob >> 3, h0 + m, (ob >> 2) & 1, val);
        }
    }
}

template <int RTG, int CT>
int launch_head_halves(const Head16Args &a, int n, int terms, hipStream_t s)
{
    constexpr int VC = 16 * CT + 8;
    const int h1 = ((a.v + 15) / 16 + 1) / 2 * 16;              // tiles split evenly or first half one more
    if (h1 > 16 * CT || a.v - h1 > 16 * CT || a.v - h1 < 1)
        return dance::fail(1, "dance_gather_head16w", "ring length does not fit the half-ring kernel");
    const size_t lds = (size_t)2 * 8 * (RTG * 32 + 1) * 16 + ((size_t)8 * VC + 2 * a.v + 16) * sizeof(float);
    int rc;
    if (terms == 1) {
        if ((rc = dance::allow_lds((const void *)k_gather_head16h<RTG, CT, 1>, lds, "dance_gather_head16w")))
            return rc;
        hipLaunchKernelGGL((k_gather_head16h<RTG, CT, 1>), dim3(n, 2), dim3(256), lds, s, a, h1);
    } else {
        if ((rc = dance::allow_lds((const void *)k_gather_head16h<RTG, CT, 3>, lds, "dance_gather_head16w")))
            return rc;
        hipLaunchKernelGGL((k_gather_head16h<RTG, CT, 3>), dim3(n, 2), dim3(256), lds, s, a, h1);
    }
    return launch_check("dance_gather_head16w");
}

}  // namespace

extern "C" int dance_gather_head16w(const float *feat, const float *att, const float *loc,
                                    const int32_t *image_idx, const void *w16, const void *wd16,
                                    const float *bias_scaled, float w_inv_scale, float wd_inv_scale,
                                    const float *scale, const float *shift, void *y16, int64_t y_stride, float *att_out, int n,
                                    int v, int b, int hf, int wf, int c, int padding_mode, int deloc_type,
                                    int mfma_terms, const uint32_t *feat_absmax, const dance_policy *policy,
                                    void *stream)
{
    DANCE_REQUIRE(mfma_terms == 3 || mfma_terms == 1);
    DANCE_REQUIRE(n >= 0 && v >= 16 && v <= 256);
    DANCE_REQUIRE(b >= 1 && hf >= 1 && wf >= 1 && (long long)hf * wf < (1ll << 30));
    DANCE_REQUIRE(c >= 64 && c % 64 == 0);
    DANCE_REQUIRE(padding_mode == DANCE_PAD_ZEROS || padding_mode == DANCE_PAD_REFLECTION);
    DANCE_REQUIRE(deloc_type >= DANCE_DELOC_DERANGE && deloc_type <= DANCE_DELOC_NONE);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(feat && loc && w16 && bias_scaled && scale && shift && y16);
    DANCE_REQUIRE(deloc_type == DANCE_DELOC_NONE || wd16 != nullptr);
    DANCE_REQUIRE(dance::is_pow2_scale(w_inv_scale) && (deloc_type == DANCE_DELOC_NONE || dance::is_pow2_scale(wd_inv_scale)));
    Head16Args a;
    a.feat = feat; a.att = att; a.loc = loc; a.image_idx = image_idx; a.att_out = att_out; a.b = b;
    a.hf = hf; a.wf = wf; a.c = c; a.reflection = padding_mode == DANCE_PAD_REFLECTION; a.deloc = deloc_type;
    a.w = (const h8 *)w16; a.cin = (c + (deloc_type != DANCE_DELOC_NONE ? 2 : 0) + 31) & ~31;
    a.wd = (const h8 *)wd16; a.absmax = feat_absmax; a.bias_s = bias_scaled; a.scale = scale; a.shift = shift; a.winv = w_inv_scale; a.wdinv = wd_inv_scale;
    a.y = (_Float16 *)y16; a.y_stride = y_stride; a.v = v;
    hipStream_t s = (hipStream_t)stream;
    // 14 and more column tiles do not fit the register file beside the rotating operand buffers:
    // rings above 208 vertices run as two half rings
    if (v > DANCE_HEAD16W_MAX_V)
        return launch_head_halves<5, 8>(a, n, mfma_terms, s);
    // fewer contours than CUs (one image's detections): 65..128-vertex rings as two half rings too,
    // twice the workgroups at half the length each (policy.few_contour_forms, up to DANCE_FEW_HEAD)
    if (v > 64 && v <= 128 && dance::policy_or_default(policy).few_contour_forms && n <= DANCE_FEW_HEAD)
        return v <= 96 ? launch_head_halves<2, 3>(a, n, mfma_terms, s) : launch_head_halves<3, 4>(a, n, mfma_terms, s);
    const int ct = (v + 15) / 16;
    // 129..208 vertices (9..13 tiles) as two half rings of 5, 6 or 7 tiles as well: twice the
    // workgroups, three per CU
    if (v > 128) {
        if (ct <= 10)
            return launch_head_halves<3, 5>(a, n, mfma_terms, s);
        return ct <= 12 ? launch_head_halves<4, 6>(a, n, mfma_terms, s) : launch_head_halves<4, 7>(a, n, mfma_terms, s);
    }
    const int rtg = v <= 64 ? 2 : 4;
    const size_t lds = (size_t)2 * 8 * (rtg * 32 + 1) * 16 + ((size_t)10 * v + 16) * sizeof(float);
    if (rtg == 2)
        return ct <= 2 ? launch_head<2, 2>(a, n, mfma_terms, lds, s)
                       : (ct == 3 ? launch_head<2, 3>(a, n, mfma_terms, lds, s) : launch_head<2, 4>(a, n, mfma_terms, lds, s));
    if (ct <= 5)
        return launch_head<4, 5>(a, n, mfma_terms, lds, s);
    if (ct == 6)
        return launch_head<4, 6>(a, n, mfma_terms, lds, s);
    return ct == 7 ? launch_head<4, 7>(a, n, mfma_terms, lds, s) : launch_head<4, 8>(a, n, mfma_terms, lds, s);
}
