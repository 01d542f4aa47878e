// init16.hip -- Dsnake's init stage up to the init snake's head block, one launch, fast precision
// mode (DsnakeHead.init, dsnake_head.py:718-733 + _SnakeNet.head):
//   sampled = bilinear gather of the C feature channels at the ring's vertices (zeros padding)
//   centre  = the same gather at (min + max) / 2 of the ring                     (:722-723)
//   x       = fuse([sampled | centre expanded])   Conv1d 2C -> C, k = 1            (:726-729)
//           = W_s * sampled + (b + W_c * centre)   -- the centre half is one row constant
//   head    = BN(relu(CircConv_9([x | de_location])))                              (:730-736)
// Replaces two gather launches, the min / max / copy ops between them, dance_row_constant, the
// exact-fp32 k = 1 convolution and k_xin_head16; the (n, C + 2, v) fp32 block between them never
// goes to HBM.
//
// One workgroup (4 waves) per contour, C = 256, rings of at most 64 vertices (Dsnake: 40).  The
// gathered channels are split into hi/lo halves into four 64-channel LDS chunk images; the fuse
// GEMM (wave w: outputs 64 w .. 64 w + 63, three f16 MFMAs per product) reads them, and its
// result overwrites them as the head block's input.
#include "common.h"
#include "mfma16.h"
#include "p16.h"
#include "sampling.h"

using dance::launch_check;

namespace {

struct InitHead16Args {
    const float *feat, *loc;      // (b, hf, wf, 256) NHWC, (n, v, 2)
    const int *image_idx;
    const uint32_t *absmax;       // (b, DANCE_ABSMAX_WORDS) max |feature| per image, or null (p16.h)
    int b, hf, wf, reflection, deloc;
    const h8 *fw;                 // fuse, sampled half: [plane][1][32][256] words
    float fwinv;
    const float *fwg, *fb;        // fuse, centre half [g][o] (256, 256); bias (256)
    const h8 *hw;                 // head block: [plane][9][36][128] words (258 -> 288 channels; feature blocks used)
    const h8 *hwd;                // head block, de_location channels as one K step: [plane][4][128] words (head16.hip)
    const float *hbs, *hscale, *hshift;
    float hwinv, hwdinv;          // inverse scales of hw / hwd (powers of two)
    _Float16 *y;                  // state slot 0 (P16)
    long long y_stride;
    int v;
};

// CT: 16-wide vertex tiles of the head block (3 for Dsnake's 40-point ring)
template <int RT, int CT, int TERMS>
__global__ __launch_bounds__(256, 2) void k_init_head16(InitHead16Args a)
{
    constexpr int C = 256, NCH = C / 64;
    static_assert(CT * 16 <= RT * 32, "the column tiles fit the image");
    using G = Geo<RT, 8>;
    using GD = Geo<RT, 4>;                                      // de_location chunk: 32 channels
    static_assert(4 * 256 * 4 + 256 * 4 + 2 * (RT * 32 + 1) * 16 <= GD::BYTES, "scratch under the de_location image");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int v = a.v;
    char *img = smem;                                           // NCH chunk images
    char *dimg = smem + NCH * G::BYTES;                         // de_location image, over the scratch:
    float *hp = reinterpret_cast<float *>(dimg);                //   [4][256] GEMV partial sums
    float *cf = hp + 4 * 256;                                   //   [256] centre feature
    float *tw = cf + 256;                                       //   [v + 1][4] tap weights (v = centre)
    int *toff = reinterpret_cast<int *>(tw + 4 * (RT * 32 + 1));//   [v + 1][4] tap pixel offsets
    float *rc = reinterpret_cast<float *>(dimg + GD::BYTES);    // [256] row constant of fuse
    float *dl = rc + 256;                                       // [2][v]
    float *red = dl + 2 * RT * 32;                              // [16]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const long long nn = blockIdx.x;
    const float *loc_n = a.loc + nn * v * 2;
    int im = a.image_idx ? a.image_idx[nn] : a.b - 1;
    const bool img_ok = im >= 0 && im < a.b;
    im = img_ok ? im : 0;
    const long long plane = (long long)a.hf * a.wf;
    const float *fimg = a.feat + im * plane * C;

    float xmin, xmax, ymin, ymax;
    dance::contour_bounds(loc_n, v, red, xmin, xmax, ymin, ymax);
    // the gathered features are split at the image's power-of-two scale (p16.h), folded into the ring
    // vertices' tap weights (the centre's stay as they are: its feature feeds an fp32 GEMV)
    const int f1 = a.absmax ? dance::feature_scale_exp(a.absmax + __builtin_amdgcn_readfirstlane(im) * DANCE_ABSMAX_WORDS)
                            : 0;
    const float fs1 = dance::pow2f(f1), fs1inv = dance::pow2f(-f1);
    for (int vi = t; vi <= v; vi += 256) {
#pragma clang fp contract(off)
        float lx, ly;
        if (vi < v) {
            lx = loc_n[2 * vi];
            ly = loc_n[2 * vi + 1];
            if (a.deloc != DANCE_DELOC_NONE) {
                dl[vi] = dance::deloc_value(lx, xmin, xmax, a.deloc);
                dl[RT * 32 + vi] = dance::deloc_value(ly, ymin, ymax, a.deloc);
            }
        } else {
            lx = (xmin + xmax) * 0.5f;
            ly = (ymin + ymax) * 0.5f;
        }
        dance::Taps tp = dance::bilinear_taps(lx, ly, a.hf, a.wf, a.reflection);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tw[vi * 4 + k] = img_ok ? tp.w[k] * (vi < v ? fs1 : 1.0f) : 0.0f;
            toff[vi * 4 + k] = tp.off[k];
        }
    }
    __syncthreads();

    // ---- centre feature (thread = channel) and the row constant  b + W_c * centre
    {
#pragma clang fp contract(off)
        const float *f0 = fimg + t;
        cf[t] = f0[(long long)toff[v * 4 + 0] * C] * tw[v * 4 + 0] + f0[(long long)toff[v * 4 + 1] * C] * tw[v * 4 + 1] +
                f0[(long long)toff[v * 4 + 2] * C] * tw[v * 4 + 2] + f0[(long long)toff[v * 4 + 3] * C] * tw[v * 4 + 3];
    }
    __syncthreads();
    {
        const int o4 = lane * 4, prt = wave;                    // 4 partial sums of 64 g's per output
        const float *wgp = a.fwg + (long long)prt * 64 * C + o4;
        float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
        for (int k0 = 0; k0 < 64; k0 += 8) {                    // 8 loads in flight per round
            float4 w4[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                w4[kk] = *reinterpret_cast<const float4 *>(wgp + (k0 + kk) * C);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float gk = cf[prt * 64 + k0 + kk];
                fma1(s[0], w4[kk].x, gk);
                fma1(s[1], w4[kk].y, gk);
                fma1(s[2], w4[kk].z, gk);
                fma1(s[3], w4[kk].w, gk);
            }
        }
        *reinterpret_cast<float4 *>(hp + prt * 256 + o4) = float4{s[0], s[1], s[2], s[3]};
    }

    // ---- gather the four 64-channel chunks (16 lanes x float4 per vertex), split into hi / lo
    {
        const int lc = t & 15, vs = t >> 4;
        for (int c0 = 0; c0 < C; c0 += 64) {
            char *xs = img + (c0 >> 6) * G::BYTES;
#pragma unroll 2
            for (int s = vs; s < v; s += 16) {
#pragma clang fp contract(off)
                const float w0 = tw[s * 4 + 0], w1 = tw[s * 4 + 1], w2 = tw[s * 4 + 2], w3 = tw[s * 4 + 3];
                const int ch = c0 + 4 * lc;
                const float4 a0 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 0] * C + ch);
                const float4 a1 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 1] * C + ch);
                const float4 a2 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 2] * C + ch);
                const float4 a3 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 3] * C + ch);
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
        }
    }
    __syncthreads();
    {
        float hs = a.fb[t];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
            hs += hp[pp * 256 + t];
        rc[t] = hs;
    }

    // ---- fuse GEMM: 256 outputs x v vertices, K = 256 (k = 1: no ring taps)
    auto nohook = [](int) __attribute__((always_inline)) {};
    dance::HeadInputScales hs = {1.0f, 1.0f, 1.0f};   // scales of the head block's input (p16.h)
    {
        const int o0 = wave * 64;
        f32x16 acc[2][RT];
        init_acc16<RT, 2>(acc, nullptr, o0, h, 1.0f);
        const long long wbs_b = (long long)C * 16, wts_b = (long long)(C >> 3) * wbs_b;
        const unsigned wlane = (unsigned)(h * C + r) * 16u;
        for (int ch = 0; ch < NCH; ++ch)
            mfma16_core<RT, 2, 4, 4, 4, TERMS>(acc, img + ch * G::BYTES, 0, v,
                                               reinterpret_cast<const char *>(a.fw) + ((long long)(ch * 8) * C + o0) * 16,
                                               wlane, wts_b, wts_b, 2 * wbs_b, 1, 1, r, h, nohook);
        __syncthreads();                        // every wave is done reading the gathered images; rc is complete
        // x = acc / 2^s + row constant (true units, kept in the accumulators) and its largest magnitude
        // over the contour: the head block's input is split at that scale (p16.h)
        const float fw_eff = a.fwinv * fs1inv;
        float xm = 0.0f;
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int row = (o0 >> 3) + ot * 4 + gq;        // channel block 0..31
                const float4 rc4 = *reinterpret_cast<const float4 *>(rc + row * 8 + 4 * h);
                const float rcv[4] = {rc4.x, rc4.y, rc4.z, rc4.w};
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const bool in = rt * 32 + r < v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float x = acc[ot][rt][4 * gq + i] * fw_eff + rcv[i];
                        acc[ot][rt][4 * gq + i] = x;
                        xm = in ? fmaxf(xm, fabsf(x)) : xm;     // (a NaN does not bind: the result is NaN anyway)
                    }
                }
            }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1)
            xm = fmaxf(__shfl_xor(xm, m, 64), xm);
        if (lane == 0)
            red[wave] = xm;
        __syncthreads();
        const float dmax = a.deloc == DANCE_DELOC_DEMIN ? fmaxf(xmax - xmin, ymax - ymin) : 1.0f;
        const float xmax_all = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        hs = dance::head_input_scales(a.hwinv, a.hwdinv, true, dance::bound_scale_exp(xmax_all),
                                      a.deloc != DANCE_DELOC_NONE, dance::bound_scale_exp(dmax));
        const float fs2 = hs.fs;
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int row = (o0 >> 3) + ot * 4 + gq;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int vv = rt * 32 + r;
                    if (vv >= v)
                        continue;
                    h4 hi, lo;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        _Float16 p, q;
                        dance::split_half(acc[ot][rt][4 * gq + i] * fs2, p, q);
                        hi[i] = p;
                        lo[i] = q;
                    }
                    char *d = img + (row >> 3) * G::BYTES + ((row & 7) * G::VP + vv) * 16 + 8 * h;
                    *reinterpret_cast<h4 *>(d) = hi;
                    *reinterpret_cast<h4 *>(d + G::PLANE) = lo;
                }
            }
    }
    const float fs2 = hs.fs, ds2 = hs.ds;
    // de_location as one K step (head16.hip): word (block kb, vertex vi) = entries 8 kb .. 8 kb + 7 of
    // [dx(vi - 4 .. vi + 4), dy(vi - 4 .. vi + 4), 0 x 14]
    if (a.deloc != DANCE_DELOC_NONE)
        for (int i = t; i < 4 * v; i += 256) {
            const int kb = i / v, vi = i - kb * v;
            h8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kb * 8 + e;
                if (k < 18) {
                    const int ch = k >= 9, tap = k - 9 * ch;
                    _Float16 p, q;
                    dance::split_half(dl[ch * RT * 32 + ring16(vi, tap - 4, v)] * ds2, p, q);
                    hi[e] = p;
                    lo[e] = q;
                }
            }
            *reinterpret_cast<h8 *>(dimg + (kb * GD::VP + vi) * 16) = hi;
            *reinterpret_cast<h8 *>(dimg + GD::PLANE + (kb * GD::VP + vi) * 16) = lo;
        }
    __syncthreads();

    // ---- head block on 16-wide vertex tiles: CircConv k = 9 over x, one K step for de_location,
    //      ReLU, BatchNorm
    const int c16 = lane & 15, q4 = lane >> 4;
    const int o0 = wave * 32;
    f32x4 acc[2][CT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float4 b4 = *reinterpret_cast<const float4 *>(a.hbs + o0 + 16 * i + 4 * q4);
        const f32x4 bv = {b4.x * fs2, b4.y * fs2, b4.z * fs2, b4.w * fs2};
#pragma unroll
        for (int j = 0; j < CT; ++j)
            acc[i][j] = bv;
    }
    if (a.deloc != DANCE_DELOC_NONE) {
        h8 ah[2], al[2];
        const char *wdp = reinterpret_cast<const char *>(a.hwd) + (unsigned)(q4 * 128 + o0 + c16) * 16u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const h8 *>(wdp + i * 256);
            if constexpr (TERMS == 3)
                al[i] = *reinterpret_cast<const h8 *>(wdp + 4 * 128 * 16 + i * 256);
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const char *bq = dimg + (unsigned)(q4 * GD::VP + min(j * 16 + c16, v - 1)) * 16u;
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
    const int cin = a.deloc != DANCE_DELOC_NONE ? C + 32 : C;
    const long long wbs_b = 128ll * 16, wts_b = (long long)(cin >> 3) * wbs_b, wps_b = 9 * wts_b;
    const unsigned wlane = (unsigned)(q4 * 128 + c16) * 16u;
    constexpr int NB = (2 * CT) % 4 == 0 && CT >= 3 ? 4 : 2;
    for (int ch = 0; ch < NCH; ++ch)
        mfma16_core_sw<RT, CT, 2, 2, NB, TERMS>(acc, img + ch * G::BYTES, 0, v,
                                                reinterpret_cast<const char *>(a.hw) + ((long long)(ch * 8) * 128 + o0) * 16,
                                                wlane, wps_b, wts_b, 4 * wbs_b, 9, 1, c16, q4, 0, nohook);
    _Float16 *yslot = a.y + nn * a.y_stride;
    const float hwinv = hs.inv;
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
        const int ob = o0 + 16 * i2 + 4 * q4;
        const float4 sc4 = *reinterpret_cast<const float4 *>(a.hscale + ob);
        const float4 sh4 = *reinterpret_cast<const float4 *>(a.hshift + ob);
        const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int vv = j * 16 + c16;
            if (vv >= v)
                continue;
            float val[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                val[i] = relu_nan(acc[i2][j][i] * hwinv) * sc[i] + sh[i];
            dance::store_p16(yslot, v, ob >> 3, vv, (ob >> 2) & 1, val);
        }
    }
}

}  // namespace

extern "C" int dance_init_head16(const float *feat, const float *loc, const int32_t *image_idx,
                                 const void *fuse_w16, float fuse_w_inv_scale, const float *fuse_wg,
                                 const float *fuse_bias, const void *head_w16, const void *head_wd16,
                                 const float *head_bias_scaled,
                                 float head_w_inv_scale, float head_wd_inv_scale, const float *head_scale, const float *head_shift,
                                 void *y16, int64_t y_stride, int n, int v, int b, int hf, int wf, int c,
                                 int padding_mode, int deloc_type, int mfma_terms, const uint32_t *feat_absmax,
                                 void *stream)
{
    DANCE_REQUIRE(mfma_terms == 3 || mfma_terms == 1);
    DANCE_REQUIRE(n >= 0 && v >= 16 && v <= 64 && v % 4 == 0);
    DANCE_REQUIRE(c == 256);
    DANCE_REQUIRE(b >= 1 && hf >= 1 && wf >= 1 && (long long)hf * wf < (1ll << 30));
    DANCE_REQUIRE(padding_mode == DANCE_PAD_ZEROS || padding_mode == DANCE_PAD_REFLECTION);
    DANCE_REQUIRE(deloc_type >= DANCE_DELOC_DERANGE && deloc_type <= DANCE_DELOC_NONE);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(feat && loc && fuse_w16 && fuse_wg && fuse_bias && head_w16 && head_wd16 && head_bias_scaled &&
                  head_scale && head_shift && y16);
    DANCE_REQUIRE(dance::is_pow2_scale(head_w_inv_scale) && dance::is_pow2_scale(head_wd_inv_scale));
    InitHead16Args a;
    a.feat = feat; a.loc = loc; a.image_idx = image_idx; a.absmax = feat_absmax; a.b = b; a.hf = hf; a.wf = wf;
    a.reflection = padding_mode == DANCE_PAD_REFLECTION; a.deloc = deloc_type;
    a.fw = (const h8 *)fuse_w16; a.fwinv = fuse_w_inv_scale; a.fwg = fuse_wg; a.fb = fuse_bias;
    a.hw = (const h8 *)head_w16; a.hwd = (const h8 *)head_wd16; a.hbs = head_bias_scaled; a.hscale = head_scale; a.hshift = head_shift;
    a.hwinv = head_w_inv_scale; a.hwdinv = head_wd_inv_scale; a.y = (_Float16 *)y16; a.y_stride = y_stride; a.v = v;
    constexpr int RT = 2;
    const size_t lds = (size_t)4 * Geo<RT, 8>::BYTES + Geo<RT, 4>::BYTES + (256 + 2 * RT * 32 + 16) * sizeof(float);
    int rc = 0;
#define INIT_LAUNCH(CTV, T)                                                                        \
    do {                                                                                           \
        if (!(rc = dance::allow_lds((const void *)k_init_head16<RT, CTV, T>, lds, __func__)))      \
            hipLaunchKernelGGL((k_init_head16<RT, CTV, T>), dim3(n), dim3(256), lds, (hipStream_t)stream, a); \
    } while (0)
    if (v <= 48 && mfma_terms == 1)
        INIT_LAUNCH(3, 1);
    else if (v <= 48)
        INIT_LAUNCH(3, 3);
    else if (mfma_terms == 1)
        INIT_LAUNCH(4, 1);
    else
        INIT_LAUNCH(4, 3);
#undef INIT_LAUNCH
    if (rc)
        return rc;
    return launch_check(__func__);
}
