// tail16.hip -- the end of a _SnakeNet in the fast precision mode, one launch:
//   global state = max over the ring of the fusion output (reduced here over the parts
//   k_fuse16w wrote)  ->  row constant of prediction[0] (its 256 global-state columns, a
//   256 x 256 GEMV per contour, snake_head.py:272-282)  ->  relu(prediction[0])  ->
//   prediction[2] 256 -> 64 + ReLU (split-half f16 MFMA)  ->  prediction[4] 64 -> 2 (fp32)  ->
//   tanh * wh, attention, loc + off, clip_locations (snake_head.py:1702-1727).
// The (n, 64, v) fp32 block between prediction[2] and prediction[4] never goes to HBM.  p1 arrives at
// the net's prediction scale 2^a_p (p16.h "Activation scales"); wg, pb, b1s and winv1 are packed
// with that scale, so nothing here knows about it.
//
// One workgroup (4 waves) takes GC consecutive contours: the GEMV runs for all of them at once
// (each 16-byte weight load feeds GC contours -- the 256 KB matrix would otherwise be re-read
// from L2 per contour), then contour by contour the 256-channel input is staged in chunks of
// 64 channels: P16 hi/lo words -> fp32 -> + row constant -> ReLU -> split again -> LDS image;
// the MFMA loop has no ring taps (k = 1).  Wave w owns output tile w & 1 (32 of the 64
// channels) and half of the ring's 32-wide vertex tiles.
#include "common.h"
#include "mfma16.h"
#include "p16.h"
#include "sampling.h"

using dance::launch_check;

namespace {

struct Tail16Args {
    const _Float16 *p1;           // (n, 2 slots) P16: prediction[0]'s state part, no row constant, no ReLU
    long long p1_stride, p1_slot; // halfs
    const float *gpart;           // (n, parts, 256) partial maxima of the fusion output
    int parts;
    const float *wg, *pb;         // (256, 256) [g][o], (256)
    const h8 *w1;                 // prediction[2] packed [plane][1][32][64] words
    const float *b1s;             // (64) bias * 2^s
    float winv1;
    const float *w3, *b3;         // prediction[4]: (64, 2), (2)
    const float *loc, *whs, *att; // (n, v, 2), (n, 2) or null, (n, v) or null
    float *out;                   // (n, v, 2)
    int v, clip_first;
    float clip_x, clip_y;
};

template <int RT, int TERMS, int GC>
__global__ __launch_bounds__(256, 2) void k_tail16(Tail16Args a, int n)
{
    constexpr int CBC = 8, NCH = 256 / (CBC * 8), NSTEP = CBC / 2, TH = (RT + 1) / 2;
    using G = Geo<RT, CBC>;
    constexpr int CPR = (RT + 1) / 2;           // 64-lane passes along a row
    constexpr int NWORD = 2 * CPR;              // words (of 8 channels x 1 vertex) per thread and chunk
    static_assert(4 * GC * 256 * 4 <= G::BYTES, "the GEMV partial sums borrow the chunk image");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *xs = smem;                                            // chunk image | GEMV partials | p2 block
    float *p2s = reinterpret_cast<float *>(smem);               // [64][VP]
    float *hp = reinterpret_cast<float *>(smem);                // [4][GC][256]
    float *gs = reinterpret_cast<float *>(smem + G::BYTES);     // [GC][256]
    float *hc = gs + GC * 256;                                  // [GC][256]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int ot = wave & 1, vh = wave >> 1;
    const int v = a.v;
    const long long nb = (long long)blockIdx.x * GC;
    const int gcn = (int)min((long long)GC, n - nb);

    // ---- global state and row constant of the GC contours
#pragma unroll
    for (int g = 0; g < GC; ++g) {
        const long long nn = nb + min(g, gcn - 1);
        const float *gp = a.gpart + nn * a.parts * 256 + t;
        float m = gp[0];
        for (int p = 1; p < a.parts; ++p)
            m = dance::nan_max(gp[p * 256], m);
        gs[g * 256 + t] = m;
    }
    __syncthreads();
    {
        const int o4 = (t & 63) * 4, prt = t >> 6;              // 4 partial sums of 64 g's per output
        const float *wgp = a.wg + (long long)prt * 64 * 256 + o4;
        float s[GC][4];
#pragma unroll
        for (int g = 0; g < GC; ++g)
            s[g][0] = s[g][1] = s[g][2] = s[g][3] = 0.0f;
#pragma unroll 1
        for (int k0 = 0; k0 < 64; k0 += 8) {                    // 8 loads in flight per round
            float4 w4[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                w4[kk] = *reinterpret_cast<const float4 *>(wgp + (k0 + kk) * 256);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int g = 0; g < GC; ++g) {
                    const float gk = gs[g * 256 + prt * 64 + k0 + kk];
                    fma1(s[g][0], w4[kk].x, gk);
                    fma1(s[g][1], w4[kk].y, gk);
                    fma1(s[g][2], w4[kk].z, gk);
                    fma1(s[g][3], w4[kk].w, gk);
                }
        }
#pragma unroll
        for (int g = 0; g < GC; ++g)
            *reinterpret_cast<float4 *>(hp + (prt * GC + g) * 256 + o4) = float4{s[g][0], s[g][1], s[g][2], s[g][3]};
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GC; ++g) {
        float hs = a.pb[t];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
            hs += hp[(pp * GC + g) * 256 + t];
        hc[g * 256 + t] = hs;
    }
    __syncthreads();

    // ---- contour by contour
    const unsigned wlane = (unsigned)(h * 64 + ot * 32 + r) * 16u;
    const long long wps_b = 32ll * 64 * 16;                     // plane stride of w1 in bytes
    unsigned pl[TH];                                            // this lane's word in each of its vertex tiles
#pragma unroll
    for (int j = 0; j < TH; ++j)
        pl[j] = (unsigned)(h * G::VP + min((vh * TH + j) * 32 + r, v - 1)) * 16u;

    for (int g = 0; g < gcn; ++g) {
        const long long nn = nb + g;
        const char *p1n = reinterpret_cast<const char *>(a.p1 + nn * a.p1_stride);
        const float *hcg = hc + g * 256;
        // word j of a chunk: row (channel block) wave + 4 * (j / CPR), vertices (j % CPR) * 64 + lane
        u32x4 rh[NWORD], rl[NWORD];
        auto load_chunk = [&](int c) __attribute__((always_inline)) {
            const char *src = p1n + (long long)(c >> 1) * a.p1_slot * 2 + (long long)((c & 1) * CBC) * v * 16;
#pragma unroll
            for (int j = 0; j < NWORD; ++j) {
                const int row = wave + 4 * (j / CPR);
                const unsigned col = min((j % CPR) * 64 + lane, v - 1);
                const char *q = src + (long long)row * v * 16 + col * 16u;
                rh[j] = *reinterpret_cast<const u32x4 *>(q);
                rl[j] = *reinterpret_cast<const u32x4 *>(q + 256ll * v);
            }
        };
        load_chunk(0);
        f32x16 acc[1][TH];
        init_acc16<TH, 1>(acc, a.b1s, ot * 32, h, 1.0f);
        for (int c = 0; c < NCH; ++c) {
            // this chunk's weights (L2): 4 steps x hi/lo, in flight while the image is built
            h8 ah[NSTEP], al[NSTEP];
            const char *wc = reinterpret_cast<const char *>(a.w1) + (long long)(c * CBC) * 64 * 16 + wlane;
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                ah[s] = *reinterpret_cast<const h8 *>(wc + s * 2 * 64 * 16);
                if constexpr (TERMS == 3)
                    al[s] = *reinterpret_cast<const h8 *>(wc + wps_b + s * 2 * 64 * 16);
            }
            // relu(prediction[0]) = relu(state part + row constant), split again into hi / lo
#pragma unroll
            for (int j = 0; j < NWORD; ++j) {
                const int row = wave + 4 * (j / CPR);
                const unsigned col = min((j % CPR) * 64 + lane, v - 1);
                const float4 ha = *reinterpret_cast<const float4 *>(hcg + c * 64 + row * 8);
                const float4 hb = *reinterpret_cast<const float4 *>(hcg + c * 64 + row * 8 + 4);
                const float hv[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
                const h8 xh = __builtin_bit_cast(h8, rh[j]), xl = __builtin_bit_cast(h8, rl[j]);
                h8 nh, nl;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float x = relu_nan(((float)xh[i] + (float)xl[i]) + hv[i]);
                    _Float16 p, q;
                    dance::split_half(x, p, q);
                    nh[i] = p;
                    nl[i] = q;
                }
                char *d = xs + (row * G::VP + col) * 16u;
                *reinterpret_cast<h8 *>(d) = nh;
                *reinterpret_cast<h8 *>(d + G::PLANE) = nl;
            }
            __syncthreads();
            if (c + 1 < NCH)
                load_chunk(c + 1);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s)
#pragma unroll
                for (int j = 0; j < TH; ++j) {
                    const char *bq = xs + pl[j] + s * 2 * G::VP * 16;
                    const h8 bh = *reinterpret_cast<const h8 *>(bq);
                    if constexpr (TERMS == 3) {
                        const h8 bl = *reinterpret_cast<const h8 *>(bq + G::PLANE);
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh, acc[0][j], 0, 0, 0);
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl, acc[0][j], 0, 0, 0);
                    }
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh, acc[0][j], 0, 0, 0);
                }
            __syncthreads();                    // every wave is done reading the image
        }
        // relu(prediction[2]) as an fp32 block [64][VP] over the image
#pragma unroll
        for (int j = 0; j < TH; ++j) {
            const int vv = (vh * TH + j) * 32 + r;
            if (vv < v) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        p2s[(ot * 32 + 8 * gq + 4 * h + i) * G::VP + vv] = relu_nan(acc[0][j][4 * gq + i] * a.winv1);
            }
        }
        __syncthreads();
        // prediction[4] + offset update, one lane per vertex (the op order of k_offset_update)
        for (int vi = t; vi < v; vi += 256) {
            const long long e = nn * v + vi;
            float z0 = a.b3[0], z1 = a.b3[1];
#pragma unroll 8
            for (int k = 0; k < 64; ++k) {
                const float x = p2s[k * G::VP + vi];
                z0 = fmaf(a.w3[2 * k], x, z0);
                z1 = fmaf(a.w3[2 * k + 1], x, z1);
            }
            if (a.whs != nullptr) {
                z0 = tanhf(z0) * a.whs[nn * 2];
                z1 = tanhf(z1) * a.whs[nn * 2 + 1];
            }
            if (a.att != nullptr) {
                const float sa = a.att[e];
                z0 *= sa;
                z1 *= sa;
            }
            float x = a.loc[e * 2] + z0, y = a.loc[e * 2 + 1] + z1;
            if (a.clip_first && nn == 0) {
                x = x < 0.0f ? 0.0f : (x > a.clip_x ? a.clip_x : x);    // NaN stays NaN, like clamp_
                y = y < 0.0f ? 0.0f : (y > a.clip_y ? a.clip_y : y);
            }
            a.out[e * 2] = x;
            a.out[e * 2 + 1] = y;
        }
        __syncthreads();                        // the block is free for the next contour's image
    }
}

template <int RT, int GC>
int launch_tail(const Tail16Args &a, int n, int terms, hipStream_t s)
{
    const size_t lds = (size_t)Geo<RT, 8>::BYTES + (size_t)2 * GC * 256 * sizeof(float);
    const dim3 grid(dance::ceil_div(n, GC));
    int rc;
    if (terms == 1) {
        if ((rc = dance::allow_lds((const void *)k_tail16<RT, 1, GC>, lds, "dance_tail16")))
            return rc;
        hipLaunchKernelGGL((k_tail16<RT, 1, GC>), grid, dim3(256), lds, s, a, n);
    } else {
        if ((rc = dance::allow_lds((const void *)k_tail16<RT, 3, GC>, lds, "dance_tail16")))
            return rc;
        hipLaunchKernelGGL((k_tail16<RT, 3, GC>), grid, dim3(256), lds, s, a, n);
    }
    return launch_check("dance_tail16");
}

template <int GC>
int launch_tail_rt(const Tail16Args &a, int n, int terms, hipStream_t s)
{
    const int v = a.v;
    if (v <= 64)
        return launch_tail<2, GC>(a, n, terms, s);
    if (v <= 128)
        return launch_tail<4, GC>(a, n, terms, s);
    if (v <= 224)
        return launch_tail<7, GC>(a, n, terms, s);
    return launch_tail<8, GC>(a, n, terms, s);
}

}  // namespace

extern "C" int dance_tail16(const void *p1, int64_t p1_stride, int64_t p1_slot_stride, const float *gpart,
                            int parts, const float *wg, const float *pred0_bias, const void *w1_16,
                            const float *b1_scaled, float w1_inv_scale, const float *w3, const float *b3,
                            const float *loc, const float *whs, const float *att, float *out, int n, int v,
                            int clip_first, float clip_x_max, float clip_y_max, int mfma_terms,
                            const dance_policy *policy, void *stream)
{
    DANCE_REQUIRE(mfma_terms == 3 || mfma_terms == 1);
    DANCE_REQUIRE(n >= 0 && v >= 16 && v <= 256 && parts >= 1 && parts <= 8);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(p1 && gpart && wg && pred0_bias && w1_16 && b1_scaled && w3 && b3 && loc && out);
    Tail16Args a;
    a.p1 = (const _Float16 *)p1; a.p1_stride = p1_stride; a.p1_slot = p1_slot_stride; a.gpart = gpart;
    a.parts = parts; a.wg = wg; a.pb = pred0_bias; a.w1 = (const h8 *)w1_16; a.b1s = b1_scaled;
    a.winv1 = w1_inv_scale; a.w3 = w3; a.b3 = b3; a.loc = loc; a.whs = whs; a.att = att; a.out = out;
    a.v = v; a.clip_first = clip_first; a.clip_x = clip_x_max; a.clip_y = clip_y_max;
    // contours per workgroup (they share every load of the row-constant matrix): two from 1024 contours
    // on for rings up to 128 vertices (tail 0.0626 -> 0.0576 ms at 1536 x 128, 0.132 -> 0.122 ms at
    // 3072; longer rings: behind, 0.127 vs 0.121 ms at 1536 x 196, 0.142 vs 0.138 ms at 256), else one.
    // Four was ahead when this kernel was written and is behind since the library is built without
    // packed fp32 forms (0.0816 vs 0.0745 ms at 1600 x 128), three is behind too.
    // policy.tail_group = 1 | 2 | 4 fixes it (0 = this choice).
    const int pg = dance::policy_or_default(policy).tail_group;
    const int group = pg > 0 ? pg : (n >= 1024 && v <= 128 ? 2 : 1);
    if (group >= 4)
        return launch_tail_rt<4>(a, n, mfma_terms, (hipStream_t)stream);
    if (group >= 2)
        return launch_tail_rt<2>(a, n, mfma_terms, (hipStream_t)stream);
    return launch_tail_rt<1>(a, n, mfma_terms, (hipStream_t)stream);
}
