// featprep16.hip -- SURVEY 8f row 1, the step right before the contour path:
// RefineNet.bottom_out = NUM_CONVS x [Conv2d 3x3 (C -> C, pad 1, no bias) -> GroupNorm(32) -> ReLU]
// on the stride-4 feature map (snake_head.py:562-615, used 1816-1820), ending directly in the
// NHWC fp32 layout the gather kernels read (no separate layout pass).
//
// Same arithmetic as snake16.hip: split-half fp16 operands, three f16 MFMAs per product,
// fp32 accumulation (fp32-level accuracy at the 16-bit matrix rate).
//
//   k_nchw_to_p16img : NCHW fp32 -> "P16 image" [b][plane hi/lo][c/8][y][x][8] fp16
//   k_conv3x3_16     : implicit GEMM, one workgroup = 4 rows x 32 pixels x 128 output channels
//                      (4 waves x 32 channels, 16x16x32 MFMA tiles).  Input chunks of 32
//                      channels with their 1-pixel halo (6 rows x 34 pixels) are double-buffered
//                      in LDS; the 9 taps are immediate offsets into that image.  Output: the
//                      raw convolution, fp32, channel-blocked [b][c/16][y][x][16] (every wave
//                      store is 1 KB contiguous), plus per-(image, group) sum / sum of squares
//                      (per-workgroup partial sums, folded in a fixed order: deterministic) for the
//                      GroupNorm that follows.
//   k_gn_relu        : GroupNorm + ReLU from those statistics, written either as the next
//                      convolution's P16 image or as the final NHWC fp32 feature map.
#include "common.h"
#include "p16.h"

using dance::launch_check;

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------- layout in
__global__ __launch_bounds__(256) void k_nchw_to_p16img(const float *__restrict__ src, int c_total, int c_off,
                                                        _Float16 *__restrict__ dst, int c, long long hw)
{
    // grid: (ceil(hw / 256), c / 8, b); thread = one pixel of one 8-channel block
    const long long p = blockIdx.x * 256ll + threadIdx.x;
    if (p >= hw)
        return;
    const int cb = blockIdx.y, b = blockIdx.z, cbt = c >> 3;
    const float *s = src + ((long long)b * c_total + c_off + cb * 8) * hw + p;
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        _Float16 x, y;
        dance::split_half(s[j * hw], x, y);
        hi[j] = x;
        lo[j] = y;
    }
    _Float16 *d = dst + (((long long)b * 2 * cbt + cb) * hw + p) * 8;
    *reinterpret_cast<h8 *>(d) = hi;
    *reinterpret_cast<h8 *>(d + (long long)cbt * hw * 8) = lo;
}

// the same from an NHWC (channels_last) source: what a previous layer of this file, or torch on
// channels_last tensors, hands over -- no NCHW round trip between the layers of a head.
// thread = one pixel of one 8-channel block, the 8-channel blocks fastest (32-byte reads side
// by side along a pixel's channels; the 16-byte writes of 8 neighbouring pixels form a line)
template <bool VEC>
__global__ __launch_bounds__(256) void k_nhwc_to_p16img(const float *__restrict__ src, int c_total, int c_off,
                                                        _Float16 *__restrict__ dst, int c, long long hw)
{
    const int cbt = c >> 3, b = blockIdx.y;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= hw * cbt)
        return;
    const long long p = i / cbt;
    const int cb = (int)(i - p * cbt);
    const float *s = src + ((long long)b * hw + p) * c_total + c_off + cb * 8;
    float x[8];
    if (VEC) {
        const float4 a = *reinterpret_cast<const float4 *>(s), e = *reinterpret_cast<const float4 *>(s + 4);
        x[0] = a.x, x[1] = a.y, x[2] = a.z, x[3] = a.w, x[4] = e.x, x[5] = e.y, x[6] = e.z, x[7] = e.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            x[j] = s[j];
    }
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        _Float16 u, v;
        dance::split_half(x[j], u, v);
        hi[j] = u;
        lo[j] = v;
    }
    _Float16 *d = dst + (((long long)b * 2 * cbt + cb) * hw + p) * 8;
    *reinterpret_cast<h8 *>(d) = hi;
    *reinterpret_cast<h8 *>(d + (long long)cbt * hw * 8) = lo;
}

// ------------------------------------------------------------------- upsample x2 (+ sum)
// out = acc + bilinear_x2(src), NHWC fp32: the last nn.Upsample(scale_factor=2, bilinear,
// align_corners=False) of a semantic-FPN scale head and the sum over the levels that follows it
// (edge_det.py:118-125, 188-194) in one pass.  Source index as torch computes it:
// max(0.5 * (dst + 0.5) - 0.5, 0).  grid (ceil(2w * c/4 / 256), 2h, b); thread = one float4.
__global__ __launch_bounds__(256) void k_upsample2x_add(const float *__restrict__ src, const float *acc,
                                                        float *out, int h, int w, int c4)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * w * c4)
        return;
    const int ox = i / c4, cq = i - ox * c4, oy = blockIdx.y, b = blockIdx.z;
    const float fy = fmaxf(0.5f * ((float)oy + 0.5f) - 0.5f, 0.f), fx = fmaxf(0.5f * ((float)ox + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx, y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float4 *s = reinterpret_cast<const float4 *>(src) + (long long)b * h * w * c4 + cq;
    const float4 v00 = s[((long long)y0 * w + x0) * c4], v01 = s[((long long)y0 * w + x1) * c4];
    const float4 v10 = s[((long long)y1 * w + x0) * c4], v11 = s[((long long)y1 * w + x1) * c4];
    float4 r;
    r.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    r.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    r.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    r.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    const long long o = (((long long)b * 2 * h + oy) * 2 * w + ox) * c4 + cq;
    if (acc) {
        const float4 a = reinterpret_cast<const float4 *>(acc)[o];
        r.x += a.x, r.y += a.y, r.z += a.z, r.w += a.w;
    }
    reinterpret_cast<float4 *>(out)[o] = r;
}

// ---------------------------------------------------------------------------- conv 3x3
struct Conv2dArgs {
    const _Float16 *x;      // P16 image
    const h8 *w;            // [plane][tap ky*3+kx][cin/8][cout] words (ops.pack_conv_weight16)
    float winv;             // 2^-s of the weight scaling
    float *y;               // [b][cout/16][h][w][16] fp32, raw convolution
    float *partial;         // [b][cout/unit][tiles per image][2]: a workgroup's sum and sum of squares over
                            // its 128 pixels of every unit = min(group_ch, 16) channels it computes
                            // (plain stores; k_sum_rows folds them into the GroupNorm statistics)
    int h, w_, cin, cout, group_ch;
};

constexpr int C2_TR = 4, C2_TC = 32;                       // output tile: rows x columns (128 pixels)
constexpr int C2_HR = C2_TR + 2, C2_PW = C2_TC + 2, C2_CB = 4;
constexpr int C2_PLANE = C2_CB * C2_HR * C2_PW * 16;        // bytes per plane of a chunk image
constexpr int C2_BYTES = 2 * C2_PLANE;                      // 33,792
constexpr int C2_WORDS = C2_BYTES / 16;                     // 2112
constexpr int C2_NSTG = (C2_WORDS + 255) / 256;              // staged 16-B loads per thread and chunk (<= 9 taps)
static_assert(C2_NSTG <= 9, "one staged load per tap");
constexpr int C2_TPR = C2_TC / 16;                          // 16-pixel column tiles per tile row

__global__ __launch_bounds__(256, 2) void k_conv3x3_16(Conv2dArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char xs[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c16 = lane & 15, q4 = lane >> 4;
    const int tiles_x = (a.w_ + C2_TC - 1) / C2_TC, rows2 = (a.h + C2_TR - 1) / C2_TR;
    // the output-channel tile is the fast grid index: the workgroups that share an input
    // tile run together and the second one finds it in L2
    const int mtiles = a.cout >> 7;
    const int mt = blockIdx.x % mtiles, bid = blockIdx.x / mtiles;
    const int tx = bid % tiles_x, ty = (bid / tiles_x) % rows2, b = bid / (tiles_x * rows2);
    const int x0 = tx * C2_TC, y0 = ty * C2_TR;
    const int o0 = mt * 128 + wave * 32;
    const long long hw = (long long)a.h * a.w_;
    const int cbt = a.cin >> 3, nch = a.cin >> 5;
    const long long wbs_b = (long long)a.cout * 16, wts_b = (long long)cbt * wbs_b, wps_b = 9 * wts_b;
    const unsigned wlane = (unsigned)(q4 * a.cout + c16) * 16u;

    // this thread's share of a chunk image: words t, t + 256, ...; image order = LDS order
    // [plane][cb][halo row][halo column]; out-of-map pixels (zero padding) read a clamped
    // address and are zeroed after the load
    unsigned soff[C2_NSTG];
    unsigned sval = 0;
#pragma unroll
    for (int k = 0; k < C2_NSTG; ++k) {
        const int e = min(t + 256 * k, C2_WORDS - 1);
        const int plane = e / (C2_WORDS / 2), r1 = e - plane * (C2_WORDS / 2);
        const int cb = r1 / (C2_HR * C2_PW), r2 = r1 - cb * (C2_HR * C2_PW);
        const int hr = r2 / C2_PW, px = r2 - hr * C2_PW;
        const int yy = y0 + hr - 1, xx = x0 + px - 1;
        const bool ok = t + 256 * k < C2_WORDS && yy >= 0 && yy < a.h && xx >= 0 && xx < a.w_;
        sval |= ok ? 1u << k : 0u;
        soff[k] = (unsigned)((((long long)plane * cbt + cb) * hw + (long long)min(max(yy, 0), a.h - 1) * a.w_ +
                              min(max(xx, 0), a.w_ - 1)) * 16);
    }
    const char *ximg = reinterpret_cast<const char *>(a.x) + (long long)b * 2 * cbt * hw * 16;
    const long long chunk_b = (long long)C2_CB * hw * 16;
    u32x4 sreg[C2_NSTG];
#define C2_STAGE(K, SRC)                                                                         \
    do {                                                                                          \
        u32x4 v_ = *reinterpret_cast<const u32x4 *>((SRC) + soff[K]);                             \
        const u32x4 z_ = {0u, 0u, 0u, 0u};                                                        \
        sreg[K] = (sval >> (K)) & 1u ? v_ : z_;                                                   \
    } while (0)
#define C2_COMMIT(LDS0)                                                                           \
    do {                                                                                          \
        _Pragma("unroll") for (int k_ = 0; k_ < C2_NSTG; ++k_)                                    \
            if (t + 256 * k_ < C2_WORDS)                                                          \
                *reinterpret_cast<u32x4 *>(xs + (LDS0) + (t + 256 * k_) * 16) = sreg[k_];         \
    } while (0)
    dance::static_for<0, C2_NSTG>([&](auto k) __attribute__((always_inline)) { C2_STAGE(decltype(k)::value, ximg); });
    C2_COMMIT(0);
    __syncthreads();

    f32x4 acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char *wbase = reinterpret_cast<const char *>(a.w) + (long long)o0 * 16;
    h8 ah[3][2], al[3][2];
#define C2_LOAD_A(SLOT, PTR)                                                                      \
    do {                                                                                          \
        const char *pa_ = (PTR);                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                        \
            ah[SLOT][i_] = *reinterpret_cast<const h8 *>(pa_ + wlane + i_ * 256);                 \
            al[SLOT][i_] = *reinterpret_cast<const h8 *>(pa_ + wps_b + wlane + i_ * 256);         \
        }                                                                                         \
    } while (0)
    C2_LOAD_A(0, wbase);
    C2_LOAD_A(1, wbase + wts_b);

    const unsigned lanebase = (unsigned)(q4 * C2_HR * C2_PW + c16) * 16u;
    for (int chunk = 0; chunk < nch; ++chunk) {
        const unsigned cur = (chunk & 1) * C2_BYTES, nxt = ((chunk + 1) & 1) * C2_BYTES;
        const bool more = chunk + 1 < nch;
        const char *ssrc = ximg + (more ? chunk + 1 : chunk) * chunk_b;     // last: valid dummy reads
        const char *wchunk = wbase + (long long)chunk * C2_CB * wbs_b;
        const unsigned pb = lanebase + cur;
        h8 bh[4], bl[4];
        // tile Q = tap * 8 + j: tap (ky, kx), j = (row j / C2_TPR, columns 16 (j % C2_TPR) ..)
#define C2_LOAD_B(Q)                                                                              \
    do {                                                                                          \
        constexpr int tap_ = (Q) / 8, j_ = (Q) % 8;                                               \
        constexpr int imm_ = ((j_ / C2_TPR + tap_ / 3) * C2_PW + 16 * (j_ % C2_TPR) + tap_ % 3) * 16; \
        bh[(Q) % 4] = *reinterpret_cast<const h8 *>(xs + pb + imm_);                              \
        bl[(Q) % 4] = *reinterpret_cast<const h8 *>(xs + pb + imm_ + C2_PLANE);                   \
    } while (0)
        C2_LOAD_B(0);
        C2_LOAD_B(1);
        C2_LOAD_B(2);
        dance::static_for<0, 72>([&](auto qq) __attribute__((always_inline)) {
            constexpr int q = decltype(qq)::value, tap = q / 8, j = q % 8;
            if constexpr (j == 0) {
                // weights two taps ahead (the last two taps fetch the next chunk's first two)
                if constexpr (tap + 2 < 9)
                    C2_LOAD_A((tap + 2) % 3, wchunk + (tap + 2) * wts_b);
                else if (more)
                    C2_LOAD_A((tap + 2) % 3, wchunk + C2_CB * wbs_b + (tap + 2 - 9) * wts_b);
                // one staged activation load per tap, issued after the weight prefetch
                if constexpr (tap < C2_NSTG)
                    C2_STAGE(tap, ssrc);
            }
            if constexpr (q + 3 < 72)
                C2_LOAD_B(q + 3);
            __builtin_amdgcn_sched_barrier(0);
            // an accumulator tile's three products back to back (mfma16.h, mfma16_core_sw)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[tap % 3][i], bh[q % 4], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tap % 3][i], bl[q % 4], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tap % 3][i], bh[q % 4], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
#undef C2_LOAD_B
        // the other buffer was last read for chunk - 1, which every wave left before the
        // barrier that published this chunk: refill it, then publish
        if (more)
            C2_COMMIT(nxt);
        __syncthreads();
    }
#undef C2_LOAD_A
#undef C2_STAGE
#undef C2_COMMIT

    // epilogue: lane = pixel (row j / C2_TPR, column 16 (j % C2_TPR) + c16), channels o0 + 16 i + 4 q4 .. + 3
    const int nblk = a.cout >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float s1 = 0.0f, s2 = 0.0f;
        float *yb = a.y + (((long long)b * nblk + ((o0 >> 4) + i)) * hw) * 16 + 4 * q4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int yy = y0 + j / C2_TPR, xx = x0 + 16 * (j % C2_TPR) + c16;
            if (yy < a.h && xx < a.w_) {
                const f32x4 v = acc[i][j] * a.winv;
                *reinterpret_cast<f32x4 *>(yb + ((long long)yy * a.w_ + xx) * 16) = v;
                s1 += (v[0] + v[1]) + (v[2] + v[3]);
                s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
        }
        // group = group_ch consecutive channels; a lane's 4 channels never straddle groups
        // (group_ch is a multiple of 4): reduce over the lanes that share the group
        const int gch = a.group_ch;
#pragma unroll
        for (int s = 1; s < 16; s <<= 1) {
            s1 += __shfl_xor(s1, s, 64);
            s2 += __shfl_xor(s2, s, 64);
        }
        if (gch >= 8) {                          // q4 pairs (8 channels) share a group
            s1 += __shfl_xor(s1, 16, 64);
            s2 += __shfl_xor(s2, 16, 64);
        }
        if (gch >= 16) {                         // the whole 16-channel row tile shares a group
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
        }
        const bool writer = c16 == 0 && (gch >= 16 ? q4 == 0 : (gch >= 8 ? (q4 & 1) == 0 : true));
        if (writer) {
            const int unit = gch >= 16 ? 16 : gch, u = (o0 + 16 * i + 4 * q4) / unit;
            const int tpi = tiles_x * rows2;
            float *st = a.partial + (((long long)b * (a.cout / unit) + u) * tpi + (bid - b * tpi)) * 2;
            st[0] = s1;
            st[1] = s2;
        }
    }
}

// ---------------------------------------------------------------------------- GN + ReLU
// y blocked fp32 -> GroupNorm(groups, eps) * gamma + beta -> ReLU -> P16 image (next conv)
// or NHWC fp32 (gather input).  thread = one pixel of one 16-channel block, 4 channels each.
struct GnArgs {
    const float *y;
    const double *stats;
    const float *gamma, *beta;
    float eps;
    _Float16 *out16;        // P16 image or null
    float *out_nhwc;        // (b, h, w, c) or null
    uint32_t *absmax;       // with out_nhwc: per-workgroup partial maxima of |output| (common.h), or null
    int c, group_ch;
    long long hw;
};

__global__ __launch_bounds__(256) void k_gn_relu(GnArgs a)
{
    // grid: (ceil(hw * 4 / 256), c / 16, b): thread -> (pixel, quarter of the 16-channel block)
    __shared__ unsigned wm[16];
    const long long e = blockIdx.x * 256ll + threadIdx.x;
    const long long p = min(e >> 2, a.hw - 1);                  // threads past the map redo its last pixel, store nothing
    const bool live = (e >> 2) < a.hw;
    const int qtr = (int)(e & 3);
    const int blk = blockIdx.y, b = blockIdx.z, nblk = a.c >> 4;
    const int ch = blk * 16 + qtr * 4;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(a.y + (((long long)b * nblk + blk) * a.hw + p) * 16 + qtr * 4);
    const int g = ch / a.group_ch;
    const double *st = a.stats + ((long long)b * (a.c / a.group_ch) + g) * 2;
    const double cnt = (double)a.group_ch * (double)a.hw;
    const double mean = st[0] / cnt;
    double var = st[1] / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float mu = (float)mean;
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(a.gamma + ch);
    const f32x4 be = *reinterpret_cast<const f32x4 *>(a.beta + ch);
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = (v[i] - mu) * rstd * ga[i] + be[i];
        r[i] = x > 0.0f ? x : (x != x ? x : 0.0f);
    }
    if (a.absmax != nullptr) {                   // (uniform; holds a barrier) a.absmax: the partials (b, gridDim.y * gridDim.x)
        dance::absmax_stage(dance::absmax_acc(dance::absmax_acc(dance::absmax_acc(dance::absmax_acc(0.0f, r[0]), r[1]), r[2]), r[3]), wm);
        __syncthreads();
        dance::absmax_partial(wm, a.absmax + ((long long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    }
    if (!live)
        return;
    if (a.out_nhwc) {
        *reinterpret_cast<f32x4 *>(a.out_nhwc + ((long long)b * a.hw + p) * a.c + ch) = f32x4{r[0], r[1], r[2], r[3]};
    } else {
        // P16 image [b][plane][c/8][p][8]: this thread's 4 channels are half of one word
        const int cbt = a.c >> 3, cb = ch >> 3, half4 = (ch >> 2) & 1;
        h4 hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            _Float16 x, y;
            dance::split_half(r[i], x, y);
            hi[i] = x;
            lo[i] = y;
        }
        _Float16 *d = a.out16 + (((long long)b * 2 * cbt + cb) * a.hw + p) * 8 + half4 * 4;
        *reinterpret_cast<h4 *>(d) = hi;
        *reinterpret_cast<h4 *>(d + (long long)cbt * a.hw * 8) = lo;
    }
}

// ---------------------------------------------------------------------------- 1x1 convolution
// out (b, k, hw) = bias + w (k, c) . x (b, hw, c) for a few output channels k: the edge predictor's
// last layer (edge_det.py:140-163, applied at :197).  One wave per pixel, a lane holds channels
// 4 lane + 256 i (16-byte loads), products summed per lane in channel order, then a butterfly over
// the wave: a fixed order, so the logits -- and the attention map computed from them -- are
// bit-reproducible (the stock 1x1 convolution is not: a split-K GEMM with atomics).
__global__ __launch_bounds__(256) void k_conv1x1_nhwc(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ bias, float *__restrict__ out,
                                                      long long npix, long long hw, int c, int k)
{
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long p = wave0; p < npix; p += nwaves) {
        const float *xp = x + p * c;
        const long long bi = p / hw, pi = p - bi * hw;
        for (int kk = 0; kk < k; ++kk) {
            const float *wk = w + (long long)kk * c;
            float s = 0.0f;
            for (int ch = 4 * lane; ch < c; ch += 256) {
                const float4 xv = *reinterpret_cast<const float4 *>(xp + ch);
                const float4 wv = *reinterpret_cast<const float4 *>(wk + ch);
                s = fmaf(xv.x, wv.x, s);
                s = fmaf(xv.y, wv.y, s);
                s = fmaf(xv.z, wv.z, s);
                s = fmaf(xv.w, wv.w, s);
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1)
                s += __shfl_xor(s, m, 64);
            if (lane == 0)
                out[(bi * k + kk) * hw + pi] = s + (bias ? bias[kk] : 0.0f);
        }
    }
}

}  // namespace

extern "C" int dance_nchw_to_p16img(const float *src, int c_total, int c_off, void *dst16, int b, int c,
                                    int h, int w, void *stream)
{
    DANCE_REQUIRE(b >= 0 && c >= 8 && c % 8 == 0 && h >= 1 && w >= 1 && c_off >= 0 && c_off + c <= c_total);
    if (b == 0)
        return 0;
    DANCE_REQUIRE(src && dst16 && b <= 65535 && c / 8 <= 65535);
    const long long hw = (long long)h * w;
    hipLaunchKernelGGL(k_nchw_to_p16img, dim3((unsigned)((hw + 255) / 256), c / 8, b), dim3(256), 0,
                       (hipStream_t)stream, src, c_total, c_off, (_Float16 *)dst16, c, hw);
    return launch_check(__func__);
}

extern "C" int dance_nhwc_to_p16img(const float *src, int c_total, int c_off, void *dst16, int b, int c,
                                    int h, int w, void *stream)
{
    DANCE_REQUIRE(b >= 0 && c >= 8 && c % 8 == 0 && h >= 1 && w >= 1 && c_off >= 0 && c_off + c <= c_total);
    if (b == 0)
        return 0;
    DANCE_REQUIRE(src && dst16 && b <= 65535);
    const long long hw = (long long)h * w, items = hw * (c / 8);
    DANCE_REQUIRE((items + 255) / 256 < (1ll << 31));
    const dim3 grid((unsigned)((items + 255) / 256), b);
    if (c_total % 4 == 0 && c_off % 4 == 0 && ((uintptr_t)src & 15) == 0)
        hipLaunchKernelGGL(k_nhwc_to_p16img<true>, grid, dim3(256), 0, (hipStream_t)stream, src, c_total, c_off,
                           (_Float16 *)dst16, c, hw);
    else
        hipLaunchKernelGGL(k_nhwc_to_p16img<false>, grid, dim3(256), 0, (hipStream_t)stream, src, c_total, c_off,
                           (_Float16 *)dst16, c, hw);
    return launch_check(__func__);
}

extern "C" int dance_upsample2x_add(const float *src, const float *acc, float *out, int b, int h, int w, int c,
                                    void *stream)
{
    DANCE_REQUIRE(b >= 0 && h >= 1 && w >= 1 && c >= 4 && c % 4 == 0);
    if (b == 0)
        return 0;
    DANCE_REQUIRE(src && out && b <= 65535 && 2 * h <= 65535);
    DANCE_REQUIRE((((uintptr_t)src | (uintptr_t)acc | (uintptr_t)out) & 15) == 0);
    const long long row = 2ll * w * (c / 4);
    DANCE_REQUIRE(row < (1ll << 31) - 256);
    hipLaunchKernelGGL(k_upsample2x_add, dim3((unsigned)((row + 255) / 256), 2 * h, b), dim3(256), 0,
                       (hipStream_t)stream, src, acc, out, h, w, c / 4);
    return launch_check(__func__);
}

extern "C" int64_t dance_conv3x3_16_partial_floats(int b, int cout, int h, int w, int groups)
{
    if (b <= 0 || cout <= 0 || h <= 0 || w <= 0 || groups <= 0 || cout % groups)
        return 0;
    const int gch = cout / groups, unit = gch >= 16 ? 16 : gch;
    const long long tpi = (long long)((h + C2_TR - 1) / C2_TR) * ((w + C2_TC - 1) / C2_TC);
    return (int64_t)b * (cout / unit) * tpi * 2;
}

extern "C" int dance_conv3x3_16(const void *x16, const void *w16, float w_inv_scale, float *y_blocked,
                                double *stats, float *stats_partial, int b, int cin, int cout, int h, int w,
                                int groups, void *stream)
{
    DANCE_REQUIRE(b >= 0 && h >= 1 && w >= 1);
    DANCE_REQUIRE(cin >= 32 && cin % 32 == 0 && cout >= 128 && cout % 128 == 0);
    DANCE_REQUIRE(groups >= 1 && cout % groups == 0 && (cout / groups) % 4 == 0 &&
                  (cout / groups == 4 || cout / groups == 8 || (cout / groups) % 16 == 0));
    DANCE_REQUIRE((long long)h * w * (cin / 8) * 2 * 16 < (1ll << 32));      // 32-bit staging offsets
    if (b == 0)
        return 0;
    DANCE_REQUIRE(x16 && w16 && y_blocked && stats && stats_partial);
    Conv2dArgs a;
    a.x = (const _Float16 *)x16; a.w = (const h8 *)w16; a.winv = w_inv_scale; a.y = y_blocked;
    a.partial = stats_partial; a.h = h; a.w_ = w; a.cin = cin; a.cout = cout; a.group_ch = cout / groups;
    const long long tpi = (long long)((h + C2_TR - 1) / C2_TR) * ((w + C2_TC - 1) / C2_TC);
    const long long tiles = (long long)b * tpi;
    DANCE_REQUIRE(tiles * (cout / 128) < (1ll << 31));
    const size_t lds = 2 * C2_BYTES;
    if (int rc = dance::allow_lds((const void *)k_conv3x3_16, lds, __func__))
        return rc;
    hipLaunchKernelGGL(k_conv3x3_16, dim3((unsigned)(tiles * (cout / 128))), dim3(256), lds, (hipStream_t)stream, a);
    if (int rc = launch_check(__func__))
        return rc;
    // the statistics: per (image, group) the partials of its units and tiles, contiguous, in a fixed order
    const int gch = cout / groups, unit = gch >= 16 ? 16 : gch;
    return dance::sum_rows_async<2>(stats_partial, (long long)b * groups, (gch / unit) * tpi, stats, (hipStream_t)stream,
                                    __func__);
}

extern "C" int dance_gn_relu(const float *y_blocked, const double *stats, const float *gamma,
                             const float *beta, float eps, int groups, void *out16, float *out_nhwc,
                             uint32_t *absmax, uint32_t *absmax_partial, int b, int c, int h, int w, void *stream)
{
    DANCE_REQUIRE(b >= 0 && c >= 16 && c % 16 == 0 && h >= 1 && w >= 1);
    DANCE_REQUIRE(groups >= 1 && c % groups == 0 && (c / groups) % 4 == 0);
    DANCE_REQUIRE((out16 != nullptr) != (out_nhwc != nullptr));
    if (b == 0)
        return 0;
    DANCE_REQUIRE(y_blocked && stats && gamma && beta && b <= 65535);
    GnArgs a;
    a.y = y_blocked; a.stats = stats; a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.out16 = (_Float16 *)out16; a.out_nhwc = out_nhwc; a.c = c; a.group_ch = c / groups;
    a.hw = (long long)h * w;
    DANCE_REQUIRE(absmax == nullptr || absmax_partial != nullptr);
    a.absmax = out_nhwc && absmax ? absmax_partial : nullptr;
    const dim3 grid((unsigned)((a.hw * 4 + 255) / 256), c / 16, b);
    hipLaunchKernelGGL(k_gn_relu, grid, dim3(256), 0, (hipStream_t)stream, a);
    if (int rc = launch_check(__func__))
        return rc;
    if (a.absmax != nullptr)
        return dance::absmax_reduce_async(absmax_partial, (long long)grid.x * grid.y, absmax, b, (hipStream_t)stream, __func__);
    return launch_check(__func__);
}

extern "C" int dance_conv1x1_nhwc(const float *x, const float *w, const float *bias, float *out, int b, int h, int w_,
                                  int c, int k, void *stream)
{
    DANCE_REQUIRE(b >= 0 && h >= 1 && w_ >= 1 && c >= 4 && c % 4 == 0 && k >= 1 && k <= 1024);
    if (b == 0)
        return 0;
    DANCE_REQUIRE(x && w && out && ((((uintptr_t)x | (uintptr_t)w) & 15) == 0));
    const long long hw = (long long)h * w_, npix = hw * b;
    const long long blocks = (npix + 3) / 4;
    hipLaunchKernelGGL(k_conv1x1_nhwc, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream,
                       x, w, bias, out, npix, hw, c, k);
    return launch_check(__func__);
}
