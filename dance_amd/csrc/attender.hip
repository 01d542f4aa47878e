// attender.hip -- the attention-map producer that feeds channel 0 of the snake head's input
// (SURVEY.md 8f row 1).  reference: EdgeSnakeFPNHead.attender, core/modeling/edge_snake/
// edge_det.py:66-100, applied at 200-204 to x = 1 - pred_edge, (B,1,H,W) at stride 4:
//   Conv2d(1->32, 3x3, pad 1, no bias) -> GroupNorm(4,32) -> ReLU
//   Conv2d(32->32, 3x3, pad 1, no bias) -> GroupNorm(4,32) -> ReLU
//   Conv2d(32->1, 3x3, pad 1, bias)     -> sigmoid                      -> att (B,1,H,W)
// Small maps, tiny channel counts: fp32 FMA on the vector ALU (9.2 k MAC per pixel in the
// middle layer), no MFMA reshaping.  GroupNorm needs whole-image statistics, so three launches:
//   k_att_stats1  conv1 evaluated per pixel (9 MAC x 32), only its group sums are kept;
//   k_att_conv2   per 32x16 tile: conv1 recomputed with halo -> GN1 -> ReLU into LDS, conv2
//                 with the weights as scalar operands (uniform s_loads of the [c][tap][o]
//                 packing), raw output (B,32,H,W) + group sums;
//   k_att_out     per 32x16 tile: GN2 -> ReLU of the raw output with halo into LDS, conv3 + bias ->
//                 sigmoid -> the attention plane the gather kernels read.
// Group sums are fp32 per thread -> wave shuffle -> workgroup (LDS) -> one partial per workgroup and value,
// folded in fp64 in a fixed order by k_sum_rows (common.h): no atomics, two forwards give identical bits.
#include "common.h"

using dance::launch_check;

namespace {

constexpr int AT_C = 32, AT_G = 4, AT_TX = 32;
constexpr int AT_HX = AT_TX + 2;                                // act tile with halo 1 (columns)
constexpr int AT_XX = AT_TX + 4;                                // x tile with halo 2 (columns)

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

// Per-lane (sum, sum of squares) of the 4 channel groups, accumulated over a thread's pixels.
struct GroupSums {
    float s1[AT_G], s2[AT_G];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int g = 0; g < AT_G; ++g)
            s1[g] = s2[g] = 0.f;
    }
    __device__ __forceinline__ void add(const float (&y)[AT_C])
    {
#pragma unroll
        for (int g = 0; g < AT_G; ++g)
#pragma unroll
            for (int i = 0; i < AT_C / AT_G; ++i) {
                const float v = y[g * (AT_C / AT_G) + i];
                s1[g] += v;
                s2[g] = fmaf(v, v, s2[g]);
            }
    }
    // lanes -> wave (shuffles) -> workgroup (LDS, red: 4 x 8 floats) -> the workgroup's 8 partials
    // (part: this workgroup's slot of 8 floats)
    __device__ __forceinline__ void flush(float *part, float *red, int t)
    {
        const int lane = t & 63, wave = t >> 6;
#pragma unroll
        for (int g = 0; g < AT_G; ++g) {
            const float a = wave_sum(s1[g]), b = wave_sum(s2[g]);
            if (lane == 0) {
                red[wave * 8 + 2 * g] = a;
                red[wave * 8 + 2 * g + 1] = b;
            }
        }
        __syncthreads();
        if (t < 8)
            part[t] = (red[t] + red[8 + t]) + (red[16 + t] + red[24 + t]);
    }
};

// GroupNorm as a per-channel affine a*x + b from the accumulated sums
__device__ __forceinline__ void gn_affine(const double *st, const float *gamma, const float *beta, double cnt,
                                          float eps, int c, float &a, float &b)
{
    const int g = c / (AT_C / AT_G);
    const double mean = st[2 * g] / cnt;
    double var = st[2 * g + 1] / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    a = gamma[c] * rstd;
    b = beta[c] - (float)mean * a;
}

__device__ __forceinline__ float px(const float *x, int h, int w, int yy, int xx)
{
    return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? x[(long long)yy * w + xx] : 0.f;
}

constexpr int AT_PPT = 8;                                       // pixels per thread in k_att_stats1

__global__ __launch_bounds__(256) void k_att_stats1(const float *__restrict__ x, const float *__restrict__ w1,
                                                    float *__restrict__ part1, int h, int w)
{
    __shared__ float red[32];
    const int b = blockIdx.y;
    const float *xb = x + (long long)b * h * w;
    GroupSums gs;
    gs.clear();
    for (int k = 0; k < AT_PPT; ++k) {
        const int p = (blockIdx.x * AT_PPT + k) * 256 + threadIdx.x;
        if (p >= h * w)
            break;
        const int yy = p / w, xx = p - yy * w;
        float nb[9], y[AT_C];
#pragma unroll
        for (int t = 0; t < 9; ++t)
            nb[t] = px(xb, h, w, yy + t / 3 - 1, xx + t % 3 - 1);
#pragma unroll
        for (int o = 0; o < AT_C; ++o) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                s = fmaf(w1[o * 9 + t], nb[t], s);
            y[o] = s;
        }
        gs.add(y);
    }
    gs.flush(part1 + ((long long)b * gridDim.x + blockIdx.x) * 8, red, threadIdx.x);
}

// conv2 tile: 32 x 16 pixels, two pixels (rows ty and ty + 8) per thread so that every scalar
// weight load feeds two FMAs and the halo is 20 % instead of 33 %; the layer-1 activations go
// through LDS 16 channels at a time (39 KB: three workgroups per CU).
constexpr int AT_TY2 = 16, AT_HY2 = AT_TY2 + 2, AT_XY2 = AT_TY2 + 4, AT_CH = 16;

__global__ __launch_bounds__(256) void k_att_conv2(const float *__restrict__ x, const float *__restrict__ w1,
                                                   const float *__restrict__ g1, const float *__restrict__ b1,
                                                   const double *__restrict__ stats1,
                                                   const float *__restrict__ w2p, float *__restrict__ y2,
                                                   float *__restrict__ part2, float eps, int h, int w)
{
    __shared__ float xt[AT_XY2][AT_XX];
    __shared__ float act[AT_CH][AT_HY2][AT_HX];
    __shared__ float ga[AT_C], gb[AT_C], red[32];
    const int b = blockIdx.z, x0 = blockIdx.x * AT_TX, y0 = blockIdx.y * AT_TY2;
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    const float *xb = x + (long long)b * h * w;
    for (int i = t; i < AT_XY2 * AT_XX; i += 256) {
        const int r = i / AT_XX, c = i - r * AT_XX;
        xt[r][c] = px(xb, h, w, y0 + r - 2, x0 + c - 2);
    }
    if (t < AT_C)
        gn_affine(stats1 + (long long)b * AT_G * 2, g1, b1, (double)(AT_C / AT_G) * h * w, eps, t, ga[t], gb[t]);
    __syncthreads();
    // layer 1 on the halo tile: conv1 -> GN1 -> ReLU; zero outside the image (conv2's padding).
    // A thread keeps the 3x3 input patches of its (up to three) tile positions in registers and
    // walks the channels with uniform (scalar) weights.
    constexpr int NPOS = AT_HY2 * AT_HX, ROUNDS = (NPOS + 255) / 256;
    float nb[ROUNDS][9];
    bool in[ROUNDS];
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        const int pos = min(t + j * 256, NPOS - 1), r = pos / AT_HX, q = pos - r * AT_HX;
        const int yy = y0 + r - 1, xx = x0 + q - 1;
        in[j] = yy >= 0 && yy < h && xx >= 0 && xx < w;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            nb[j][k] = xt[r + k / 3][q + k % 3];
    }
    float acc[2][AT_C];
#pragma unroll
    for (int o = 0; o < AT_C; ++o)
        acc[0][o] = acc[1][o] = 0.f;
#pragma unroll 1
    for (int c0 = 0; c0 < AT_C; c0 += AT_CH) {
        if (c0)
            __syncthreads();                    // the previous 16 channels have been consumed
#pragma unroll 2
        for (int cc = 0; cc < AT_CH; ++cc) {
            const int c = c0 + cc;
            const float a = ga[c], bb = gb[c];
#pragma unroll
            for (int j = 0; j < ROUNDS; ++j) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    s = fmaf(w1[c * 9 + k], nb[j][k], s);
                if (t + j * 256 < NPOS)
                    (&act[cc][0][0])[t + j * 256] = in[j] ? fmaxf(fmaf(s, a, bb), 0.f) : 0.f;
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int cc = 0; cc < AT_CH; ++cc) {
            float a0[9], a1[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                a0[k] = act[cc][ty + k / 3][tx + k % 3];
                a1[k] = act[cc][ty + 8 + k / 3][tx + k % 3];
            }
            const float *wc = w2p + (c0 + cc) * 9 * AT_C;       // uniform: scalar loads
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int o = 0; o < AT_C; ++o) {
                    const float wv = wc[k * AT_C + o];
                    acc[0][o] = fmaf(wv, a0[k], acc[0][o]);
                    acc[1][o] = fmaf(wv, a1[k], acc[1][o]);
                }
        }
    }
    GroupSums gs;
    gs.clear();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int yy = y0 + ty + 8 * p, xx = x0 + tx;
        if (yy < h && xx < w) {
            float *dst = y2 + (long long)b * AT_C * h * w + (long long)yy * w + xx;
#pragma unroll
            for (int o = 0; o < AT_C; ++o)
                dst[(long long)o * h * w] = acc[p][o];
            gs.add(acc[p]);
        }
    }
    gs.flush(part2 + (((long long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8, red, t);
}

__global__ __launch_bounds__(256) void k_att_out(const float *__restrict__ y2, const float *__restrict__ g2,
                                                 const float *__restrict__ b2, const double *__restrict__ stats2,
                                                 const float *__restrict__ w3, const float *__restrict__ bias3,
                                                 float *__restrict__ att, float eps, int h, int w)
{
    // same 32 x 16 tile, two pixels per thread, 16 channels at a time through LDS
    __shared__ float act[AT_CH][AT_HY2][AT_HX];
    __shared__ float ga[AT_C], gb[AT_C];
    const int b = blockIdx.z, x0 = blockIdx.x * AT_TX, y0 = blockIdx.y * AT_TY2;
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    if (t < AT_C)
        gn_affine(stats2 + (long long)b * AT_G * 2, g2, b2, (double)(AT_C / AT_G) * h * w, eps, t, ga[t], gb[t]);
    const float *yb = y2 + (long long)b * AT_C * h * w;
    constexpr int NPOS = AT_HY2 * AT_HX, ROUNDS = (NPOS + 255) / 256;
    long long off[ROUNDS];                      // pixel offset of this thread's tile positions (-1: outside)
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        const int pos = min(t + j * 256, NPOS - 1), r = pos / AT_HX, q = pos - r * AT_HX;
        const int yy = y0 + r - 1, xx = x0 + q - 1;
        off[j] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? (long long)yy * w + xx : -1;
    }
    float s0 = bias3[0], s1 = s0;
#pragma unroll 1
    for (int c0 = 0; c0 < AT_C; c0 += AT_CH) {
        __syncthreads();                        // ga / gb ready; the previous 16 channels consumed
#pragma unroll 4
        for (int cc = 0; cc < AT_CH; ++cc) {
            const int c = c0 + cc;
            const float a = ga[c], bb = gb[c];
#pragma unroll
            for (int j = 0; j < ROUNDS; ++j)
                if (t + j * 256 < NPOS)
                    (&act[cc][0][0])[t + j * 256] =
                        off[j] >= 0 ? fmaxf(fmaf(yb[(long long)c * h * w + off[j]], a, bb), 0.f) : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int cc = 0; cc < AT_CH; ++cc)
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float wv = w3[(c0 + cc) * 9 + k];
                s0 = fmaf(wv, act[cc][ty + k / 3][tx + k % 3], s0);
                s1 = fmaf(wv, act[cc][ty + 8 + k / 3][tx + k % 3], s1);
            }
    }
    const int xx = x0 + tx;
    if (xx < w) {
        if (y0 + ty < h)
            att[(long long)b * h * w + (long long)(y0 + ty) * w + xx] = 1.0f / (1.0f + expf(-s0));
        if (y0 + ty + 8 < h)
            att[(long long)b * h * w + (long long)(y0 + ty + 8) * w + xx] = 1.0f / (1.0f + expf(-s1));
    }
}

}  // namespace

namespace {
struct AttWorkspace {
    double *stats1, *stats2;
    float *y2, *part1, *part2;
    long long wg1, tiles2;
    int64_t bytes;
};

AttWorkspace att_carve(void *base, int b, int h, int w)
{
    AttWorkspace a;
    a.wg1 = dance::ceil_div(h * w, 256 * AT_PPT);
    a.tiles2 = (long long)dance::ceil_div(w, AT_TX) * dance::ceil_div(h, AT_TY2);
    char *p = (char *)base;
    a.stats1 = (double *)p;
    a.stats2 = a.stats1 + (long long)b * AT_G * 2;
    a.y2 = (float *)(a.stats2 + (long long)b * AT_G * 2);
    a.part1 = a.y2 + (long long)b * AT_C * h * w;
    a.part2 = a.part1 + (long long)b * a.wg1 * 8;
    a.bytes = (int64_t)((char *)(a.part2 + (long long)b * a.tiles2 * 8) - p);
    return a;
}
}  // namespace

extern "C" int64_t dance_attender_workspace_bytes(int b, int h, int w)
{
    if (b <= 0 || h <= 0 || w <= 0)
        return 0;
    return att_carve(nullptr, b, h, w).bytes;
}

extern "C" int dance_attender(const float *x, const float *w1, const float *gn1_w, const float *gn1_b,
                              const float *w2_packed, const float *gn2_w, const float *gn2_b, const float *w3,
                              const float *b3, float eps, float *att, void *workspace, int64_t workspace_bytes,
                              int b, int h, int w, void *stream)
{
    DANCE_REQUIRE(b >= 0 && h >= 1 && w >= 1 && (long long)h * w < (1ll << 28));
    if (b == 0)
        return 0;
    DANCE_REQUIRE(b <= 65535);
    DANCE_REQUIRE(x && w1 && gn1_w && gn1_b && w2_packed && gn2_w && gn2_b && w3 && b3 && att && workspace);
    DANCE_REQUIRE(workspace_bytes >= dance_attender_workspace_bytes(b, h, w));
    hipStream_t s = (hipStream_t)stream;
    const AttWorkspace ws = att_carve(workspace, b, h, w);
    hipLaunchKernelGGL(k_att_stats1, dim3((unsigned)ws.wg1, b), dim3(256), 0, s, x, w1, ws.part1, h, w);
    if (int rc = dance::sum_rows_async<8>(ws.part1, b, ws.wg1, ws.stats1, s, __func__))
        return rc;
    const dim3 tiles2(dance::ceil_div(w, AT_TX), dance::ceil_div(h, AT_TY2), b);
    hipLaunchKernelGGL(k_att_conv2, tiles2, dim3(256), 0, s, x, w1, gn1_w, gn1_b, ws.stats1, w2_packed, ws.y2, ws.part2,
                       eps, h, w);
    if (int rc = dance::sum_rows_async<8>(ws.part2, b, ws.tiles2, ws.stats2, s, __func__))
        return rc;
    hipLaunchKernelGGL(k_att_out, tiles2, dim3(256), 0, s, ws.y2, gn2_w, gn2_b, ws.stats2, w3, b3, att, eps, h, w);
    return launch_check(__func__);
}
