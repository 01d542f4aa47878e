// attender.hip -- the attention-map producer that feeds channel 0 of the snake head's input
// (SURVEY.md 8f row 1).  reference: EdgeSnakeFPNHead.attender, core/modeling/edge_snake/
// edge_det.py:66-100, applied at 200-204 to x = 1 - pred_edge, (B,1,H,W) at stride 4:
//   Conv2d(1->32, 3x3, pad 1, no bias) -> GroupNorm(4,32) -> ReLU
//   Conv2d(32->32, 3x3, pad 1, no bias) -> GroupNorm(4,32) -> ReLU
//   Conv2d(32->1, 3x3, pad 1, bias)     -> sigmoid                      -> att (B,1,H,W)
// Small maps, tiny channel counts: fp32 FMA on the vector ALU (9.2 k MAC per pixel in the
// middle layer), no MFMA reshaping.  GroupNorm needs whole-image statistics, so three launches:
//   k_att_stats1  conv1 evaluated per pixel (9 MAC x 32), only its group sums are kept;
//   k_att_conv2   per 32x8 tile: conv1 recomputed with halo -> GN1 -> ReLU into LDS, conv2
//                 with the weights as scalar operands (uniform s_loads of the [c][tap][o]
//                 packing), raw output (B,32,H,W) + group sums;
//   k_att_out     per tile: GN2 -> ReLU of the raw output with halo into LDS, conv3 + bias ->
//                 sigmoid -> the attention plane the gather kernels read.
// Group sums are fp32 per thread -> wave shuffle -> fp64 atomics (as featprep16.hip).
#include "common.h"

using dance::launch_check;

namespace {

constexpr int AT_C = 32, AT_G = 4, AT_TX = 32, AT_TY = 8;
constexpr int AT_HX = AT_TX + 2, AT_HY = AT_TY + 2;             // act tile with halo 1
constexpr int AT_XX = AT_TX + 4, AT_XY = AT_TY + 4;             // x tile with halo 2

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

// (sum, sum of squares) of 32 channel values per lane -> 4 groups -> wave -> fp64 atomics
__device__ __forceinline__ void group_stats(const float (&y)[AT_C], bool valid, double *st, int lane)
{
#pragma unroll
    for (int g = 0; g < AT_G; ++g) {
        float s1 = 0.f, s2 = 0.f;
        if (valid) {
#pragma unroll
            for (int i = 0; i < AT_C / AT_G; ++i) {
                const float v = y[g * (AT_C / AT_G) + i];
                s1 += v;
                s2 = fmaf(v, v, s2);
            }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) {
            atomicAdd(st + 2 * g, (double)s1);
            atomicAdd(st + 2 * g + 1, (double)s2);
        }
    }
}

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

__global__ __launch_bounds__(256) void k_att_stats1(const float *__restrict__ x, const float *__restrict__ w1,
                                                    double *__restrict__ stats1, int h, int w)
{
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < h * w;
    const float *xb = x + (long long)b * h * w;
    float y[AT_C];
    if (valid) {
        const int yy = p / w, xx = p - yy * w;
        float nb[9];
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
    }
    group_stats(y, valid, stats1 + (long long)b * AT_G * 2, threadIdx.x & 63);
}

__global__ __launch_bounds__(256) void k_att_conv2(const float *__restrict__ x, const float *__restrict__ w1,
                                                   const float *__restrict__ g1, const float *__restrict__ b1,
                                                   const double *__restrict__ stats1,
                                                   const float *__restrict__ w2p, float *__restrict__ y2,
                                                   double *__restrict__ stats2, float eps, int h, int w)
{
    __shared__ float xt[AT_XY][AT_XX];
    __shared__ float act[AT_C][AT_HY][AT_HX];
    __shared__ float ga[AT_C], gb[AT_C];
    const int b = blockIdx.z, x0 = blockIdx.x * AT_TX, y0 = blockIdx.y * AT_TY;
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    const float *xb = x + (long long)b * h * w;
    for (int i = t; i < AT_XY * AT_XX; i += 256) {
        const int r = i / AT_XX, c = i - r * AT_XX;
        xt[r][c] = px(xb, h, w, y0 + r - 2, x0 + c - 2);
    }
    if (t < AT_C)
        gn_affine(stats1 + (long long)b * AT_G * 2, g1, b1, (double)(AT_C / AT_G) * h * w, eps, t, ga[t], gb[t]);
    __syncthreads();
    // layer 1 on the halo tile: conv1 -> GN1 -> ReLU; zero outside the image (conv2's padding)
    for (int i = t; i < AT_C * AT_HY * AT_HX; i += 256) {
        const int c = i / (AT_HY * AT_HX), rem = i - c * (AT_HY * AT_HX), r = rem / AT_HX, q = rem - r * AT_HX;
        const int yy = y0 + r - 1, xx = x0 + q - 1;
        float v = 0.f;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k)
                s = fmaf(w1[c * 9 + k], xt[r + k / 3][q + k % 3], s);
            v = fmaxf(fmaf(s, ga[c], gb[c]), 0.f);
        }
        act[c][r][q] = v;
    }
    __syncthreads();
    float acc[AT_C];
#pragma unroll
    for (int o = 0; o < AT_C; ++o)
        acc[o] = 0.f;
#pragma unroll 1
    for (int c = 0; c < AT_C; ++c) {
        float a[9];
#pragma unroll
        for (int k = 0; k < 9; ++k)
            a[k] = act[c][ty + k / 3][tx + k % 3];
        const float *wc = w2p + c * 9 * AT_C;               // uniform: scalar loads
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int o = 0; o < AT_C; ++o)
                acc[o] = fmaf(wc[k * AT_C + o], a[k], acc[o]);
    }
    const int yy = y0 + ty, xx = x0 + tx;
    const bool valid = yy < h && xx < w;
    if (valid) {
        float *dst = y2 + (long long)b * AT_C * h * w + (long long)yy * w + xx;
#pragma unroll
        for (int o = 0; o < AT_C; ++o)
            dst[(long long)o * h * w] = acc[o];
    }
    group_stats(acc, valid, stats2 + (long long)b * AT_G * 2, t & 63);
}

__global__ __launch_bounds__(256) void k_att_out(const float *__restrict__ y2, const float *__restrict__ g2,
                                                 const float *__restrict__ b2, const double *__restrict__ stats2,
                                                 const float *__restrict__ w3, const float *__restrict__ bias3,
                                                 float *__restrict__ att, float eps, int h, int w)
{
    __shared__ float act[AT_C][AT_HY][AT_HX];
    __shared__ float ga[AT_C], gb[AT_C];
    const int b = blockIdx.z, x0 = blockIdx.x * AT_TX, y0 = blockIdx.y * AT_TY;
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    if (t < AT_C)
        gn_affine(stats2 + (long long)b * AT_G * 2, g2, b2, (double)(AT_C / AT_G) * h * w, eps, t, ga[t], gb[t]);
    __syncthreads();
    const float *yb = y2 + (long long)b * AT_C * h * w;
    for (int i = t; i < AT_C * AT_HY * AT_HX; i += 256) {
        const int c = i / (AT_HY * AT_HX), rem = i - c * (AT_HY * AT_HX), r = rem / AT_HX, q = rem - r * AT_HX;
        const int yy = y0 + r - 1, xx = x0 + q - 1;
        float v = 0.f;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w)
            v = fmaxf(fmaf(yb[(long long)c * h * w + (long long)yy * w + xx], ga[c], gb[c]), 0.f);
        act[c][r][q] = v;
    }
    __syncthreads();
    const int yy = y0 + ty, xx = x0 + tx;
    if (yy >= h || xx >= w)
        return;
    float s = bias3[0];
#pragma unroll 4
    for (int c = 0; c < AT_C; ++c)
#pragma unroll
        for (int k = 0; k < 9; ++k)
            s = fmaf(w3[c * 9 + k], act[c][ty + k / 3][tx + k % 3], s);
    att[(long long)b * h * w + (long long)yy * w + xx] = 1.0f / (1.0f + expf(-s));
}

}  // namespace

extern "C" int64_t dance_attender_workspace_bytes(int b, int h, int w)
{
    if (b <= 0 || h <= 0 || w <= 0)
        return 0;
    return (int64_t)b * AT_C * h * w * 4 + (int64_t)2 * b * AT_G * 2 * 8;
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
    double *stats1 = reinterpret_cast<double *>(workspace);
    double *stats2 = stats1 + (long long)b * AT_G * 2;
    float *y2 = reinterpret_cast<float *>(stats2 + (long long)b * AT_G * 2);
    hipError_t e = hipMemsetAsync(stats1, 0, (size_t)2 * b * AT_G * 2 * 8, s);
    if (e != hipSuccess)
        return dance::fail((int)e, __func__, hipGetErrorString(e));
    const dim3 tiles(dance::ceil_div(w, AT_TX), dance::ceil_div(h, AT_TY), b);
    hipLaunchKernelGGL(k_att_stats1, dim3(dance::ceil_div(h * w, 256), b), dim3(256), 0, s, x, w1, stats1, h, w);
    hipLaunchKernelGGL(k_att_conv2, tiles, dim3(256), 0, s, x, w1, gn1_w, gn1_b, stats1, w2_packed, y2, stats2, eps,
                       h, w);
    hipLaunchKernelGGL(k_att_out, tiles, dim3(256), 0, s, y2, gn2_w, gn2_b, stats2, w3, b3, att, eps, h, w);
    return launch_check(__func__);
}
