// sampling.h -- device-side bilinear tap computation shared by the stand-alone gather
// kernel and the fused gather + first-CircConv kernel.
//
// Follows torch's grid_sample(bilinear, align_corners=False) as the reference calls it
// (snake_head.py:1542-1572 reflection padding; dsnake_head.py:678-692 zeros padding):
//   g  = loc / (size_px / 2) - 1          (snake_head.py:1546-1547, size_px = 4 * size_f)
//   f  = (g + 1) * (size_f / 2) - 0.5     (un-normalise, align_corners=False)
//   reflection: reflect about -0.5 and size_f - 0.5, then clip to [0, size_f - 1]
//   taps floor(f), floor(f)+1 with weights (1-t), t; taps outside the map contribute 0.
// Contraction is disabled so the coordinate arithmetic rounds like the separate torch ops
// (the g -> f step amplifies one ulp of g to ~1e-5 feature pixels).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/dance_hip.h"

namespace dance {

struct Taps {
    int off[4];     // pixel index y * wf + x inside the image plane, 0 when invalid
    float w[4];     // bilinear weight, 0 when the tap is outside the map
};

__device__ __forceinline__ float reflect_coord(float f, float size)
{
#pragma clang fp contract(off)
    const float twice_span = size * 2.0f;
    const float a = fabsf(f + 0.5f);
    const float flips = truncf(a / twice_span);
    const float extra = a - flips * twice_span;
    float r = fminf(extra, twice_span - extra) - 0.5f;
    r = fminf(fmaxf(r, 0.0f), size - 1.0f);
    return r;
}

// loc in image pixels; hf, wf feature-map size; reflection != 0 selects reflection padding.
__device__ __forceinline__ Taps bilinear_taps(float lx, float ly, int hf, int wf, int reflection)
{
#pragma clang fp contract(off)
    Taps t;
    const float half_w_px = (float)(wf * 4) / 2.0f;
    const float half_h_px = (float)(hf * 4) / 2.0f;
    const float gx = lx / half_w_px - 1.0f;
    const float gy = ly / half_h_px - 1.0f;
    float fx = (gx + 1.0f) * ((float)wf / 2.0f) - 0.5f;
    float fy = (gy + 1.0f) * ((float)hf / 2.0f) - 0.5f;
    if (reflection) {
        fx = reflect_coord(fx, (float)wf);
        fy = reflect_coord(fy, (float)hf);
    }
    // NaN / infinite coordinates: no tap is valid; the weights stay NaN-free so that the
    // only NaNs downstream are the ones the reference's de_location produces.
    const bool finite = (fx - fx == 0.0f) && (fy - fy == 0.0f);
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float tx = fx - x0f, ty = fy - y0f;
    const float ex = 1.0f - tx, ey = 1.0f - ty;
    // keep the int conversion in range whatever the coordinate is
    const float xc = fminf(fmaxf(x0f, -2.0f), (float)wf + 1.0f);
    const float yc = fminf(fmaxf(y0f, -2.0f), (float)hf + 1.0f);
    const int x0 = finite ? (int)xc : -2, y0 = finite ? (int)yc : -2;
    const float wgt[4] = {ey * ex, ey * tx, ty * ex, ty * tx};   // nw, ne, sw, se
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x0 + (k & 1), y = y0 + (k >> 1);
        const bool in = finite && x >= 0 && x < wf && y >= 0 && y < hf && x0f == xc && y0f == yc;
        t.off[k] = in ? y * wf + x : 0;
        t.w[k] = in ? wgt[k] : 0.0f;
    }
    return t;
}

// NaN-propagating min / max (torch.min / torch.max semantics)
__device__ __forceinline__ float nan_min(float a, float b) { return (a != a || a < b) ? a : b; }
__device__ __forceinline__ float nan_max(float a, float b) { return (a != a || a > b) ? a : b; }

// contour-wide min / max of x and y (NaN-propagating), every thread gets the result.
// red: 16 floats of LDS.  All 256 threads must call.
__device__ __forceinline__ void contour_bounds(const float *__restrict__ loc_n, int v, float *red,
                                               float &xmin, float &xmax, float &ymin, float &ymax)
{
    const int t = threadIdx.x;
    float ax = INFINITY, bx = -INFINITY, ay = INFINITY, by = -INFINITY;
    for (int i = t; i < v; i += blockDim.x) {
        const float2 p = *reinterpret_cast<const float2 *>(loc_n + 2 * i);
        ax = nan_min(p.x, ax);
        bx = nan_max(p.x, bx);
        ay = nan_min(p.y, ay);
        by = nan_max(p.y, by);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        ax = nan_min(__shfl_xor(ax, m, 64), ax);
        bx = nan_max(__shfl_xor(bx, m, 64), bx);
        ay = nan_min(__shfl_xor(ay, m, 64), ay);
        by = nan_max(__shfl_xor(by, m, 64), by);
    }
    const int wave = t >> 6, nw = blockDim.x >> 6;
    if ((t & 63) == 0) {
        red[wave * 4 + 0] = ax;
        red[wave * 4 + 1] = bx;
        red[wave * 4 + 2] = ay;
        red[wave * 4 + 3] = by;
    }
    __syncthreads();
    xmin = red[0]; xmax = red[1]; ymin = red[2]; ymax = red[3];
    for (int i = 1; i < nw; ++i) {
        xmin = nan_min(red[i * 4 + 0], xmin);
        xmax = nan_max(red[i * 4 + 1], xmax);
        ymin = nan_min(red[i * 4 + 2], ymin);
        ymax = nan_max(red[i * 4 + 3], ymax);
    }
}

// de_location of one coordinate (snake_head.py:1587-1610, dsnake_head.py:706-707)
__device__ __forceinline__ float deloc_value(float p, float mn, float mx, int type)
{
#pragma clang fp contract(off)
    if (type == DANCE_DELOC_DERANGE)
        return (p - mn) / (mx - mn);
    if (type == DANCE_DELOC_DEMEAN)
        return 2.0f * (p - mn) / (mx - mn) - 1.0f;
    return p - mn;      // demin
}

}  // namespace dance
