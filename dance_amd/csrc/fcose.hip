// FCOSE box decode in front of the contour path, gfx950.
//
// Reference: FCOSEOutputs.forward_for_single_feature_map, core/modeling/fcose/
// fcose_outputs.py:444-521 (sigmoid scores x centerness, threshold, `nonzero`, box / extreme
// point decode), called per FPN level from predict_proposals (419-442).  The reference builds
// the full (N, HW, C) score tensor, a boolean mask, `nonzero()` per image and five gathers,
// with two host syncs per image and level; here two HBM-bound passes over the logits do it:
//
//   k_fcose_count: one thread per location, loops over the classes (coalesced along HW in the
//                  reference's own NCHW layout), writes how many classes pass the threshold.
//   (torch.cumsum of those counts on the device: the candidates' output slots, in the
//    reference's `nonzero` order = image, level, location, class.)
//   k_fcose_fill:  locations with a non-zero count recompute their scores and write the
//                  candidates: score, class, location, box (loc -+ reg * stride) and the four
//                  extreme points (ExtremePoints.from_boxes, core/structures/pointset.py:110-130).
//
// Compiled with -ffp-contract=off: the reference's box arithmetic is separate fp32 multiplies
// and adds, and the decoded boxes are bit-identical to it.  The sigmoid is 1 / (1 + exp(-x))
// in fp32 as torch's; exp may differ from torch's in the last bit.
#include "common.h"

namespace {

constexpr int FC_THREADS = 256;

__device__ __forceinline__ float fc_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// the product the reference thresholds / keeps: with_ctr: (sig(cls) * sig(ctr)) > th, else
// sig(cls) > th; the kept value is sig(cls) * sig(ctr) either way (fcose_outputs.py:463-470)
__device__ __forceinline__ bool fc_candidate(float logit, float ctr_sig, float thresh, int with_ctr,
                                             float &value)
{
    const float s = fc_sigmoid(logit);
    value = s * ctr_sig;
    return (with_ctr ? value : s) > thresh;
}

__global__ __launch_bounds__(FC_THREADS) void k_fcose_count(const float *__restrict__ cls,
                                                            const float *__restrict__ ctr,
                                                            int32_t *__restrict__ counts,
                                                            int64_t counts_stride, int64_t lvl_off,
                                                            int c, int hw, float thresh, int with_ctr)
{
    const int loc = blockIdx.x * FC_THREADS + threadIdx.x;
    const int img = blockIdx.y;
    if (loc >= hw)
        return;
    const float cs = fc_sigmoid(ctr[(int64_t)img * hw + loc]);
    const float *p = cls + (int64_t)img * c * hw + loc;
    int cnt = 0;
#pragma unroll 8
    for (int k = 0; k < c; ++k) {
        float v;
        cnt += fc_candidate(p[(int64_t)k * hw], cs, thresh, with_ctr, v) ? 1 : 0;
    }
    counts[(int64_t)img * counts_stride + lvl_off + loc] = cnt;
}

__global__ __launch_bounds__(FC_THREADS) void k_fcose_fill(
    const float *__restrict__ cls, const float *__restrict__ ctr, const float *__restrict__ reg,
    const float *__restrict__ ext, const float *__restrict__ locations,
    const int32_t *__restrict__ counts, const int64_t *__restrict__ cum, int64_t counts_stride,
    int64_t lvl_off, int c, int hw, float stride, float thresh, int with_ctr,
    float *__restrict__ raw, float *__restrict__ scores, int64_t *__restrict__ classes,
    float *__restrict__ out_loc, float *__restrict__ boxes, float *__restrict__ extpts)
{
    const int loc = blockIdx.x * FC_THREADS + threadIdx.x;
    const int img = blockIdx.y;
    if (loc >= hw)
        return;
    const int64_t slot = (int64_t)img * counts_stride + lvl_off + loc;
    const int cnt = counts[slot];
    if (cnt == 0)
        return;
    int64_t o = cum[slot] - cnt;   // cum is the inclusive prefix sum
    const float cs = fc_sigmoid(ctr[(int64_t)img * hw + loc]);
    const float lx = locations[2 * loc], ly = locations[2 * loc + 1];
    const float *r = reg + (int64_t)img * 4 * hw + loc;
    // predict_proposals: r = r * s (fcose_outputs.py:431), then loc -+ r (501-506)
    const float x1 = lx - r[0] * stride;
    const float y1 = ly - r[(int64_t)hw] * stride;
    const float x2 = lx + r[2 * (int64_t)hw] * stride;
    const float y2 = ly + r[3 * (int64_t)hw] * stride;
    float e[8];
    if (ext) {
        const float *q = ext + (int64_t)img * 4 * hw + loc;
        const float w = x2 - x1, h = y2 - y1;
        e[0] = lx + w * q[0];                 e[1] = y1;   // top
        e[2] = x1;  e[3] = ly + h * q[(int64_t)hw];        // left
        e[4] = lx + w * q[2 * (int64_t)hw];   e[5] = y2;   // bottom
        e[6] = x2;  e[7] = ly + h * q[3 * (int64_t)hw];    // right
    }
    const float *p = cls + (int64_t)img * c * hw + loc;
    int written = 0;
    for (int k = 0; k < c && written < cnt; ++k) {
        float v;
        if (!fc_candidate(p[(int64_t)k * hw], cs, thresh, with_ctr, v))
            continue;
        raw[o] = v;
        scores[o] = sqrtf(v);
        classes[o] = k;
        out_loc[2 * o] = lx;
        out_loc[2 * o + 1] = ly;
        boxes[4 * o] = x1;
        boxes[4 * o + 1] = y1;
        boxes[4 * o + 2] = x2;
        boxes[4 * o + 3] = y2;
        if (ext) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                extpts[8 * o + j] = e[j];
        }
        ++o;
        ++written;
    }
}

}  // namespace

extern "C" int dance_fcose_count(const float *cls, const float *ctr, int32_t *counts,
                                 int64_t counts_stride, int64_t lvl_off, int n, int c, int hw,
                                 float thresh, int thresh_with_ctr, void *stream)
{
    DANCE_REQUIRE(n >= 0 && c >= 1 && hw >= 0 && lvl_off >= 0 && lvl_off + hw <= counts_stride);
    DANCE_REQUIRE(n <= 65535);
    if (n == 0 || hw == 0)
        return 0;
    DANCE_REQUIRE(cls && ctr && counts);
    hipLaunchKernelGGL(k_fcose_count, dim3(dance::ceil_div(hw, FC_THREADS), n), dim3(FC_THREADS), 0,
                       (hipStream_t)stream, cls, ctr, counts, counts_stride, lvl_off, c, hw, thresh,
                       thresh_with_ctr);
    return dance::launch_check(__func__);
}

extern "C" int dance_fcose_fill(const float *cls, const float *ctr, const float *reg, const float *ext,
                                const float *locations, const int32_t *counts, const int64_t *cum,
                                int64_t counts_stride, int64_t lvl_off, int n, int c, int hw,
                                float stride, float thresh, int thresh_with_ctr, float *raw,
                                float *scores, int64_t *classes, float *out_loc, float *boxes,
                                float *extpts, void *stream)
{
    DANCE_REQUIRE(n >= 0 && c >= 1 && hw >= 0 && lvl_off >= 0 && lvl_off + hw <= counts_stride);
    DANCE_REQUIRE(n <= 65535);
    if (n == 0 || hw == 0)
        return 0;
    DANCE_REQUIRE(cls && ctr && reg && locations && counts && cum);
    DANCE_REQUIRE(raw && scores && classes && out_loc && boxes && (extpts || !ext));
    hipLaunchKernelGGL(k_fcose_fill, dim3(dance::ceil_div(hw, FC_THREADS), n), dim3(FC_THREADS), 0,
                       (hipStream_t)stream, cls, ctr, reg, ext, locations, counts, cum, counts_stride,
                       lvl_off, c, hw, stride, thresh, thresh_with_ctr, raw, scores, classes, out_loc,
                       boxes, extpts);
    return dance::launch_check(__func__);
}
