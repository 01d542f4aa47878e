// FCOSE box decode in front of the contour path, gfx950.
//
// Reference: FCOSEOutputs.forward_for_single_feature_map, core/modeling/fcose/
// fcose_outputs.py:444-521 (sigmoid scores x centerness, threshold, `nonzero`, box / extreme
// point decode), called per FPN level from predict_proposals (419-442).  The reference builds
// the full (N, HW, C) score tensor, a boolean mask, `nonzero()` per image and five gathers,
// with two host syncs per image and level; here two HBM-bound passes over the logits do it:
//
//   k_fcose_count: every level in one launch; one thread per location loops over the classes
//                  (coalesced along HW in the reference's own NCHW layout) and writes how
//                  many pass the threshold.
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

// every level of the pyramid in one launch: blockIdx.x walks the levels' 256-location blocks
struct LevelTable {
    const float *cls[DANCE_FCOSE_MAX_LEVELS], *ctr[DANCE_FCOSE_MAX_LEVELS], *reg[DANCE_FCOSE_MAX_LEVELS];
    const float *ext[DANCE_FCOSE_MAX_LEVELS], *loc[DANCE_FCOSE_MAX_LEVELS];
    int64_t off[DANCE_FCOSE_MAX_LEVELS];       // first slot of the level in an image's counts row
    int blk[DANCE_FCOSE_MAX_LEVELS + 1];       // first block of the level
    int hw[DANCE_FCOSE_MAX_LEVELS];
    float stride[DANCE_FCOSE_MAX_LEVELS];
    int64_t total;                             // slots per image = sum of hw
    int n_levels;
};

__device__ __forceinline__ int fc_level(const LevelTable &t, int block)
{
    int l = 0;
    while (l + 1 < t.n_levels && block >= t.blk[l + 1])
        ++l;
    return l;
}

__global__ __launch_bounds__(FC_THREADS) void k_fcose_count(const LevelTable t, int32_t *__restrict__ counts,
                                                            int c, float thresh, int with_ctr)
{
    const int l = fc_level(t, blockIdx.x);
    const int hw = t.hw[l];
    const int loc = (blockIdx.x - t.blk[l]) * FC_THREADS + threadIdx.x;
    const int img = blockIdx.y;
    if (loc >= hw)
        return;
    const float cs = fc_sigmoid(t.ctr[l][(int64_t)img * hw + loc]);
    const float *p = t.cls[l] + (int64_t)img * c * hw + loc;
    int cnt = 0;
#pragma unroll 8
    for (int k = 0; k < c; ++k) {
        float v;
        cnt += fc_candidate(p[(int64_t)k * hw], cs, thresh, with_ctr, v) ? 1 : 0;
    }
    counts[(int64_t)img * t.total + t.off[l] + loc] = cnt;
}

__global__ __launch_bounds__(FC_THREADS) void k_fcose_fill(
    const LevelTable t, const int32_t *__restrict__ counts, const int64_t *__restrict__ cum, int c,
    float thresh, int with_ctr, float *__restrict__ raw, float *__restrict__ scores,
    int64_t *__restrict__ classes, float *__restrict__ out_loc, float *__restrict__ boxes,
    float *__restrict__ extpts)
{
    const int l = fc_level(t, blockIdx.x);
    const int hw = t.hw[l];
    const int loc = (blockIdx.x - t.blk[l]) * FC_THREADS + threadIdx.x;
    const int img = blockIdx.y;
    if (loc >= hw)
        return;
    const int64_t slot = (int64_t)img * t.total + t.off[l] + loc;
    const int cnt = counts[slot];
    if (cnt == 0)
        return;
    int64_t o = cum[slot] - cnt;   // cum is the inclusive prefix sum
    const float stride = t.stride[l];
    const float cs = fc_sigmoid(t.ctr[l][(int64_t)img * hw + loc]);
    const float lx = t.loc[l][2 * loc], ly = t.loc[l][2 * loc + 1];
    const float *r = t.reg[l] + (int64_t)img * 4 * hw + loc;
    // predict_proposals: r = r * s (fcose_outputs.py:431), then loc -+ r (501-506)
    const float x1 = lx - r[0] * stride;
    const float y1 = ly - r[(int64_t)hw] * stride;
    const float x2 = lx + r[2 * (int64_t)hw] * stride;
    const float y2 = ly + r[3 * (int64_t)hw] * stride;
    const float *ext = t.ext[l];
    float e[8];
    if (ext) {
        const float *q = ext + (int64_t)img * 4 * hw + loc;
        const float w = x2 - x1, h = y2 - y1;
        e[0] = lx + w * q[0];                 e[1] = y1;   // top
        e[2] = x1;  e[3] = ly + h * q[(int64_t)hw];        // left
        e[4] = lx + w * q[2 * (int64_t)hw];   e[5] = y2;   // bottom
        e[6] = x2;  e[7] = ly + h * q[3 * (int64_t)hw];    // right
    }
    const float *p = t.cls[l] + (int64_t)img * c * hw + loc;
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

// validates the caller's level list and lays the levels' blocks out; 0 blocks = nothing to do
int make_table(const dance_fcose_level *levels, int n_levels, bool need_reg, LevelTable &t, const char *fn)
{
    if (n_levels < 1 || n_levels > DANCE_FCOSE_MAX_LEVELS || !levels)
        return dance::fail((int)hipErrorInvalidValue, fn, "1 .. DANCE_FCOSE_MAX_LEVELS levels expected");
    t.n_levels = n_levels;
    t.total = 0;
    t.blk[0] = 0;
    bool any_ext = false, all_ext = true;
    for (int l = 0; l < n_levels; ++l) {
        const dance_fcose_level &v = levels[l];
        if (v.hw < 0 || (v.hw > 0 && (!v.cls || !v.ctr || (need_reg && (!v.reg || !v.locations)))))
            return dance::fail((int)hipErrorInvalidValue, fn, "level with hw > 0 and a null tensor");
        t.cls[l] = v.cls, t.ctr[l] = v.ctr, t.reg[l] = v.reg, t.ext[l] = v.ext, t.loc[l] = v.locations;
        t.hw[l] = v.hw, t.stride[l] = v.stride, t.off[l] = t.total;
        t.total += v.hw;
        t.blk[l + 1] = t.blk[l] + dance::ceil_div(v.hw, FC_THREADS);
        if (v.hw > 0)
            any_ext |= v.ext != nullptr, all_ext &= v.ext != nullptr;
    }
    if (need_reg && any_ext && !all_ext)
        return dance::fail((int)hipErrorInvalidValue, fn, "ext given for some levels only");
    return 0;
}

}  // namespace

extern "C" int dance_fcose_count(const dance_fcose_level *levels, int n_levels, int32_t *counts, int n,
                                 int c, float thresh, int thresh_with_ctr, void *stream)
{
    DANCE_REQUIRE(n >= 0 && n <= 65535 && c >= 1);
    LevelTable t;
    if (int e = make_table(levels, n_levels, false, t, __func__))
        return e;
    const int blocks = t.blk[n_levels];
    if (n == 0 || blocks == 0)
        return 0;
    DANCE_REQUIRE(counts);
    hipLaunchKernelGGL(k_fcose_count, dim3(blocks, n), dim3(FC_THREADS), 0, (hipStream_t)stream, t, counts,
                       c, thresh, thresh_with_ctr);
    return dance::launch_check(__func__);
}

extern "C" int dance_fcose_fill(const dance_fcose_level *levels, int n_levels, const int32_t *counts,
                                const int64_t *cum, int n, int c, float thresh, int thresh_with_ctr,
                                float *raw, float *scores, int64_t *classes, float *out_loc, float *boxes,
                                float *extpts, void *stream)
{
    DANCE_REQUIRE(n >= 0 && n <= 65535 && c >= 1);
    LevelTable t;
    if (int e = make_table(levels, n_levels, true, t, __func__))
        return e;
    const int blocks = t.blk[n_levels];
    if (n == 0 || blocks == 0)
        return 0;
    bool with_ext = false;
    for (int l = 0; l < n_levels; ++l)
        with_ext |= t.hw[l] > 0 && t.ext[l];
    DANCE_REQUIRE(counts && cum && raw && scores && classes && out_loc && boxes && (extpts || !with_ext));
    hipLaunchKernelGGL(k_fcose_fill, dim3(blocks, n), dim3(FC_THREADS), 0, (hipStream_t)stream, t, counts,
                       cum, c, thresh, thresh_with_ctr, raw, scores, classes, out_loc, boxes, extpts);
    return dance::launch_check(__func__);
}
