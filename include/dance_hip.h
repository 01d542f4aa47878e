/*
 * dance_hip.h -- C ABI of libdance_hip.so, the MI355X (gfx950) implementation of DANCE's
 * contour-deformation hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch.cuda memory) unless its name
 *     ends in _host; tensors are dense, row-major, shapes as written;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); kernels are
 *     enqueued on it and the call returns without synchronising; no entry point
 *     allocates, frees or copies (safe under graph capture) -- scratch comes in through
 *     explicit workspace arguments;
 *   - return value: 0 (hipSuccess) or a hipError_t; argument errors return
 *     hipErrorInvalidValue (1) before anything is enqueued; dance_last_error() has the
 *     text;
 *   - "reference" citations are into lkevinzc/dance: the op each entry point replaces.
 */
#ifndef DANCE_HIP_H_
#define DANCE_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- library ---------- */
int dance_abi_version(void);                 /* bumps when a signature changes          */
const char *dance_last_error(void);          /* thread-local text of the last failure   */
/* Fills CU count and LDS bytes per workgroup of the current device (host out-params). */
int dance_device_info(int *cu_count_host, int *lds_bytes_host);

/* How the fast-mode entry points pick among kernel forms that give identical results.  Passed by
 * pointer per call (NULL = the defaults below): the library holds no process-global switches and
 * reads no environment variables.
 *   few_contour_forms  1: batches with fewer contours than CUs (the reference's one-image
 *                      inference, ~100 instances on 256 CUs) run forms that spread a contour over
 *                      more hardware: half-ring head workgroups (n <= DANCE_FEW_HEAD, 65..128
 *                      vertices), the 8-wave residual stack (n <= DANCE_FEW_STACK, 97..128), two
 *                      fusion parts (n <= DANCE_FEW_FUSE, 97..128); 0: never.  Bit-identical.
 *   fuse_tail_split    1: the fusion kernel runs the contours of a mostly empty last round of
 *                      workgroups as half rings (dance_fuse16w_tail_split); 0: never.  Bit-identical.
 *   tail_group         contours per workgroup of dance_tail16: 1, 2 or 4; 0 = by batch size. */
typedef struct dance_policy {
    int few_contour_forms;
    int fuse_tail_split;
    int tail_group;
    int coop_stack;     /* 1: with few_contour_forms, batches of at most DANCE_FEW_COOP contours (and at most half
                           the CUs) run the residual stack on TWO workgroups per contour that exchange their
                           halves through L2 after every block (dance_res_stack16); 0: never.  Bit-identical. */
} dance_policy;
#define DANCE_FEW_HEAD 128
#define DANCE_FEW_STACK 256
#define DANCE_FEW_FUSE 128
#define DANCE_FEW_COOP 128

/* ------------------------------------------------------- extreme_utils (L6) -------- */
/* reference: core/layers/extreme_utils/src/utils.cu:114-212 (pybind `collect_extreme_point`,
 * utils.cpp:5).  ext_hm (b,4,h,w) f32, bbox (b,n,4) i64, radius (b,n) i64,
 * vote (b,h,w,2) f32, ct (b,n,2) f32 -> extreme_point (b,n,4,2) i64 (every element
 * written).  Pixels outside the map, which the reference reads unchecked, are skipped. */
int dance_collect_extreme_point(const float *ext_hm, const int64_t *bbox,
                                const int64_t *radius, const float *vote, const float *ct,
                                int64_t *extreme_point, int b, int n, int h, int w,
                                void *stream);

/* reference: utils.cu:215-300 (`calculate_edge_num`, utils.cpp:6).  Mutates edge_num
 * (b,n,orig_p_num) i64 in place; edge_num_sum (b,n) i64; edge_idx_sort (b,n,orig_p_num). */
int dance_calculate_edge_num(int64_t *edge_num, const int64_t *edge_num_sum,
                             const int64_t *edge_idx_sort, int b, int n, int orig_p_num,
                             int64_t p_num, void *stream);

/* reference: utils.cu:303-374 (`calculate_wnp`, utils.cpp:7).  weight (b,n,p_num,1) f32
 * and ind (b,n,p_num,2) i64: every slot is written (slots no edge covers get 0, which is
 * what the reference's zero-initialised outputs hold there). */
int dance_calculate_wnp(const int64_t *edge_num, const int64_t *edge_start_idx,
                        float *weight, int64_t *ind, int b, int n, int orig_p_num,
                        int p_num, void *stream);

/* reference: utils.cu:377-440 (`roll_array`, utils.cpp:8).  array (b,n,d) f32,
 * step (b) i64 -> new_array (b,n,d): new[b][(i+step[b]) mod n] = array[b][i]. */
int dance_roll_array(const float *array, const int64_t *step, float *new_array, int b,
                     int n, int d, void *stream);

/* reference: `nms` of the same pybind module (utils.cpp:9, utils.h:36-43, src/nms.cu:13-128;
 * exported, not called by the reference's Python).  boxes_sorted (n,5) f32 = x1,y1,x2,y2,score
 * in descending score order (the reference sorts with at::sort before its kernel, nms.cu:74-76;
 * the Python shim does the same).  "+1" IoU, strict > thresh.  mask_ws: (n + 1) * ceil(n/64)
 * uint64 of scratch.  keep (n) i64 receives the kept positions in the sorted order, ascending;
 * count (1) i32 their number.  The sweep runs on the device: no host copy of the mask. */
int dance_nms_sorted(const float *boxes_sorted, int n, float thresh, uint64_t *mask_ws,
                     int64_t *keep, int32_t *count, void *stream);

/* The same with the IoU of torchvision.ops.nms (widths / heights without the "+1"): the kernel
 * behind detectron2.layers.batched_nms, which the reference's multi-class `ml_nms` calls
 * (core/layers/ml_nms.py:25, used at fcose_outputs.py:528; dance_amd.ml_nms.ml_nms moves the
 * classes apart by a coordinate offset first, as torchvision does). */
int dance_nms_sorted_plain(const float *boxes_sorted, int n, float thresh, uint64_t *mask_ws,
                           int64_t *keep, int32_t *count, void *stream);

/* dance_nms_sorted_plain that allocates nothing and takes no workspace.  NO counterpart in the
 * reference: it exists for callers that cannot call hipMalloc -- code under HIP graph capture,
 * tight serving loops, harnesses that forbid allocation.  The scratch (IoU bit mask + removal
 * words) is a file-scope device array of (4096 + 1) * 64 uint64 = 2.0 MiB that every GPU the
 * library runs on holds until the process ends; its per-device address is looked up at the first
 * call on that device (hipGetSymbolAddress; no allocation then either).  n <= 4096
 * (DANCE_NMS_STATIC_MAX), larger lists return hipErrorInvalidValue with a text naming
 * dance_nms_sorted_plain.  Same results as dance_nms_sorted_plain, bit for bit.  All calls on
 * one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race;
 * serialise them or use the workspace form.  Nothing has to be cleared between calls.
 * Under HIP graph capture: make ONE call per device outside the capture first (the address
 * look-up of the first call may load the code object, which allocates and synchronises), and keep
 * the current device equal to the device that owns `stream` (the array is picked by hipGetDevice;
 * a stream of another device is refused with hipErrorInvalidDevice). */
#define DANCE_NMS_STATIC_MAX 4096
int dance_nms_sorted_plain_static(const float *boxes_sorted, int n, float thresh, int64_t *keep,
                                  int32_t *count, void *stream);

/* dance_nms_sorted_plain for many independent lists in two launches -- the per-image loop of
 * select_over_all_levels (fcose_outputs.py:523-529) without a launch pair and a host sync per
 * image.  boxes_sorted: the segments' (n_i,5) rows back to back, each sorted by descending score;
 * seg (n_seg,3) i64 on the device: first row, n_i, first mask word of segment i (segment i needs
 * (n_i + 1) * ceil(n_i/64) words of mask_ws); n_max = largest n_i; keep (sum n_i) i64: kept
 * positions relative to the segment, written from the segment's first row; count (n_seg) i32. */
int dance_nms_segments_plain(const float *boxes_sorted, const int64_t *seg, int n_seg, int n_max,
                             float thresh, uint64_t *mask_ws, int64_t *keep, int32_t *count,
                             void *stream);

/* reference: snake_head.py:1752-1761 (`RefineNet.contour_upsample_2x`, used only with
 * LAST_UP_SAMPLE).  loc (n,v,2) f32 -> out (n,2v,2): even slots copy, odd slots are the
 * mid-points (p[i] + p[i+1 mod v]) / 2, rounded as the reference's add-then-divide. */
int dance_contour_upsample_2x(const float *loc, float *out, int n, int v, void *stream);

/* ----------------------------------------------- feature prep ------ */
/* reference: RefineNet.bottom_out, snake_head.py:562-615 (construction: Conv2d 3x3, padding 1,
 * bias = not norm, GroupNorm(32, C), ReLU) and 1816-1820 (use, timer scope snake_feat_prep).
 * The three entry points below are one bottom_out layer in the split-half fp16 arithmetic of
 * the fast mode (fp32-level accuracy); dilation 1 only.
 *
 * dance_nchw_to_p16img: channels [c_off, c_off + c) of src (b, c_total, h, w) fp32 ->
 *   dst16 "P16 image" [b][plane hi/lo][c/8][h][w][8] fp16 (4 bytes per element). */
int dance_nchw_to_p16img(const float *src, int c_total, int c_off, void *dst16, int b, int c,
                         int h, int w, void *stream);
/* dance_nhwc_to_p16img: the same from an NHWC (channels_last) source (b, h, w, c_total): the
 *   layout dance_gn_relu's out_nhwc and torch's channels_last tensors have, so consecutive
 *   conv3x3 + GN + ReLU layers of the semantic-FPN head (edge_det.py:106-138) need no NCHW copy. */
int dance_nhwc_to_p16img(const float *src, int c_total, int c_off, void *dst16, int b, int c,
                         int h, int w, void *stream);
/* dance_upsample2x_add: out (b, 2h, 2w, c) = acc + bilinear x2 of src (b, h, w, c), NHWC fp32,
 *   align_corners = False -- the last nn.Upsample of a semantic-FPN scale head and the sum over
 *   the levels (edge_det.py:118-125, 188-194) in one pass.  acc may be NULL (plain upsampling)
 *   and may be `out` itself (in place).  c % 4 == 0, 16-byte aligned pointers. */
int dance_upsample2x_add(const float *src, const float *acc, float *out, int b, int h, int w, int c,
                         void *stream);
/* dance_conv3x3_16: raw 3x3 convolution (zero padding 1, no bias) of a P16 image with weights
 *   packed [plane][tap ky*3+kx][cin/8][cout][8] fp16 of w * 2^s (w_inv_scale = 2^-s).
 *   y_blocked (b, cout/16, h, w, 16) fp32; stats (b, groups, 2) f64 = per-group sum and sum of
 *   squares of y, summed in a fixed order (no atomics: two forwards give identical bits) from the
 *   workgroups' partial sums in stats_partial, dance_conv3x3_16_partial_floats(...) floats of
 *   scratch. */
int64_t dance_conv3x3_16_partial_floats(int b, int cout, int h, int w, int groups);
int dance_conv3x3_16(const void *x16, const void *w16, float w_inv_scale, float *y_blocked,
                     double *stats, float *stats_partial, int b, int cin, int cout, int h, int w,
                     int groups, void *stream);
/* dance_gn_relu: GroupNorm(groups, eps) * gamma + beta, then ReLU, of y_blocked using stats;
 *   exactly one of out16 (P16 image, the next layer's input) / out_nhwc ((b, h, w, c) fp32,
 *   what the gather kernels read) is non-null.  absmax (with out_nhwc; may be NULL): the
 *   per-image max |output| words of dance_nchw_to_nhwc, measured on the way out; absmax_partial
 *   (b, dance_absmax_partials(DANCE_ABSMAX_OF_GN_RELU, c, h * w)) scratch, required with absmax. */
int dance_gn_relu(const float *y_blocked, const double *stats, const float *gamma,
                  const float *beta, float eps, int groups, void *out16, float *out_nhwc,
                  uint32_t *absmax, uint32_t *absmax_partial, int b, int c, int h, int w,
                  void *stream);

/* dance_conv1x1_nhwc: out (b, k, h, w) = bias (k, or NULL) + w (k, c) . x (b, h, w, c) NHWC fp32, a 1x1
 *   convolution to a few output channels summed in a fixed order (bit-reproducible, unlike the
 *   stock split-K kernels): the edge predictor's last layer, edge_det.py:140-163 applied at :197.
 *   c % 4 == 0, x and w 16-byte aligned. */
int dance_conv1x1_nhwc(const float *x, const float *w, const float *bias, float *out, int b, int h,
                       int w_, int c, int k, void *stream);

/* ------------------------------------ attention-map producer ------ */
/* reference: EdgeSnakeFPNHead.attender, core/modeling/edge_snake/edge_det.py:66-100, applied
 * at 200-204 to x = 1 - pred_edge: [Conv2d 3x3 pad 1 no bias -> GroupNorm(4,32) -> ReLU] for
 * 1->32 and 32->32 channels, then Conv2d(32->1, 3x3, pad 1, bias) -> sigmoid.  fp32 FMA.
 *   x (b,h,w) f32 (the reference's (b,1,h,w)); w1 (32,1,3,3), w3 (1,32,3,3) as in the
 *   state-dict; w2_packed (32 cin, 3, 3, 32 cout) = attender.1.weight.permute(1,2,3,0);
 *   gn*_w / gn*_b (32); b3 (1); att (b,h,w) f32 -- the attention plane the gather kernels
 *   read (channel 0 of the snake head's input).  workspace: dance_attender_workspace_bytes. */
int64_t dance_attender_workspace_bytes(int b, int h, int w);
int dance_attender(const float *x, const float *w1, const float *gn1_w, const float *gn1_b,
                   const float *w2_packed, const float *gn2_w, const float *gn2_b, const float *w3,
                   const float *b3, float eps, float *att, void *workspace, int64_t workspace_bytes,
                   int b, int h, int w, void *stream);

/* --------------------------------------- polygon -> RLE mask ------ */
/* reference call sites: get_polygon_rles = mask_util.merge(mask_util.frPyObjects([p.tolist()],
 * h, w)) per instance, on the CPU after a device -> host copy (core/evaluation/
 * coco_evaluation.py:311-318,352-357; core/modeling/postprocessing.py:266-274;
 * core/modeling/dsnake_baseline/postprocessing.py:13-21,71).  The arithmetic is pycocotools'
 * (cocoapi common/maskApi.c rleFrPoly; not part of the reference tree, unpinned there:
 * README.md:27); results are bit-identical to oracle/rle_ref.c, the restatement of that
 * published algorithm.
 *   polys (n,v,2) f32 x,y in pixels of the (h,w) image, v <= 512; sizes (n,2) i32 (h,w) per
 *   polygon, or NULL = the scalars h, w for all (h <= 327680, h*w < 2^31);
 *   counts (n,cap) u32: run lengths in column-major order, first run = zeros; m (n) i32 = number
 *   of runs (rows with m > cap hold the first cap runs); area (n) u32 = sum of the runs of ones
 *   (rleArea), may be NULL; status (n) i32, may be NULL: 0 ok, 1 = m > cap, 2 = size not
 *   supported.  Coordinates beyond +-209,715 px and non-finite ones (the library's conversion of
 *   those is undefined) are clamped to that range. */
int dance_polygon_rle(const float *polys, const int32_t *sizes, int n, int v, int h, int w,
                      uint32_t *counts, int cap, int32_t *m, uint32_t *area, int32_t *status,
                      void *stream);
/* rleToString of the same library (the "counts" bytes the reference's evaluator stores,
 * coco_evaluation.py:369-374).  len (n) i32 = bytes each row needs.  str NULL: lengths only.
 * offsets NULL: row i is written at str + i * str_cap; else at str + offsets[i], offsets being
 * (n+1) i64 (exclusive prefix sum of the lengths): one packed buffer, one device -> host copy.
 * A row that does not fit its room is left incomplete (its len says so). */
int dance_rle_to_string(const uint32_t *counts, const int32_t *m, int n, int cap, uint8_t *str,
                        const int64_t *offsets, int64_t str_cap, int32_t *len, void *stream);
/* rleDecode: mask (n, w, h) u8 -- the library's column-major memory order, mask[i][x][y] --
 * zero-filled on the stream, then the runs of ones are written. */
int dance_rle_decode(const uint32_t *counts, const int32_t *m, int n, int cap, uint8_t *mask,
                     int h, int w, void *stream);

/* ------------------------------------------- FCOSE box decode ------ */
/* reference: FCOSEOutputs.forward_for_single_feature_map, core/modeling/fcose/
 * fcose_outputs.py:444-521, for every FPN level (predict_proposals, 419-442, multiplies each
 * level's reg_pred by its stride: 431).  Two passes over the level's logits, in the reference's
 * own NCHW layout: cls (n,c,h,w), ctr (n,1,h,w), reg / ext (n,4,h,w) f32, hw = h*w.
 *
 * All levels go in one launch: the caller lists them (host array of dance_fcose_level).
 *
 * dance_fcose_count: counts (n, total) i32, total = sum of the levels' hw; entry
 *   [img][level start + loc] = number of classes of that location whose score passes:
 *   sigmoid(cls) * sigmoid(ctr) > thresh when thresh_with_ctr, else sigmoid(cls) > thresh
 *   (463-465).  reg / ext / locations / stride of the levels are not read.
 * dance_fcose_fill: cum = inclusive prefix sum (i64) of the whole counts array; candidate j of a
 *   location lands in slot cum - count + j, i.e. in the order of the reference's `nonzero()`
 *   (image, level, location, class).  Per candidate: raw = sigmoid(cls) * sigmoid(ctr),
 *   scores = sqrt(raw) (511), classes i64 (512), out_loc (.,2) = locations[loc] (513), boxes
 *   (.,4) = loc -+ reg * stride (501-506, bit-identical: separate fp32 multiply and add), extpts
 *   (.,4,2) = ExtremePoints.from_boxes (core/structures/pointset.py:110-130); ext (all levels or
 *   none) and extpts may be NULL.  locations (hw,2) f32 as FCOSE.compute_locations gives them
 *   (fcose.py:129-142).  The per-level top-k (492-499) and select_over_all_levels (523-542) stay
 *   with the caller (dance_amd/fcose_outputs.py: torch.topk, dance_nms_sorted_plain, kthvalue). */
#define DANCE_FCOSE_MAX_LEVELS 8
typedef struct dance_fcose_level {
    const float *cls, *ctr, *reg, *ext, *locations;   /* device pointers, see above */
    int hw;                                           /* h * w of the level */
    float stride;                                     /* FPN stride: reg is multiplied by it */
} dance_fcose_level;
int dance_fcose_count(const dance_fcose_level *levels, int n_levels, int32_t *counts, int n, int c,
                      float thresh, int thresh_with_ctr, void *stream);
int dance_fcose_fill(const dance_fcose_level *levels, int n_levels, const int32_t *counts,
                     const int64_t *cum, int n, int c, float thresh, int thresh_with_ctr, float *raw,
                     float *scores, int64_t *classes, float *out_loc, float *boxes, float *extpts,
                     void *stream);

/* --------------------------------- training-side target matching --- */
/* The NumPy half of RefineNet.compute_targets_for_polys, ONE WORKGROUP PER INSTANCE.  Float
 * results are float64 (what NumPy 2.2 promotes the reference's lines to) and bit-identical to the
 * reference's on the same inputs; status (n) i32: 0 ok, 1 = the reference raises / falls back
 * there, 2 = sizes outside the kernels' limits (2048 polygon vertices / contour points).
 *
 * dance_uniform_sample: reference RefineNet.uniform_sample (snake_head.py:661-725).  polys
 *   (sum p, 2) f32 with poly_off (n + 1) i32; newpnum (n) i32; out (sum newpnum, 2) f64 at
 *   out_off (n + 1) i64 (points).  p > newpnum: the shortest edges' start points dropped (the
 *   kept float32 rows, widened); else points spread over the edges in proportion to their length.
 *   Ties in edge length go to the lowest index (the reference's np.argsort is unstable there).
 * dance_dense_targets: uniform_sample(cnt, p * m), rolled so that the point nearest first[n]
 *   (f32 x, y: the initial contour's vertex 0) leads, every p-th point kept (snake_head.py:
 *   1134-1144): out (n, m, 2) f64; the p * m points are never stored.
 * dance_uniform_sample_1d: reference RefineNet.uniform_sample_1d (snake_head.py:728-762): open
 *   poly-lines pts (sum nn, 2) f64 at off (n + 1) i32 -> new_n[n] points each.
 * dance_aux_extreme_points: reference get_aux_extreme_points (core/modeling/fcose/utils.py:
 *   94-189): pts (n, num_points, 2) f64 closed contours -> reordered (same shape), ext_idx
 *   (n, cap) i64 sorted unique indices, count (n).
 * dance_segment_matching: reference RefineNet.single_uniform_segment_matching (snake_head.py:
 *   1265-1384, INITIAL = octagon) with its fall-back reorder_perloss (1520-1540): dense
 *   (n, v * up_rate, 2) f64, sampled (n, v, 2) f32, edge_idx (n, e_num) i64 -> out (n, v, 2) f64;
 *   status 1 = the fall-back was taken.
 * dance_multisegment_matching: reference RefineNet.single_uniform_multisegment_matching
 *   (snake_head.py:1386-1498, the shipped INITIAL = box + NEW_MATCHING form) together with the
 *   roll of the initial contour before it (1150-1155): dense (n, v * up_rate, 2) f64 = the
 *   reordered contour of dance_aux_extreme_points, ext_idx (n, cap) i64 / ext_count (n) its
 *   indices; sampled (n, v, 2) f32 -> rolled (n, v, 2) f32 (starts at the vertex nearest
 *   dense[0]) and out (n, v, 2) f64; status 1 = fall-back, 2 = fewer than 2 or more than 64
 *   indices (the reference fails on fewer than 2). */
int dance_uniform_sample(const float *polys, const int32_t *poly_off, const int32_t *newpnum,
                         const int64_t *out_off, double *out, int32_t *status, int n, void *stream);
int dance_dense_targets(const float *polys, const int32_t *poly_off, const float *first, int m,
                        double *out, int32_t *status, int n, void *stream);
int dance_uniform_sample_1d(const double *pts, const int32_t *off, const int32_t *new_n,
                            const int64_t *out_off, double *out, int32_t *status, int n, void *stream);
int dance_aux_extreme_points(const double *pts, double *reordered, int64_t *ext_idx, int32_t *count,
                             int32_t *status, int n, int num_points, int cap, void *stream);
int dance_segment_matching(const double *dense, const float *sampled, const int64_t *edge_idx,
                           double *out, int32_t *status, int n, int v, int up_rate, int e_num,
                           void *stream);
int dance_multisegment_matching(const double *dense, const float *sampled, const int64_t *ext_idx,
                                const int32_t *ext_count, float *rolled, double *out, int32_t *status,
                                int n, int v, int up_rate, int cap, void *stream);

/* ------------------------------------------------- contour initialisation (a11-a15) - */
enum {
    DANCE_POLY_RAW = 0,      /* src = polygons (n,e,2)                                  */
    DANCE_POLY_BOX = 1,      /* src = boxes (n,4) xyxy -> 4 corners, snake_head.py:870-879 */
    DANCE_POLY_QUAD = 2,     /* src = boxes (n,4) -> edge mid-points, dsnake_head.py:501-517 */
    DANCE_POLY_OCTAGON = 3   /* src = extreme points (n,4,2) -> 12-gon, snake_head.py:796-830 */
};
/* reference: RefineNet.uniform_upsample (snake_head.py:764-794) fused with its two
 * kernels and with the polygon builders above.  out (n,p_num,2) f32,
 * edge_start_idx (n,e) i64 (e = 4, 4, 12 for BOX, QUAD, OCTAGON; may be NULL),
 * status (n) i32: 1 where the counts could not be brought to p_num (the reference
 * asserts, snake_head.py:782); such rows are zero-filled.  e <= 64, p_num <= 4096. */
int dance_uniform_upsample(int kind, const float *src, int n, int e, int p_num, float *out,
                           int64_t *edge_start_idx, int32_t *status, void *stream);

/* dance_uniform_upsample(DANCE_POLY_OCTAGON) with what DsnakeHead.forward does to the init snake's
 * extreme points in front of get_octagon (dsnake_head.py:800-817) folded into the load: clamp != 0
 * clamps x to [0, x_max] and y to [0, y_max] (tensor.clamp_: NaN stays NaN); where override_mask[i]
 * (n, u8; NULL together with override_pts) is non-zero the instance's points are taken from
 * override_pts (n,4,2) instead (ground-truth extreme points, never clamped).  ext_used (n,4,2) or
 * NULL receives the points the octagon was built from (the module's `ext_points`).
 * ext_points is read with strides (floats): instance i, point k at ext_points + i * poly_stride +
 * k * point_stride -- (8, 2) for a dense (n,4,2) tensor, (80, 20) for every 10th vertex of the init
 * snake's (n,40,2) ring (`pred[:, ::10]`, dsnake_head.py:745) without a copy. */
int dance_octagon_upsample(const float *ext_points, int64_t poly_stride, int64_t point_stride,
                           const float *override_pts,
                           const uint8_t *override_mask, int clamp, float x_max, float y_max, int n,
                           int p_num, float *out, float *ext_used, int64_t *edge_start_idx,
                           int32_t *status, void *stream);

/* ----------------------------------------------------------------- layout ---------- */
/* (b, c_total, hw) NCHW -> feat (b, hw, c_total - att_channels) NHWC and, when
 * att_channels == 1, channel 0 -> att (b, hw).  Replaces the split at
 * snake_head.py:1812-1814 / 1949-1951 plus the layout change the gather wants.
 * absmax (b, DANCE_ABSMAX_WORDS) u32 or NULL: per image, partial maxima of |feature| as float bit
 * patterns (a non-negative float orders like an unsigned; a NaN is above everything); the fast
 * mode's head block scales its input by their maximum (dance_snake_evolve16).  Measured without
 * atomics: every workgroup stores its own maximum into absmax_partial (b, dance_absmax_partials(
 * DANCE_ABSMAX_OF_LAYOUT, c, hw)) u32 scratch (required with absmax), a second small launch folds
 * them -- order-independent, hence deterministic. */
#define DANCE_ABSMAX_WORDS 8
enum { DANCE_ABSMAX_OF_LAYOUT = 0, DANCE_ABSMAX_OF_GN_RELU = 1, DANCE_ABSMAX_OF_FEATURES = 2 };
/* scratch words PER IMAGE the producer needs for an image of c channels x hw pixels */
int64_t dance_absmax_partials(int producer, int c, int64_t hw);
int dance_nchw_to_nhwc(const float *src, float *feat, float *att, uint32_t *absmax,
                       uint32_t *absmax_partial, int b, int c_total, int att_channels, int hw,
                       void *stream);
/* The same words for an NHWC feature tensor that arrives from elsewhere: feat (b, per_image floats);
 * absmax_partial (b, dance_absmax_partials(DANCE_ABSMAX_OF_FEATURES, 1, per_image)). */
int dance_feature_absmax(const float *feat, int b, int64_t per_image, uint32_t *absmax,
                         uint32_t *absmax_partial, void *stream);

/* ------------------------------------------------------------ gather (a1-a3) ------- */
enum { DANCE_PAD_ZEROS = 0, DANCE_PAD_REFLECTION = 1 };
enum { DANCE_DELOC_DERANGE = 0, DANCE_DELOC_DEMEAN = 1, DANCE_DELOC_DEMIN = 2,
       DANCE_DELOC_NONE = 3 };
/* reference: RefineNet.get_locations_feature (snake_head.py:1542-1578) / DsnakeHead's
 * (dsnake_head.py:670-695) + de_location (1580-1614 / 697-708) + the concat of
 * evolve (1643-1652).
 *   feat (b,hf,wf,c) f32 NHWC, c % 4 == 0; att (b,hf,wf) or NULL; loc (n,v,2) image px;
 *   image_idx (n) i32 or NULL (= image b-1, what the reference's loop leaves behind);
 *   x_out (n, c + (deloc != NONE ? 2 : 0), v) channel-major; att_out (n,v) or NULL. */
int dance_gather_features(const float *feat, const float *att, const float *loc,
                          const int32_t *image_idx, float *x_out, float *att_out, int n,
                          int v, int b, int hf, int wf, int c, int padding_mode,
                          int deloc_type, void *stream);

/* ------------------------------------------------------- snake blocks (a4-a6) ------ */
enum {
    DANCE_EP_RELU = 1,      /* y = max(y, 0) after bias                                 */
    DANCE_EP_AFFINE = 2,    /* y = y*scale[o] + shift[o]  (eval BatchNorm1d after ReLU) */
    DANCE_EP_RESIDUAL = 4,  /* y += residual[n][o][v]                                   */
    DANCE_EP_ROWINIT = 8,   /* accumulator starts at rowinit[n][o] instead of bias[o]   */
    DANCE_EP_MAXV = 16,     /* store max over v into y (n,cout) instead of (n,cout,v)   */
    DANCE_EP_OUT_P16 = 32   /* y is a split-half (P16) block, y_stride in halfs         */
};
/* Circular dilated Conv1d over the vertex ring as an fp32 MFMA implicit GEMM:
 *   y[n][o][v] = ep( bias[o] + sum_{t<taps} sum_{c<cin} w[t][c][o] * x[n][c][(v+(t-taps/2)*dilation) mod V] )
 * reference: DilatedCircularConv (snake_head.py:92-116) + SnakeBlock (140-153) + the
 * residual add of _SnakeNet.forward (266); taps = 1 gives the k=1 Conv1d's (fusion 249,
 * prediction 252-258, Dsnake fuse dsnake_head.py:242).
 *   x: channel-major (cin, v) per contour, contour stride x_stride floats;
 *   w_packed (taps, cin_pad, cout) f32 with cin_pad = cin rounded up to 16, zero rows;
 *   y: (cout, v) per contour with contour stride y_stride floats, or (n,cout) for MAXV;
 *   residual: same layout/stride as y.  v % 4 == 0, 16 <= v <= 256, cout % 32 == 0. */
int dance_circconv(const float *x, int64_t x_stride, const float *w_packed,
                   const float *bias, const float *scale, const float *shift,
                   const float *residual, const float *rowinit, float *y, int64_t y_stride,
                   int n, int v, int cin, int cout, int taps, int dilation, int flags,
                   void *stream);

/* Fused first block: gather (as dance_gather_features) straight into LDS, then the head
 * SnakeBlock (cin = c + 2, taps 9, dilation 1, ReLU + affine).  reference: evolve
 * (snake_head.py:1639-1652) + _SnakeNet.head (263).  Also writes att_out (n,v). */
int dance_gather_circconv_head(const float *feat, const float *att, const float *loc,
                               const int32_t *image_idx, const float *w_packed,
                               const float *bias, const float *scale, const float *shift,
                               float *y, int64_t y_stride, float *att_out, int n, int v,
                               int b, int hf, int wf, int c, int cout, int padding_mode,
                               int deloc_type, void *stream);

/* h[n][o] = bias[o] + sum_g wg[g][o] * g[n][g]: the vertex-invariant input columns of a
 * k=1 Conv1d folded into a per-contour row constant (used through DANCE_EP_ROWINIT).
 * reference: the expanded global state fed to prediction[0] (snake_head.py:272-282) and
 * the expanded centre feature fed to Dsnake's fuse (dsnake_head.py:723-729).
 *   g (n,g_dim), wg (g_dim,cout), bias (cout) -> h (n,cout); g_dim <= 1024. */
int dance_row_constant(const float *g, const float *wg, const float *bias, float *h, int n,
                       int g_dim, int cout, void *stream);

/* ------------------------------------------------ offset head + update (a6 tail-a8) - */
/* y[n][v][:] = loc + scale(tanh?)(w3^T p2 + b3) * att, optional clamp of contour 0.
 * reference: prediction[4] (snake_head.py:257), evolve tail 1702-1711, clip_locations
 * 1715-1727 (clip_first: only contour 0 is clamped -- the reference's behaviour when
 * image_idx is None; with a tensor image_idx it clamps nothing).
 *   p2 (n,64,v) f32; w3 (64,2); b3 (2); loc (n,v,2); whs (n,2) or NULL (no tanh scaling);
 *   att (n,v) or NULL; out (n,v,2); clip_x_max = image width - 1, clip_y_max = height - 1. */
int dance_offset_update(const float *p2, int64_t p2_stride, const float *w3, const float *b3,
                        const float *loc, const float *whs, const float *att, float *out,
                        int n, int v, int cmid, int clip_first, float clip_x_max,
                        float clip_y_max, void *stream);

/* ------------------------------------------------------- one whole iteration ------- */
/* Device pointers to one _SnakeNet's packed parameters (dance_amd/snake.py packs them
 * from a reference state-dict).  All f32. */
typedef struct dance_snake_net {
    const float *head_w, *head_b, *head_scale, *head_shift;           /* (9,cin_pad,128)      */
    const float *res_w[7], *res_b[7], *res_scale[7], *res_shift[7];   /* (9,128,128)          */
    const float *fusion_w, *fusion_b;                                 /* (1,1024,256)         */
    const float *pred0_wg;                                            /* (256,256) [g][o]     */
    const float *pred0_ws, *pred0_b;                                  /* (1,1024,256)         */
    const float *pred1_w, *pred1_b;                                   /* (1,256,64)           */
    const float *pred2_w, *pred2_b;                                   /* (64,2), (2)          */
    int dilation[7];
    int feat_channels;                                                /* c (256)              */
} dance_snake_net;

/* Bytes of workspace one dance_snake_evolve call needs for n contours of v vertices. */
int64_t dance_snake_workspace_bytes(int n, int v);

/* One evolve step (reference: RefineNet.evolve snake_head.py:1624-1713 /
 * DsnakeHead.evolve dsnake_head.py:748-765): gather -> snake net -> offset update.
 *   individual_scale: tanh(off) * whs (whs required); use_att: multiply by the sampled
 *   attention (att map required); x_in: optional precomputed (n, c+2, v) input that
 *   replaces the gather (Dsnake init path), else NULL;
 *   stage_events_host: NULL, or DANCE_EVOLVE_EVENTS hipEvent_t handles recorded on `stream`
 *   before the first kernel and after each stage (head | res0..6 | fusion | row constant |
 *   prediction[0] | prediction[2] | offset update) -- timing instrumentation for bench.py;
 *   individual entries may be NULL (that boundary is then not recorded: an event record
 *   costs ~4 us of stream time).  When the residual blocks run as one launch
 *   (dance_res_stack16) the events after res0..res5 follow it immediately. */
#define DANCE_EVOLVE_EVENTS 14
int dance_snake_evolve(const dance_snake_net *net_host, const float *feat, const float *att,
                       const float *loc, const int32_t *image_idx, const float *whs,
                       const float *x_in, float *out, void *workspace,
                       int64_t workspace_bytes, int n, int v, int b, int hf, int wf,
                       int padding_mode, int deloc_type, int individual_scale, int use_att,
                       int clip_first, float clip_x_max, float clip_y_max, int fused_head,
                       void **stage_events_host, void *stream);

/* ------------------------------------------------ fast precision mode (P16) ------- *
 * Activations and weights as fp16 hi/lo pairs (22 significant bits), every product as three
 * f16 MFMAs accumulated in fp32: fp32-level accuracy at 16/3 of the fp32-MFMA rate
 * (dance_amd/csrc/p16.h has the format).  dance_snake_evolve stays the exact-fp32 path.
 *
 * Range.  A pair carries its 22 bits while 2^-3 <= |x| < 65504 and degrades gracefully below
 * (absolute resolution 2^-25); beyond 65504 it is inf and surfaces as NaN coordinates -- loud,
 * never silently wrong.  Everything that is split is therefore first multiplied by a power of
 * two that is undone exactly where the fp32 accumulator is read: the head block's input by a
 * per-contour factor from the image's measured max |feature| (absmax words) and the contour's
 * de_location range; the state slots and prediction[0]'s output by per-net factors fixed when
 * the weights are packed (dance_amd/snake.py: from the BatchNorm parameters / weight norms), which
 * arrive folded into the constants of dance_snake_net16.
 *
 * P16 block of C channels (C % 128 == 0 or C <= 128; always slots of 128):
 *   [slot][plane hi/lo][c/8][v][8] halfs, slot stride given by the caller.
 * Packed weights w16: [plane][tap][cin_pad16/8][cout][8] halfs of (w * 2^s), with
 * bias_scaled = bias * 2^s and w_inv_scale = 2^-s (dance_amd.ops.pack_conv_weight16).
 *
 * mfma_terms: 3 = the parity mode described above; 1 = only the hi x hi product (plain fp16
 * operands, fp32 accumulate: one third of the MFMAs, ~1e-2 coordinate error -- the
 * approximate "half precision" variant of BASELINE config 2, never the default).            */

/* One SnakeBlock / k=1 Conv1d on P16 input (the single-block form; dance_res_stack16 is what
 * dance_snake_evolve16 runs).  Output either P16 (y16) or fp32 channel-major (y_f32);
 * residual != 0 adds the input itself times residual_mul (a power of two: 1 when input and
 * output are stored at the same activation scale; needs cin == cout <= 128); rowinit (n,cout)
 * replaces the bias as accumulator start; maxv != 0 writes max over the ring to y_f32 (n,cout).
 * reference: SnakeBlock snake_head.py:140-153 + residual :266; k=1 Conv1d :252-258. */
int dance_conv16(const void *x, int64_t x_stride, int64_t x_slot_stride, const void *w16,
                 const float *bias_scaled, float w_inv_scale, const float *scale,
                 const float *shift, const float *rowinit, int residual, int relu, int maxv,
                 void *y16, int64_t y_stride, int64_t y_slot_stride, float *y_f32,
                 int64_t yf_stride, float residual_mul, int n, int v, int cin, int cout, int taps,
                 int dilation, int mfma_terms, void *stream);

/* fusion Conv1d + max over the ring, and the 1024 state columns of prediction[0], in one pass over
 * the (n, 1024, v) P16 state (csrc/fuse16w.hip, v_mfma_f32_16x16x32_f16, 16-wide vertex tiles).
 * w16 packs fusion.weight (outputs 0..255, inverse scale w_inv_scale) and prediction[0]'s state
 * columns (outputs 256..511, pred_w_inv_scale -- which also carries the scale the stored block
 * is kept at) as one 512-output GEMM; bias_scaled (512) = fusion bias * 2^s, then zeros.
 * Both convolutions are pointwise over the ring, so a ring runs as `parts` workgroups
 * (dance_fuse16_parts: 2 above 128 vertices and, for few contours, from 97; dance_fuse16w_supported
 * (parts, v): one part for 16..128 vertices, two for 97..256), a part as ceil(vertices / 16) column
 * tiles rounded up to an instantiated count.  Outputs: gpart (n, parts, 256) f32 = max of the fusion
 * output over each part's vertices, and y16 (2-slot P16) = prediction[0]'s state part WITHOUT bias /
 * row constant / ReLU.  dance_tail16 takes both.  reference: _SnakeNet.forward snake_head.py:269-278. */
int dance_fuse16_parts(int n, int v, const dance_policy *policy);
int dance_fuse16w_supported(int parts, int v);
int dance_fuse16w_raw(const void *x, int64_t x_stride, int64_t x_slot_stride, const void *w16,
                      const float *bias_scaled, float w_inv_scale, float pred_w_inv_scale, void *y16,
                      int64_t y_stride, int64_t y_slot_stride, float *gpart, int n, int v, int parts,
                      int mfma_terms, void *stream);

/* dance_fuse16w_raw for rings of 97..128 vertices whose LAST n_split contours run as two half-ring
 * workgroups each: the kernel holds one workgroup per CU, so n contours take ceil(n / CUs) rounds and
 * the last round of n mod CUs workgroups costs as much as a full one (1600 contours on 256 CUs: 6.25
 * -> 7 rounds); as half rings it takes about half as long on twice the CUs.  gpart2 is (n, 2, 256):
 * whole rings write their max to both part slots, so dance_tail16 is called with parts = 2.
 * dance_fuse16w_tail_split(n, v, policy): the n_split dance_snake_evolve16 uses -- n mod CUs when there
 * is more than one round and that remainder fills at most half the CUs, else 0.  Outputs bit-identical
 * to dance_fuse16w_raw with one part (same products, same order; the max over a ring is the max of its
 * halves' maxima).  reference: as dance_fuse16w_raw. */
int dance_fuse16w_tail_split(int n, int v, const dance_policy *policy);
int dance_fuse16w_raw_split(const void *x, int64_t x_stride, int64_t x_slot_stride, const void *w16,
                            const float *bias_scaled, float w_inv_scale, float pred_w_inv_scale,
                            void *y16, int64_t y_stride, int64_t y_slot_stride, float *gpart2, int n,
                            int v, int n_split, int mfma_terms, void *stream);

/* The end of a _SnakeNet and of evolve in one launch: global state = max over gpart's parts ->
 * row constant h = pred0_bias + wg^T g (as dance_row_constant) -> relu(p1 + h) -> prediction[2]
 * (w1_16: packed (2,1,32,64,8), b1_scaled, w1_inv_scale) + ReLU -> prediction[4] + offset update
 * exactly as dance_offset_update (same arguments).  The (n,64,v) block between the last two
 * stays on chip.  When p1 is kept at a power-of-two scale 2^a, wg / pred0_bias / b1_scaled arrive
 * multiplied and w1_inv_scale divided by it (dance_snake_net16).
 * reference: snake_head.py:272-282 (row constant, prediction), 1702-1727. */
int dance_tail16(const void *p1, int64_t p1_stride, int64_t p1_slot_stride, const float *gpart,
                 int parts, const float *wg, const float *pred0_bias, const void *w1_16,
                 const float *b1_scaled, float w1_inv_scale, const float *w3, const float *b3,
                 const float *loc, const float *whs, const float *att, float *out, int n, int v,
                 int clip_first, float clip_x_max, float clip_y_max, int mfma_terms,
                 const dance_policy *policy, void *stream);

/* Fused gather + de_location + head SnakeBlock writing slot 0 of a P16 block (as
 * dance_gather_circconv_head; c % 64 == 0) on v_mfma_f32_16x16x32_f16: the ring is padded to a
 * multiple of 16 columns (196 vertices: 208), and the two de_location channels are one K step
 * (wd16, see dance_snake_net16.head_wd) instead of a zero-padded chunk through all 9 ring taps.
 * Rings above 128 vertices (and 65..128-vertex rings of few contours) run as two half rings in two
 * workgroups, each with 4 halo vertices either side.  feat_absmax (b, DANCE_ABSMAX_WORDS) or NULL:
 * the input is split at a per-contour power-of-two scale (see "Range" above); NULL = unscaled.
 * reference: snake_head.py:1542-1572, 1587-1610, 1639-1652, _SnakeNet.head :233. */
#define DANCE_HEAD16W_MAX_V 208
int dance_gather_head16w(const float *feat, const float *att, const float *loc,
                         const int32_t *image_idx, const void *w16, const void *wd16,
                         const float *bias_scaled, float w_inv_scale, float wd_inv_scale,
                         const float *scale, const float *shift, void *y16, int64_t y_stride,
                         float *att_out, int n, int v, int b, int hf, int wf, int c, int padding_mode,
                         int deloc_type, int mfma_terms, const uint32_t *feat_absmax,
                         const dance_policy *policy, void *stream);

/* Dsnake's init stage up to the init snake's head block in one launch (c = 256, 16 <= v <= 64):
 * bilinear gather at the ring's vertices and at its centre (min + max) / 2, fuse Conv1d(2c -> c,
 * k = 1) with the centre half folded into a per-contour row constant (fuse_w16: the sampled half
 * packed (2,1,32,256,8) with inverse scale fuse_w_inv_scale; fuse_wg: the centre half as [g][o]
 * (256,256) fp32; fuse_bias (256)), de_location, head SnakeBlock on 16-wide vertex tiles (head_w16 /
 * head_wd16 as in dance_gather_head16w) -> slot 0 of a P16 block.  The gathered features are split
 * at the image's scale (feat_absmax, may be NULL), the fused block at its own measured maximum.
 * reference: DsnakeHead.init dsnake_head.py:718-736 (+ get_locations_feature :678-695,
 * de_location :697-708, _SnakeNet.head :99, :120). */
int dance_init_head16(const float *feat, const float *loc, const int32_t *image_idx,
                      const void *fuse_w16, float fuse_w_inv_scale, const float *fuse_wg,
                      const float *fuse_bias, const void *head_w16, const void *head_wd16,
                      const float *head_bias_scaled, float head_w_inv_scale,
                      float head_wd_inv_scale, const float *head_scale, const float *head_shift,
                      void *y16, int64_t y_stride, int n, int v, int b, int hf, int wf, int c,
                      int padding_mode, int deloc_type, int mfma_terms, const uint32_t *feat_absmax,
                      void *stream);

/* DsnakeHead.fuse (dsnake_head.py:242, Conv1d(2c, c, 1)) packed for dance_init_head16. */
typedef struct dance_init_fuse16 {
    const void *w16;            /* sampled half, split-half packing (2,1,c/8,c,8) */
    const float *wg_center;     /* centre half, [g][o] (c, c) */
    const float *bias;          /* (c) */
    float w_inv_scale;
} dance_init_fuse16;

/* One _SnakeNet packed for the fast mode (dance_amd/snake.py SnakeNetWeights._pack16).  With the
 * state slot l kept at scale 2^(a_l) and prediction[0]'s output at 2^(a_p):
 *   head_scale / head_shift = folded BatchNorm * 2^(a_0);
 *   res_bs[l] = bias * 2^(s_l + a_l), res_winv[l] = 2^-(s_l + a_l), res_scale / res_shift[l] =
 *   folded BatchNorm * 2^(a_(l+1)), res_rmul[l] = 2^(a_(l+1) - a_l);
 *   fuse_w: column block j of both halves divided by 2^(a_j) before packing; fuse_winv = 2^-s,
 *   pred0_winv = 2^(a_p - s'); pred0_wg / pred0_b multiplied by 2^(a_p);
 *   pred1_bs = bias * 2^(s + a_p), pred1_winv = 2^-(s + a_p). */
typedef struct dance_snake_net16 {
    const void *head_w;  const float *head_bs, *head_scale, *head_shift;
    const void *res_w[7]; const float *res_bs[7], *res_scale[7], *res_shift[7];
    const void *fuse_w;  const float *fuse_bs, *pred0_wg, *pred0_b;   /* fuse_w: (2,1,128,512,8) */
    const void *pred1_w; const float *pred1_bs;
    const float *pred2_w, *pred2_b;
    float head_winv, res_winv[7], res_rmul[7], fuse_winv, pred0_winv, pred1_winv;
    int dilation[7];
    int feat_channels;
    const void *head_wd;        /* the head block's two de_location input channels as one 32-deep
                                   K step: (2,1,4,128,8) packing of W'[o][k], k = 9 ring taps of dx,
                                   9 of dy, 14 zeros (dance_gather_head16w) */
    float head_wd_winv;         /* head_wd's own inverse scale: the feature columns (head_w, whose
                                   de_location columns are zero) and the de_location columns are
                                   packed at separate powers of two */
} dance_snake_net16;

/* The 7 residual SnakeBlocks (res_* fields of the net) in one launch, v <= 256: state is the
 * (n, 8 slots) P16 block, slot 0 = head output (input), slots 1..7 receive the block outputs;
 * same results as 7 dance_conv16 calls up to summation order (1e-5), one kernel tail and launch
 * gap instead of 7.  The ring runs ceil(v / 16) vertex tiles of 16 (at most one more): LDS images
 * of 64, 96, 128, 160, 192, 224 or 256 columns; above 128 vertices (and for 97..128 with few
 * contours, policy) 8 waves per contour with the tiles split over two groups of waves, odd counts as
 * k + (k - 1).  coop_scratch: 16 + 2 n int32 words of device scratch for the two-workgroup form of
 * policy.coop_stack (zeroed here on the stream; word 0 = status, raised if a workgroup gave up waiting
 * for its partner after 2 s -- that contour then comes out NaN), or NULL: that form is never used.
 * reference: _SnakeNet.forward snake_head.py:262-268. */
int dance_res_stack16(const dance_snake_net16 *net_host, void *state, int64_t stride, int64_t slot,
                      int n, int v, int mfma_terms, const dance_policy *policy, int32_t *coop_scratch,
                      void *stream);

/* dance_snake_evolve in the fast precision mode (same workspace size, same stage events):
 * dance_gather_head16w -> dance_res_stack16 -> dance_fuse16w_raw(_split) -> dance_tail16.
 * feat_absmax: the features' per-image max words (dance_nchw_to_nhwc / dance_gn_relu /
 * dance_feature_absmax), or NULL = the head block's input is split unscaled. */
int dance_snake_evolve16(const dance_snake_net16 *net_host, const float *feat, const float *att,
                         const uint32_t *feat_absmax, const float *loc, const int32_t *image_idx,
                         const float *whs, float *out, void *workspace, int64_t workspace_bytes,
                         int n, int v, int b, int hf, int wf, int padding_mode, int deloc_type,
                         int individual_scale, int use_att, int clip_first, float clip_x_max,
                         float clip_y_max, int mfma_terms, const dance_policy *policy,
                         void **stage_events_host, void *stream);

/* Dsnake's whole init stage in the fast precision mode: dance_init_head16 with the init snake's
 * head block, then the residual stack, fusion and tail of dance_snake_evolve16 (no attention, no
 * individual scale, no clipping): out = loc + offsets, (n, v, 2).  Same workspace and stage
 * events as dance_snake_evolve16.  reference: DsnakeHead.init dsnake_head.py:718-744. */
int dance_snake_init16(const dance_snake_net16 *net_host, const dance_init_fuse16 *fuse_host,
                       const float *feat, const float *loc, const int32_t *image_idx, float *out,
                       void *workspace, int64_t workspace_bytes, int n, int v, int b, int hf, int wf,
                       int padding_mode, int deloc_type, int mfma_terms, const uint32_t *feat_absmax,
                       const dance_policy *policy, void **stage_events_host, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DANCE_HIP_H_ */
