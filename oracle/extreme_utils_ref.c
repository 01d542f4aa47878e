/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or called from
 * the product (dance_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker.
 *
 * Plain-C (host, scalar, single thread) restatement of the four contour-path device
 * routines of the reference's `extreme_utils` CUDA extension
 * (reference: core/layers/extreme_utils/src/utils.cu).  The CUDA source cannot be
 * compiled in this image (needs nvcc and the long-removed THC headers), so this file
 * restates the arithmetic index by index; every function cites the lines it follows.
 *
 * Parity status:
 *   calculate_edge_num / calculate_wnp: pinned -- exercised through the reference's own
 *     Python caller `uniform_upsample` (snake_head.py:764-794) when the golden vectors
 *     are generated (oracle/gen_golden.py), and cross-checked against the reference's
 *     NumPy twin `uniform_sample` (snake_head.py:661-725).
 *   collect_extreme_point / roll_array: "parity unpinned" -- no caller, test or fixture
 *     exists in the reference tree; the spec is this restatement of utils.cu (fp32,
 *     no fused multiply-add) plus hand-computed known answers in tests/.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* utils.cu:9-17.  float math throughout; the caller compares the result against a
 * double literal (0.9) or an int literal (0), see collect_band_rows/cols. */
static float cosine_to_centre(float cx, float cy, int64_t x, int64_t y, float u, float v)
{
    float dx = cx - (float)x;
    float dy = cy - (float)y;
    float n1 = sqrtf(u * u + v * v);
    float n2 = sqrtf(dx * dx + dy * dy);
    float dot = u * dx + v * dy;
    return dot / (n1 * n2);
}

/* utils.cu:28-69 (collect_tt): scan rows y_side-radius..y_side+radius, columns
 * x_min..x_max of one heat-map plane; skip pixels whose vote does not point at the
 * centre (cosine < 0.9, compared in double as the C++ literal demands); strict '<'
 * keeps the first maximum in scan order.  The start score is read at the side's
 * mid-point without any bounds check (utils.cu:40-42). */
static void scan_horizontal_side(const float *hm, const float *vote, const float *ct,
                                 int64_t *out, int64_t x_min, int64_t x_max,
                                 int64_t y_side, int64_t radius, int h, int w)
{
    int64_t best_x = (x_min + x_max) / 2;
    int64_t best_y = y_side;
    float best = hm[best_y * w + best_x];
    for (int64_t i = -radius; i <= radius; ++i) {
        int64_t y = y_side + i;
        if (y < 0 || y > h - 1)
            continue;
        for (int64_t x = x_min; x <= x_max; ++x) {
            float s = hm[y * w + x];
            float vx = vote[(y * w + x) * 2];
            float vy = vote[(y * w + x) * 2 + 1];
            if ((double)cosine_to_centre(ct[0], ct[1], x, y, vx, vy) < 0.9)
                continue;
            if (best < s) {
                best = s;
                best_x = x;
                best_y = y;
            }
        }
    }
    out[0] = best_x;
    out[1] = best_y;
}

/* utils.cu:72-111 (collect_ll): same for a vertical side; columns x_side-radius..
 * x_side+radius (outer loop), rows y_min..y_max (inner loop); threshold is 0. */
static void scan_vertical_side(const float *hm, const float *vote, const float *ct,
                               int64_t *out, int64_t x_side, int64_t y_min,
                               int64_t y_max, int64_t radius, int h, int w)
{
    (void)h;
    int64_t best_x = x_side;
    int64_t best_y = (y_min + y_max) / 2;
    float best = hm[best_y * w + best_x];
    for (int64_t i = -radius; i <= radius; ++i) {
        int64_t x = x_side + i;
        if (x < 0 || x > w - 1)
            continue;
        for (int64_t y = y_min; y <= y_max; ++y) {
            float s = hm[y * w + x];
            float vx = vote[(y * w + x) * 2];
            float vy = vote[(y * w + x) * 2 + 1];
            if (cosine_to_centre(ct[0], ct[1], x, y, vx, vy) < 0)
                continue;
            if (best < s) {
                best = s;
                best_x = x;
                best_y = y;
            }
        }
    }
    out[0] = best_x;
    out[1] = best_y;
}

/* utils.cu:114-153 + host wrapper 155-212.
 * ext_hm (b,4,h,w) f32; bbox (b,n,4) i64 truncated to int (utils.cu:136-139);
 * radius (b,n) i64; vote (b,h,w,2) f32; ct (b,n,2) f32; out (b,n,4,2) i64. */
void ref_collect_extreme_point(const float *ext_hm, const int64_t *bbox,
                               const int64_t *radius, const float *vote, const float *ct,
                               int64_t *extreme_point, int b, int n, int h, int w)
{
    for (int ib = 0; ib < b; ++ib)
        for (int in = 0; in < n; ++in)
            for (int dir = 0; dir < 4; ++dir) {
                const float *hm = ext_hm + ((int64_t)ib * 4 + dir) * h * w;
                const int64_t *bx = bbox + ((int64_t)ib * n + in) * 4;
                int x_min = (int)bx[0], y_min = (int)bx[1];
                int x_max = (int)bx[2], y_max = (int)bx[3];
                int64_t r = radius[(int64_t)ib * n + in];
                const float *vt = vote + (int64_t)ib * h * w * 2;
                const float *c = ct + ((int64_t)ib * n + in) * 2;
                int64_t *o = extreme_point + (((int64_t)ib * n + in) * 4 + dir) * 2;
                if (dir == 0)
                    scan_horizontal_side(hm, vt, c, o, x_min, x_max, y_min, r, h, w);
                else if (dir == 1)
                    scan_vertical_side(hm, vt, c, o, x_min, y_min, y_max, r, h, w);
                else if (dir == 2)
                    scan_horizontal_side(hm, vt, c, o, x_min, x_max, y_max, r, h, w);
                else
                    scan_vertical_side(hm, vt, c, o, x_max, y_min, y_max, r, h, w);
            }
}

/* utils.cu:215-255.  In-place fix-up of per-edge point counts so that they sum to
 * p_num.  edge_num (b,n,E) i64 mutated; edge_num_sum (b,n); edge_idx_sort (b,n,E). */
void ref_calculate_edge_num(int64_t *edge_num, const int64_t *edge_num_sum,
                            const int64_t *edge_idx_sort, int b, int n, int E,
                            int64_t p_num)
{
    for (int64_t poly = 0; poly < (int64_t)b * n; ++poly) {
        int64_t *cnt = edge_num + poly * E;
        const int64_t *order = edge_idx_sort + poly * E;
        int64_t sum = edge_num_sum[poly];
        if (sum == p_num)
            continue;
        if (sum < p_num) {
            cnt[order[0]] += p_num - sum;
            continue;
        }
        int64_t surplus = sum - p_num;
        int k = 0;
        /* the reference's `while (pass_num > 0)` runs off the array when every edge is
         * already down to one point (E > p_num: undefined behaviour there); stop at E */
        while (surplus > 0 && k < E) {
            int64_t e = order[k];
            if (cnt[e] > surplus) {
                cnt[e] -= surplus;
                surplus = 0;
            } else {
                surplus -= cnt[e] - 1;
                cnt[e] = 1;
                ++k;
            }
        }
    }
}

/* utils.cu:303-331.  weight (b,n,p_num,1) f32 and ind (b,n,p_num,2) i64 must arrive
 * zero-filled (utils.cu:350-351); only slots covered by an edge are written. */
void ref_calculate_wnp(const int64_t *edge_num, const int64_t *edge_start_idx,
                       float *weight, int64_t *ind, int b, int n, int E, int p_num)
{
    for (int64_t poly = 0; poly < (int64_t)b * n; ++poly)
        for (int e = 0; e < E; ++e) {
            int64_t k = edge_num[poly * E + e];
            int s = (int)edge_start_idx[poly * E + e];
            float *wq = weight + poly * p_num + s;
            int64_t *iq = ind + (poly * p_num + s) * 2;
            for (int64_t i = 0; i < k; ++i) {
                wq[i] = (float)i / (float)k;
                iq[2 * i] = e;
                iq[2 * i + 1] = (e + 1) % E;
            }
        }
}

/* utils.cu:377-401.  new[b][(n + step[b]) mod N][d] = old[b][n][d] with a positive
 * modulo; step is truncated to int (utils.cu:396).  new_array arrives zero-filled. */
void ref_roll_array(const float *array, const int64_t *step, float *new_array, int b,
                    int n, int d)
{
    for (int ib = 0; ib < b; ++ib) {
        int s = (int)step[ib];
        for (int in = 0; in < n; ++in) {
            int dst = ((in + s) % n + n) % n;
            memcpy(new_array + ((int64_t)ib * n + dst) * d,
                   array + ((int64_t)ib * n + in) * d, sizeof(float) * (size_t)d);
        }
    }
}

/* nms.cu:13-21 (devIoU, "+1" pixel convention, fp32, no contraction) and nms.cu:70-128
 * (nms_cuda): boxes (n,5) = x1,y1,x2,y2,score ALREADY sorted by descending score
 * (nms.cu:74-76 sorts before the kernel); greedy sweep in that order, a box is dropped when
 * its IoU with an earlier kept box is > thresh (strict).  keep receives the kept positions
 * (indices into the sorted order, ascending); returns their number. */
static float ref_iou(const float *a, const float *b)
{
    float left = a[0] > b[0] ? a[0] : b[0], right = a[2] < b[2] ? a[2] : b[2];
    float top = a[1] > b[1] ? a[1] : b[1], bottom = a[3] < b[3] ? a[3] : b[3];
    float width = right - left + 1, height = bottom - top + 1;
    width = width > 0.f ? width : 0.f;
    height = height > 0.f ? height : 0.f;
    float interS = width * height;
    float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
    float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
    return interS / (Sa + Sb - interS);
}

/* The same sweep with the IoU of torchvision.ops.nms -- what detectron2.layers.batched_nms, the
 * call behind the reference's ml_nms (core/layers/ml_nms.py:25, fcose_outputs.py:528), runs:
 * widths and heights WITHOUT the "+1" (continuous coordinates).  torchvision / detectron2 are
 * absent from the reference tree and from this image: restated from the published algorithm,
 * PARITY UNPINNED. */
static float ref_iou0(const float *a, const float *b)
{
    float left = a[0] > b[0] ? a[0] : b[0], right = a[2] < b[2] ? a[2] : b[2];
    float top = a[1] > b[1] ? a[1] : b[1], bottom = a[3] < b[3] ? a[3] : b[3];
    float width = right - left, height = bottom - top;
    width = width > 0.f ? width : 0.f;
    height = height > 0.f ? height : 0.f;
    float interS = width * height;
    float Sa = (a[2] - a[0]) * (a[3] - a[1]);
    float Sb = (b[2] - b[0]) * (b[3] - b[1]);
    return interS / (Sa + Sb - interS);
}

int ref_nms_sorted_plain(const float *boxes, int n, float thresh, int64_t *keep)
{
    int num = 0;
    unsigned char *removed = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int i = 0; i < n; ++i) {
        if (removed[i])
            continue;
        keep[num++] = i;
        for (int j = i + 1; j < n; ++j)
            if (ref_iou0(boxes + 5 * i, boxes + 5 * j) > thresh)
                removed[j] = 1;
    }
    free(removed);
    return num;
}

int ref_nms_sorted(const float *boxes, int n, float thresh, int64_t *keep)
{
    int num = 0;
    unsigned char *removed = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int i = 0; i < n; ++i) {
        if (removed[i])
            continue;
        keep[num++] = i;
        for (int j = i + 1; j < n; ++j)
            if (ref_iou(boxes + 5 * i, boxes + 5 * j) > thresh)
                removed[j] = 1;
    }
    free(removed);
    return num;
}

/* ------------------------------------------------------------------------------------
 * Fused resampler: restates RefineNet.uniform_upsample (snake_head.py:764-794) around
 * the two routines above, in the op order of the torch code (separately rounded fp32
 * mul/add, IEEE sqrt/div, round-half-even, stable descending argsort, edge lengths
 * summed left to right).  poly (N,E,2) f32 -> out (N,p,2) f32, edge_start_idx (N,E)
 * i64, status[i]=1 if polygon i could not be brought to p_num points (the reference
 * asserts instead, snake_head.py:782).  E <= 64.
 * ---------------------------------------------------------------------------------- */
void ref_uniform_upsample(const float *poly, int N, int E, int p_num, float *out,
                          int64_t *edge_start_idx, int32_t *status)
{
    for (int i = 0; i < N; ++i) {
        const float *P = poly + (int64_t)i * E * 2;
        float len[64];
        int64_t cnt[64], order[64], start[64];
        float total = 0.0f;
        for (int e = 0; e < E; ++e) {
            int nx = (e + 1) % E;                      /* torch.roll(poly,-1,2) :771 */
            float dx = P[nx * 2] - P[e * 2];
            float dy = P[nx * 2 + 1] - P[e * 2 + 1];
            float sx = dx * dx, sy = dy * dy;          /* .pow(2).sum(3).sqrt() :772 */
            len[e] = sqrtf(sx + sy);
            total = total + len[e];
        }
        int64_t sum = 0;
        for (int e = 0; e < E; ++e) {
            float q = (len[e] * (float)p_num) / total;  /* :773-775 */
            float r = nearbyintf(q);                    /* torch.round: half to even */
            /* .long() of NaN/inf is undefined in torch; keep the x86 value */
            int64_t c = isfinite(r) ? (int64_t)r : INT64_MIN;
            if (c < 1)
                c = 1;                                   /* clamp(min=1) :776 */
            cnt[e] = c;
            sum += c;
        }
        /* argsort(descending=True), ties keep the lower index first (:779) */
        for (int e = 0; e < E; ++e)
            order[e] = e;
        for (int a = 1; a < E; ++a) {
            int64_t key = order[a];
            int j = a - 1;
            while (j >= 0 && cnt[order[j]] < cnt[key]) {
                order[j + 1] = order[j];
                --j;
            }
            order[j + 1] = key;
        }
        ref_calculate_edge_num(cnt, &sum, order, 1, 1, E, p_num);
        int64_t acc = 0;
        for (int e = 0; e < E; ++e) {
            start[e] = acc;                              /* cumsum - edge_num :784 */
            acc += cnt[e];
            edge_start_idx[(int64_t)i * E + e] = start[e];
        }
        float *O = out + (int64_t)i * p_num * 2;
        if (acc != p_num) {
            status[i] = 1;
            memset(O, 0, sizeof(float) * 2 * (size_t)p_num);
            continue;
        }
        status[i] = 0;
        for (int e = 0; e < E; ++e) {
            int nx = (e + 1) % E;
            for (int64_t k = 0; k < cnt[e]; ++k) {
                float wgt = (float)k / (float)cnt[e];
                float om = 1.0f - wgt;                   /* poly1*(1-w) + poly2*w :792 */
                float x = P[e * 2] * om;
                float x2 = P[nx * 2] * wgt;
                float y = P[e * 2 + 1] * om;
                float y2 = P[nx * 2 + 1] * wgt;
                O[(start[e] + k) * 2] = x + x2;
                O[(start[e] + k) * 2 + 1] = y + y2;
            }
        }
    }
}
