/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or called by the product
 * (dance_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * CPU restatement of the polygon -> run-length-encoded mask step that follows the contour
 * path.  The reference calls it as
 *     mask_util.merge(mask_util.frPyObjects([p.tolist()], h, w))        one polygon at a time
 * (core/evaluation/coco_evaluation.py:311-318, core/modeling/postprocessing.py:266-274,
 * core/modeling/dsnake_baseline/postprocessing.py:13-21,71), i.e. the arithmetic lives in a
 * THIRD-PARTY dependency that is absent from /root/reference and from this image:
 * pycocotools, installed by the reference from cocoapi master, unpinned (README.md:27).
 * What is restated here is that library's published algorithm (cocoapi common/maskApi.c:
 * rleFrPoly, rleToString, rleDecode, rleArea; merge of a single RLE is a copy):
 *
 *   1. vertices (doubles) are scaled by 5 and rounded: (int)(5 * c + 0.5);
 *   2. every edge is walked one step at a time along its major axis, the minor coordinate
 *      being (int)(start + slope * t + 0.5) with slope = minor delta / major delta in double;
 *      edges that run backwards along their major axis are walked from their far end but
 *      indexed so the points still come out in polygon order;
 *   3. wherever two consecutive points differ in x, the crossing is mapped back to pixel
 *      units: xd = (x' + 0.5) / 5 - 0.5 with x' = the smaller x (right-to-left) or x - 1
 *      (left-to-right); only integral xd inside [0, w-1] count; yd = ceil(clamp((min y + 0.5)
 *      / 5 - 0.5, 0, h));
 *   4. crossings become column-major positions x * h + y, h * w is appended, the list is
 *      sorted, differenced, and zero-length runs are folded into their neighbours.
 *
 * PARITY UNPINNED: the reference holds no test, fixture or golden vector for this step and
 * pycocotools cannot be imported here, so nothing pins this restatement to the library's
 * actual output; tests/ checks it against hand-computed known answers and geometric
 * properties only.  Compile with -ffp-contract=off (the library's wheels are plain x86-64
 * builds: every double operation rounds separately).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int cmp_u32(const void *a, const void *b)
{
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x > y ? 1 : (x < y ? -1 : 0);
}

static int iabs(int a) { return a < 0 ? -a : a; }

/* xy: k vertices (x0,y0,x1,y1,...) in doubles.  Writes min(m, cap) counts and returns m. */
long ref_rle_fr_poly(const double *xy, long k, long h, long w, uint32_t *cnts, long cap)
{
    const double scale = 5.0;
    long j, total = 0, n = 0;
    if (k <= 0)
        return 0;
    int *vx = (int *)malloc(sizeof(int) * (size_t)(k + 1));
    int *vy = (int *)malloc(sizeof(int) * (size_t)(k + 1));
    for (j = 0; j < k; ++j) {
        vx[j] = (int)(scale * xy[2 * j + 0] + .5);
        vy[j] = (int)(scale * xy[2 * j + 1] + .5);
    }
    vx[k] = vx[0];
    vy[k] = vy[0];
    for (j = 0; j < k; ++j) {
        const int ax = iabs(vx[j] - vx[j + 1]), ay = iabs(vy[j] - vy[j + 1]);
        total += (ax > ay ? ax : ay) + 1;
    }
    /* dense boundary, in polygon order */
    int *u = (int *)malloc(sizeof(int) * (size_t)total);
    int *v = (int *)malloc(sizeof(int) * (size_t)total);
    for (j = 0; j < k; ++j) {
        int xs = vx[j], xe = vx[j + 1], ys = vy[j], ye = vy[j + 1];
        const int dx = iabs(xe - xs), dy = iabs(ys - ye);
        const int backwards = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
        int d, t;
        if (backwards) {
            t = xs; xs = xe; xe = t;
            t = ys; ys = ye; ye = t;
        }
        const double slope = dx >= dy ? (double)(ye - ys) / dx : (double)(xe - xs) / dy;
        if (dx >= dy) {
            for (d = 0; d <= dx; ++d) {
                t = backwards ? dx - d : d;
                u[n] = t + xs;
                v[n] = (int)(ys + slope * t + .5);
                ++n;
            }
        } else {
            for (d = 0; d <= dy; ++d) {
                t = backwards ? dy - d : d;
                v[n] = t + ys;
                u[n] = (int)(xs + slope * t + .5);
                ++n;
            }
        }
    }
    free(vx);
    free(vy);
    /* column crossings, back in pixel units */
    uint32_t *a = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n + 1));
    long c = 0;
    for (j = 1; j < n; ++j) {
        if (u[j] == u[j - 1])
            continue;
        double xd = (double)(u[j] < u[j - 1] ? u[j] : u[j] - 1);
        xd = (xd + .5) / scale - .5;
        if (floor(xd) != xd || xd < 0 || xd > (double)(w - 1))
            continue;
        double yd = (double)(v[j] < v[j - 1] ? v[j] : v[j - 1]);
        yd = (yd + .5) / scale - .5;
        if (yd < 0)
            yd = 0;
        else if (yd > (double)h)
            yd = (double)h;
        yd = ceil(yd);
        a[c++] = (uint32_t)((int)xd * (int)h + (int)yd);
    }
    free(u);
    free(v);
    a[c++] = (uint32_t)(h * w);
    qsort(a, (size_t)c, sizeof(uint32_t), cmp_u32);
    uint32_t prev = 0;
    for (j = 0; j < c; ++j) {
        const uint32_t t = a[j];
        a[j] -= prev;
        prev = t;
    }
    /* fold zero-length runs */
    uint32_t *b = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)c);
    long m = 0;
    j = 0;
    b[m++] = a[j++];
    while (j < c) {
        if (a[j] > 0) {
            b[m++] = a[j++];
        } else {
            ++j;
            if (j < c)
                b[m - 1] += a[j++];
        }
    }
    for (j = 0; j < m && j < cap; ++j)
        cnts[j] = b[j];
    free(a);
    free(b);
    return m;
}

/* The library's compact ASCII form of the counts (a LEB128 variant: 5 data bits + 1
 * continuation bit per character, offset 48; from the fourth count on the difference to the
 * count two places back is stored).  s needs 7 * m + 1 bytes.  Returns the length. */
long ref_rle_to_string(const uint32_t *cnts, long m, char *s)
{
    long i, p = 0;
    for (i = 0; i < m; ++i) {
        long x = (long)cnts[i];
        int more = 1;
        if (i > 2)
            x -= (long)cnts[i - 2];
        while (more) {
            char ch = (char)(x & 0x1f);
            x >>= 5;
            more = (ch & 0x10) ? x != -1 : x != 0;
            if (more)
                ch |= 0x20;
            s[p++] = (char)(ch + 48);
        }
    }
    s[p] = 0;
    return p;
}

/* Column-major (Fortran order, as the library's decode) h*w byte mask. */
void ref_rle_decode(const uint32_t *cnts, long m, uint8_t *mask, long hw)
{
    long j, pos = 0;
    uint8_t val = 0;
    for (j = 0; j < m; ++j) {
        long r = (long)cnts[j];
        while (r-- > 0 && pos < hw)
            mask[pos++] = val;
        val = (uint8_t)!val;
    }
}

uint32_t ref_rle_area(const uint32_t *cnts, long m)
{
    uint32_t a = 0;
    long j;
    for (j = 1; j < m; j += 2)
        a += cnts[j];
    return a;
}
