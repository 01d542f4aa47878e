"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by dance_amd/).

CPU restatement of the NumPy half of the reference's training-side target matching
(SURVEY 8f row 4), written as explicit loops with every rounding spelled out:

  uniform_sample                  core/modeling/edge_snake/snake_head.py:661-725
  uniform_sample_1d               snake_head.py:728-762
  get_aux_extreme_points          core/modeling/fcose/utils.py:94-189
  single_uniform_segment_matching snake_head.py:1265-1384 (INITIAL = "octagon" form)
  single_uniform_multisegment_matching snake_head.py:1386-1498 (INITIAL = "box", the shipped form)
  reorder_perloss                 snake_head.py:1520-1540 (the fall-back of the matching)

Pinned: oracle/gen_golden.py `targets` runs the reference's own functions (loaded from
/root/reference) on seeded polygons, asserts this restatement equal to them (indices and
float64 coordinates bit for bit) and writes tests/golden/g12_targets.npz.

Arithmetic = what the reference's lines compute under NumPy 2.2 (this image): float32 edge
lengths summed in NumPy's pairwise order, `float32 array / np.int32 scalar` promoted to float64
(so resampled points are float64), float64 everywhere downstream.  One deliberate definition:
`np.argsort` of the edge lengths (snake_head.py:668) is an unstable sort whose tie order is an
implementation detail of NumPy's SIMD quicksort; here ties go to the lowest index (stable), and
the fixtures avoid exact ties at the positions that matter.
"""
import numpy as np

F32 = np.float32


def pairwise_sum(a):
    """np.sum of a contiguous 1-D float array: NumPy's pairwise summation (blocks of at most
    128 elements, 8 running sums each, halves split at multiples of 8)."""
    dt = a.dtype.type
    n = len(a)
    if n < 8:
        r = dt(0.0)
        for x in a:
            r = dt(r + x)
        return r
    if n <= 128:
        r = [dt(a[j]) for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = dt(r[j] + a[i + j])
            i += 8
        res = dt(dt(dt(r[0] + r[1]) + dt(r[2] + r[3])) + dt(dt(r[4] + r[5]) + dt(r[6] + r[7])))
        while i < n:
            res = dt(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return dt(pairwise_sum(a[:n2]) + pairwise_sum(a[n2:]))


def edge_lengths32(pts):
    """float32 length of edge i -> i + 1 (cyclic), snake_head.py:665-667."""
    p = len(pts)
    out = np.empty(p, F32)
    for i in range(p):
        j = (i + 1) % p
        dx = F32(pts[j, 0] - pts[i, 0])
        dy = F32(pts[j, 1] - pts[i, 1])
        out[i] = np.sqrt(F32(F32(dx * dx) + F32(dy * dy)))
    return out


def uniform_sample_counts(pts, newpnum):
    """Points per edge (int32) for the `add points` branch, snake_head.py:682-709."""
    p = len(pts)
    el = edge_lengths32(pts)
    order = np.argsort(el, kind="stable")
    total = pairwise_sum(el)
    num = np.empty(p, np.int32)
    for i in range(p):
        q = F32(F32(el[i] * F32(newpnum)) / total)
        num[i] = np.int32(np.rint(q))                        # half to even, like np.round
        if num[i] == 0:
            num[i] = 1
    s = int(num.sum())
    if s > newpnum:
        k = p - 1
        rest = s - newpnum
        while rest > 0:
            e = order[k]
            if num[e] > rest:
                num[e] -= rest
                rest = 0
            else:
                rest -= num[e] - 1
                num[e] = 1
                k -= 1
    elif s < newpnum:
        num[order[p - 1]] += newpnum - s
    return num, order


def uniform_sample(pts, newpnum):
    """pts (P, 2) float32 -> (newpnum, 2): float32 rows of pts when P > newpnum (the shortest
    edges' start points dropped), else float64 points spread uniformly over the edges."""
    pts = np.asarray(pts, F32)
    p = len(pts)
    if p > newpnum:
        order = np.argsort(edge_lengths32(pts), kind="stable")
        keep = np.sort(order[p - newpnum:])
        return pts[keep]
    num, _ = uniform_sample_counts(pts, newpnum)
    out = np.empty((newpnum, 2), np.float64)
    o = 0
    for i in range(p):
        j = (i + 1) % p
        k = int(num[i])
        for m in range(k):
            w = np.float64(F32(m)) / np.float64(k)
            for c in range(2):
                out[o, c] = np.float64(pts[i, c]) * (np.float64(1.0) - w) + np.float64(pts[j, c]) * w
            o += 1
    assert o == newpnum
    return out


def uniform_sample_1d(pts, new_n):
    """Open poly-line pts (n, 2) float64 -> new_n points at equal arc-length steps starting at
    pts[0] (the end point is not emitted), snake_head.py:728-762.  Returns None where the
    reference raises (a degenerate poly-line: some mark lies beyond every node)."""
    pts = np.asarray(pts, np.float64)
    if new_n == 1:
        return pts[:1].copy()
    n = len(pts)
    if n == new_n + 1:
        return pts[:-1].copy()
    seg = np.empty(n - 1, np.float64)
    for i in range(n - 1):
        dx = pts[i + 1, 0] - pts[i, 0]
        dy = pts[i + 1, 1] - pts[i, 1]
        seg[i] = np.sqrt(dx * dx + dy * dy)
    start = np.empty(n, np.float64)
    acc = np.float64(0.0)
    start[0] = acc
    for i in range(n - 1):                                   # np.cumsum: sequential
        acc = acc + seg[i]
        start[i + 1] = acc
    total = pairwise_sum(seg)                                # np.sum: pairwise
    per = total / np.float64(new_n)
    out = np.empty((new_n, 2), np.float64)
    out[0] = pts[0]
    for i in range(new_n - 1):
        mark = np.float64(i + 1) * per
        a = -1
        for j in range(n):                                   # first node beyond the mark
            if start[j] - mark > 0:
                a = j
                break
        if a < 0:
            return None
        b = a - 1                                            # a >= 1: start[0] - mark < 0
        before = start[b] - mark
        after = start[a] - mark
        w = -before / (after - before)
        for c in range(2):
            out[i + 1, c] = (np.float64(1.0) - w) * pts[b, c] + w * pts[a, c]
    return out


def _sqdist(p, q):
    dx = p[0] - q[0]
    dy = p[1] - q[1]
    return dx * dx + dy * dy


def _argmin_sq(cands, q):
    """first index of the smallest squared distance (np.argmin semantics, NaN first)."""
    best, bi = None, 0
    for i in range(len(cands)):
        d = _sqdist(cands[i], q)
        if d != d:
            return i
        if best is None or d < best:
            best, bi = d, i
    return bi


def get_aux_extreme_points(pts):
    """pts (N, 2) float64, a closed clockwise contour.  Clusters of near-extreme points on each
    side (top, bottom, left, right: bands of 2 % of the extent, neighbours within 2 %) give
    auxiliary extreme points; the contour is rolled to start at the point nearest the first of
    them.  Returns (re_ordered_pts (N, 2), ext_idxs sorted unique int64), fcose/utils.py:94-189."""
    pts = np.asarray(pts, np.float64)
    n = len(pts)
    l, t = pts[:, 0].min(), pts[:, 1].min()
    r, b = pts[:, 0].max(), pts[:, 1].max()
    thresh = 0.02
    w = r - l + 1
    h = b - t + 1
    aux = []

    def side(axis, lo_side, edge, extent):
        other = 1 - axis
        if lo_side:
            band = [i for i in range(n) if (pts[i, axis] - edge) <= thresh * extent]
        else:
            band = [i for i in range(n) if (edge - pts[i, axis]) <= thresh * extent]
        while band:
            # np.argmin / np.argmax over the band: first extreme in band order
            k = band[0]
            for i in band[1:]:
                if (pts[i, axis] < pts[k, axis]) if lo_side else (pts[i, axis] > pts[k, axis]):
                    k = i
            idxs = [k]
            for step in (1, -1):
                tmp = (k + step) % n
                while tmp != k and ((pts[tmp, axis] - pts[k, axis]) if lo_side
                                    else (pts[k, axis] - pts[tmp, axis])) <= thresh * extent:
                    idxs.append(tmp)
                    tmp = (tmp + step) % n
            vals = [pts[i, other] for i in idxs]
            mid = (max(vals) + min(vals)) / 2
            aux.append((mid, edge) if axis == 1 else (edge, mid))
            band = [i for i in band if i not in idxs]

    side(1, True, t, h)
    side(1, False, b, h)
    side(0, True, l, w)
    side(0, False, r, w)
    aux = np.array(aux, np.float64)
    shift = _argmin_sq(pts, aux[0])
    re = np.roll(pts, -shift, axis=0)
    ext = np.array([_argmin_sq(re, a) for a in aux], np.int64)
    ext[0] = 0
    return re, np.unique(ext)


def reorder_perloss(targets, pts):
    """The cyclic shift of targets (V, 2) float64 that minimises the summed squared distance to
    pts (V, 2) float32 -- first minimum (snake_head.py:1520-1540, one instance)."""
    v = len(pts)
    best, bs = None, 0
    for s in range(v):
        tot = np.float64(0.0)
        for i in range(v):
            tot = tot + _sqdist(targets[(s + i) % v], pts[i].astype(np.float64))
        if best is None or tot < best:
            best, bs = tot, s
    return np.roll(targets, -bs, axis=0), bs


def single_uniform_segment_matching(dense, sampled, edge_idx, up_rate, num_sampling):
    """dense (V * up_rate, 2) float64 ordered ground-truth points, sampled (V, 2) float32 initial
    octagon contour, edge_idx (12,) start indices of the octagon's edges.  The four extreme
    vertices (every third edge start) are matched to their nearest dense points; each of the four
    arcs between them is resampled to as many points as the contour has on that arc.
    Returns (targets (V, 2) float64, status): status 0 = segments, 1 = the reference's fall-back
    (reorder_perloss of dense[::up_rate]), snake_head.py:1265-1384."""
    dense = np.asarray(dense, np.float64)
    v, big = num_sampling, num_sampling * up_rate
    ext = [int(e) for e in edge_idx[::3]]
    ch = [sampled[e].astype(np.float64) for e in ext]
    aug_ext = sorted(ext + [v])
    mn = [_argmin_sq(dense, c) for c in ch] + [big]
    if mn[3] < mn[1]:
        mn[3] = big - 2
    if mn[0] != 0:
        mn[0] = 0
    mn.sort()
    if mn[2] == big - 1:
        mn[2], mn[3] = big - 3, big - 2
    if mn[3] == big - 1:
        mn[3] -= 1
        mn[2] -= 1
    segs, ok = [], True
    for i in range(4):
        cnt = aug_ext[i + 1] - aug_ext[i]
        if cnt == 0:
            continue
        if mn[i + 1] - mn[i] <= 0:
            mn[i + 1] = mn[i] + 1
        if i == 3:
            arc = np.concatenate([dense[mn[i]:], dense[:1]], axis=0)
        else:
            arc = dense[mn[i]:mn[i + 1] + 1]
        if len(arc) < 2 and cnt != 1:                        # the reference's indexing raises here
            ok = False
            break
        s = uniform_sample_1d(arc, cnt)
        if s is None:
            ok = False
            break
        segs.append(s)
    if ok:
        out = np.concatenate(segs, axis=0)
        if len(out) == v:
            return out, 0
    return reorder_perloss(dense[::up_rate], sampled)[0], 1


def roll_to_first_target(sampled, dense):
    """The initial contour rolled to start at the vertex nearest the first dense target
    (snake_head.py:1150-1155).  sampled (V, 2) float32."""
    d0 = np.asarray(dense[0], np.float64)
    tt = _argmin_sq([s.astype(np.float64) for s in sampled], d0)
    return np.roll(sampled, -tt, axis=0)


def single_uniform_multisegment_matching(dense, sampled, ext_idx, up_rate, num_sampling):
    """The matching of the shipped configuration (INITIAL = box, NEW_MATCHING): ext_idx = the
    auxiliary extreme points' indices into dense (sorted, unique, first 0), matched to their
    nearest vertices of the (already rolled) contour; every arc between consecutive ones is
    resampled to the contour's vertex count on it.  Returns (targets (V, 2) float64, status) as
    single_uniform_segment_matching; snake_head.py:1386-1498."""
    dense = np.asarray(dense, np.float64)
    v, big = num_sampling, num_sampling * up_rate
    mn = [int(e) for e in ext_idx]
    k = len(mn)
    samp64 = [s.astype(np.float64) for s in sampled]
    ext = [_argmin_sq(samp64, dense[m]) for m in mn]
    if ext[0] != 0:
        ext[0] = 0
    if ext[-1] < ext[1]:
        ext[-1] = v - 2
    aug_ext = sorted(sorted(ext) + [v])
    mn = mn + [big]
    if mn[-1] < mn[1]:
        mn[-1] = big - 2
    if mn[0] != 0:
        mn[0] = 0
    mn.sort()
    if mn[-2] == big - 1:
        mn[-2], mn[-1] = big - 3, big - 2
    if mn[-1] == big - 1:
        mn[-1] -= 1
        mn[-2] -= 1
    segs, ok = [], True
    for i in range(k):
        cnt = aug_ext[i + 1] - aug_ext[i]
        if cnt == 0:
            continue
        if mn[i + 1] - mn[i] <= 0:
            mn[i + 1] = mn[i] + 1
        if i == k - 1:
            arc = np.concatenate([dense[mn[i]:], dense[:1]], axis=0)
        else:
            arc = dense[mn[i]:mn[i + 1] + 1]
        if len(arc) < 2 and cnt != 1:
            ok = False
            break
        s = uniform_sample_1d(arc, cnt)
        if s is None:
            ok = False
            break
        segs.append(s)
    if ok:
        out = np.concatenate(segs, axis=0)
        if len(out) == v:
            return out, 0
    return reorder_perloss(dense[::up_rate], sampled)[0], 1


# --------------------------------------------------------------------------- the driver
def ring_is_ccw(cnt):
    """`Polygon(cnt).exterior.is_ccw` as the reference calls it (snake_head.py:1101-1125).
    Shapely (1.7.1 pinned, requirements.txt:1) is NOT under /root/reference and not importable
    here: this restates its published shapely/algorithms/cga.py -- signed_area(ring) =
    sum_i x_i * (y_{i+1} - y_{i-1}) / 2 over the CLOSED ring as a sequential Python float sum,
    is_ccw = signed_area >= 0.  PARITY OF THIS PREDICATE IS UNPINNED (no fixture can exist here);
    for polygons of non-zero area every formulation of the orientation agrees."""
    pts = [(float(x), float(y)) for x, y in np.asarray(cnt).reshape(-1, 2)]
    if pts[0] != pts[-1]:
        pts.append(pts[0])
    xs = [q[0] for q in pts]
    ys = [q[1] for q in pts]
    xs.append(xs[1])
    ys.append(ys[1])
    return sum(xs[i] * (ys[i + 1] - ys[i - 1]) for i in range(1, len(pts))) / 2.0 >= 0.0


def compute_targets_for_polys_box(polygons_per_image, boxes_per_image, image_sizes, num_sampling, up_rate=5):
    """RefineNet.compute_targets_for_polys (snake_head.py:982-1254) for the shipped configuration:
    INITIAL = "box", ORIGINAL off, NEW_MATCHING on (up_rate 5, :997-1000), REORDER_METHOD != "curvegcn",
    JITTERING 0.  polygons_per_image[i][j] = the instance's list of flat (2 P) arrays
    (detectron2 PolygonMasks.polygons, :967-980), boxes_per_image[i] (n_i, 4) float32 xyxy tensor,
    image_sizes[i] = (h, w).  Returns the reference's dict (torch CPU tensors)."""
    import torch

    from . import snake_ref
    locs, tars, dense_t, image_index, scales, whs, init_locs, init_tars = [], [], [], [], [], [], [], []
    for im_i, (polys_im, bboxes) in enumerate(zip(polygons_per_image, boxes_per_image)):
        if bboxes.numel() == 0:                                                  # :1011
            continue
        h_img, w_img = image_sizes[im_i]
        ws = bboxes[:, 2] - bboxes[:, 0]
        hs = bboxes[:, 3] - bboxes[:, 1]
        box = snake_ref.box_polygon(bboxes)                                      # :1074-1077
        octagons = snake_ref.uniform_upsample(box[None], num_sampling)[0]        # :1078-1080
        init_box = snake_ref.uniform_upsample(box[None], 40)[0]                  # :1083
        ex_pts = init_box                                                        # :1084
        for j, polys in enumerate(polys_im):
            contour = [np.array(x).reshape(-1, 2).astype(np.float32) for x in polys]   # :973-975
            if len(contour) > 1:                                                 # fragmented, :976-977
                continue
            cnt = contour[0]
            oct_t, in_box, ex_tar, w, h = octagons[j], init_box[j], ex_pts[j], ws[j], hs[j]
            scale = torch.min(w, h)
            cnt = cnt[::-1] if ring_is_ccw(cnt) else cnt                         # :1101
            if ring_is_ccw(cnt):                                                 # :1105-1106
                continue
            oct_pts = oct_t.numpy()
            assert not ring_is_ccw(oct_pts), "1) contour must be clock-wise!"    # :1118-1120
            assert not ring_is_ccw(in_box.numpy()), "0) init box must be clock-wise!"
            big = uniform_sample(cnt, len(cnt) * num_sampling * up_rate)         # :1134-1136
            tt = int(np.argmin(np.power(big - oct_pts[0], 2).sum(axis=1)))       # :1139-1141
            dense = np.roll(big, -tt, axis=0)[::len(cnt)]                        # :1142-1144
            dense, ext = get_aux_extreme_points(dense)                           # :1147-1149
            oct_pts = roll_to_first_target(oct_pts, dense)                       # :1150-1155
            oct_t = torch.from_numpy(oct_pts)                                    # :1156
            matched, _status = single_uniform_multisegment_matching(dense, oct_pts, ext, up_rate, num_sampling)
            matched = torch.tensor(matched).float()                              # :1160-1162
            matched[..., 0].clamp_(min=0, max=w_img - 1)                         # :1172-1173
            matched[..., 1].clamp_(min=0, max=h_img - 1)
            locs.append(oct_t)
            dense_t.append(matched)                                              # :1175: the SAME tensor
            tars.append(matched)
            image_index.append(im_i)
            scales.append(scale)
            whs.append([w, h])
            init_locs.append(in_box)
            init_tars.append(ex_tar)
    return {"sample_locs": torch.stack(locs, 0), "sample_targets": torch.stack(tars, 0),
            "sample_dense_targets": torch.stack(dense_t, 0), "scales": torch.stack(scales, 0),
            "whs": torch.tensor(whs), "image_idx": torch.tensor(image_index),
            "init_locs": torch.stack(init_locs, 0), "init_targets": torch.stack(init_tars, 0)}
