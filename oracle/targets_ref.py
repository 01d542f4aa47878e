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
