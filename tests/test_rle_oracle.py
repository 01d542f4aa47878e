"""CPU: the C restatement of pycocotools' polygon -> RLE step (oracle/rle_ref.c) against
hand-computed known answers and geometric properties.  PARITY UNPINNED: the reference holds no
fixture for this step and pycocotools is not importable here (oracle/rle_ref.c header)."""
import numpy as np

from oracle import rle_ref


def _string_to_counts(s):
    """Independent decoder of the compact string (the inverse the library calls rleFrString)."""
    out, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(out) > 2:
            x += out[-2]
        out.append(x)
    return np.array(out, dtype=np.int64)


def test_rectangle_known_answer():
    # integer corners (10,10)-(20,20): pixels 10..19 in both axes, column-major runs
    c = rle_ref.fr_poly([10, 10, 20, 10, 20, 20, 10, 20], 30, 40)
    assert c.tolist() == [310] + [10, 20] * 9 + [10, 610]
    assert rle_ref.area(c) == 100
    m = rle_ref.decode(c, 30, 40)
    want = np.zeros((30, 40), np.uint8)
    want[10:20, 10:20] = 1
    assert np.array_equal(m, want)
    assert rle_ref.to_string(c) == b"f9:d000000000000000000^b0"


def test_whole_image_and_empty():
    h, w = 7, 9
    c = rle_ref.fr_poly([0, 0, w, 0, w, h, 0, h], h, w)
    assert c.tolist() == [0, h * w] and rle_ref.area(c) == h * w
    assert rle_ref.fr_poly([3.2, 3.2, 3.2, 3.2, 3.2, 3.2], h, w).tolist() == [h * w]       # a point
    assert rle_ref.fr_poly([-30, -30, -20, -30, -20, -20], h, w).tolist() == [h * w]        # outside
    assert int(rle_ref.decode(rle_ref.fr_poly([1, 1, 5, 1], h, w), h, w).sum()) == 0       # a segment


def test_orientation_start_vertex_and_integer_shift():
    rng = np.random.default_rng(5)
    h, w = 120, 160
    for _ in range(40):
        k = int(rng.integers(3, 40))
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        r = rng.uniform(5, 40, k)
        xy = np.stack([70 + r * np.cos(ang), 55 + r * np.sin(ang)], 1).astype(np.float32).astype(np.float64)
        c = rle_ref.fr_poly(xy, h, w)
        assert int(c.sum()) == h * w
        assert np.array_equal(c, rle_ref.fr_poly(xy[::-1], h, w))
        assert np.array_equal(c, rle_ref.fr_poly(np.roll(xy, 3, 0), h, w))
        m0 = rle_ref.decode(c, h, w)
        m1 = rle_ref.decode(rle_ref.fr_poly(xy + np.array([7.0, -4.0]), h, w), h, w)
        assert np.array_equal(m0[4:, :-7], m1[:-4, 7:])
        assert rle_ref.area(c) == int(m0.sum())
        assert np.array_equal(_string_to_counts(rle_ref.to_string(c)), c.astype(np.int64))


def test_mask_agrees_with_point_in_polygon_away_from_the_boundary():
    rng = np.random.default_rng(11)
    h, w = 90, 110
    ys, xs = np.mgrid[0:h, 0:w]
    px, py = xs + 0.5, ys + 0.5                     # pixel centres
    for _ in range(10):
        k = int(rng.integers(5, 24))
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        r = rng.uniform(12, 38, k)
        x, y = 55 + r * np.cos(ang), 45 + r * np.sin(ang)
        mask = rle_ref.decode(rle_ref.fr_poly(np.stack([x, y], 1), h, w), h, w).astype(bool)
        inside = np.zeros((h, w), bool)
        dist = np.full((h, w), 1e9)
        for i in range(k):
            x0, y0, x1, y1 = x[i], y[i], x[(i + 1) % k], y[(i + 1) % k]
            cond = (y0 > py) != (y1 > py)
            xint = x0 + (py - y0) * (x1 - x0) / np.where(y1 == y0, 1e-30, y1 - y0)
            inside ^= cond & (px < xint)
            t = np.clip(((px - x0) * (x1 - x0) + (py - y0) * (y1 - y0)) / ((x1 - x0) ** 2 + (y1 - y0) ** 2), 0, 1)
            dist = np.minimum(dist, np.hypot(px - (x0 + t * (x1 - x0)), py - (y0 + t * (y1 - y0))))
        far = dist > 1.0
        assert np.array_equal(mask[far], inside[far])


def test_get_polygon_rles_shape():
    polys = np.array([[10, 10, 20, 10, 20, 20, 10, 20], [1, 1, 5, 1, 5, 4, 1, 4]], dtype=np.float32)
    out = rle_ref.get_polygon_rles(polys, (30, 40))
    assert [o["size"] for o in out] == [[30, 40], [30, 40]]
    assert out[0]["counts"] == b"f9:d000000000000000000^b0" and isinstance(out[1]["counts"], bytes)


def _parity_restatement(xy, h, w):
    """The HIP kernel's formulation (csrc/raster.hip) in plain Python: every step of every edge is
    examined on its own, a crossing toggles its position, runs are read off the positions left
    with an odd count, and h*w always closes the last run.  Equal to the library's
    sort-difference-fold on every input is what lets the kernel use a bitmap instead of a sort."""
    import math
    xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
    k = len(xy)
    vx = [int(np.trunc(5.0 * x + .5)) for x in xy[:, 0]]
    vy = [int(np.trunc(5.0 * y + .5)) for y in xy[:, 1]]
    vx.append(vx[0])
    vy.append(vy[0])
    odd = set()
    for e in range(k):
        xs, xe, ys, ye = vx[e], vx[e + 1], vy[e], vy[e + 1]
        dx, dy = abs(xe - xs), abs(ys - ye)
        n = max(dx, dy)
        if n == 0:
            continue
        xmajor = dx >= dy
        back = (xmajor and xs > xe) or (not xmajor and ys > ye)
        if back:
            xs, xe, ys, ye = xe, xs, ye, ys
        s = (ye - ys) / dx if xmajor else (xe - xs) / dy

        def point(t):
            if xmajor:
                return t + xs, int(np.trunc(ys + s * t + .5))
            return int(np.trunc(xs + s * t + .5)), t + ys

        for d in range(1, n + 1):
            up, vp = point(n - d + 1 if back else d - 1)
            uc, vc = point(n - d if back else d)
            if uc == up:
                continue
            xd = ((uc if uc < up else uc - 1) + .5) / 5.0 - .5
            if math.floor(xd) != xd or xd < 0 or xd > w - 1:
                continue
            yd = math.ceil(min(max((min(vc, vp) + .5) / 5.0 - .5, 0), h))
            p = int(xd) * h + int(yd)
            if p < h * w:
                odd ^= {p}
    out, prev = [], 0
    for p in sorted(odd):
        out.append(p - prev)
        prev = p
    out.append(h * w - prev)
    return np.array(out, dtype=np.uint32)


def test_sort_and_fold_equals_position_parity():
    rng = np.random.default_rng(0)
    for it in range(60):
        h, w = [(60, 80), (37, 53), (120, 90)][it % 3]
        k = int(rng.integers(3, 24))
        mode = it % 5
        if mode == 0:
            xy = rng.uniform(-20, max(h, w) + 20, (k, 2))
        elif mode == 1:
            xy = rng.integers(-5, max(h, w) + 5, (k, 2)).astype(float)          # vertices on pixel corners
        elif mode == 2:
            xy = rng.integers(-5, max(h, w) + 5, (k, 2)) + 0.5                  # ... on pixel centres
        elif mode == 3:
            xy = np.round(rng.uniform(0, w, (k, 2)) * 5) / 5 + 0.1              # 5x + .5 integral: rounding ties
        else:
            xy = rng.uniform(0, w, (k, 2))
            xy = np.concatenate([xy, xy])                                       # traced twice: all pairs cancel
        xy = xy.astype(np.float32).astype(np.float64)
        assert np.array_equal(rle_ref.fr_poly(xy, h, w), _parity_restatement(xy, h, w)), (it, mode)
    for c, (h, w) in (([[0, 0], [80, 0], [80, 60], [0, 60]], (60, 80)), ([[79, 0], [80, 0], [80, 60], [79, 60]], (60, 80))):
        assert np.array_equal(rle_ref.fr_poly(c, h, w), _parity_restatement(c, h, w))
