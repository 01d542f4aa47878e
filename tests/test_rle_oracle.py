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
