"""GPU: polygon -> RLE (csrc/raster.hip) against the C oracle (oracle/rle_ref.c): run lengths,
run counts, areas, compact strings and decoded masks all bit-exact.  PARITY UNPINNED with
respect to pycocotools itself (see oracle/rle_ref.c); the oracle is checked against
hand-computed answers in tests/test_rle_oracle.py."""
import math

import numpy as np
import pytest
import torch

from dance_amd import rle
from dance_amd._lib import check, current_stream, lib, ptr
from oracle import rle_ref

pytestmark = pytest.mark.gpu


def star_polygons(n, v, h, w, seed, spill=0.0):
    """Clockwise star-shaped float32 polygons; `spill` > 0 lets them leave the image."""
    g = torch.Generator().manual_seed(seed)
    ang = torch.sort(torch.rand((n, v), generator=g) * 2 * math.pi, dim=1)[0]
    rmax = min(h, w) * 0.45
    r = (0.15 + 0.85 * torch.rand((n, v), generator=g)) * rmax * torch.rand((n, 1), generator=g).clamp_min(0.05)
    cx = (torch.rand((n, 1), generator=g) * (1 + 2 * spill) - spill) * w
    cy = (torch.rand((n, 1), generator=g) * (1 + 2 * spill) - spill) * h
    return torch.stack([cx + r * torch.cos(ang), cy + r * torch.sin(ang)], dim=2).float().contiguous()


def check_against_oracle(polys, h, w, device, cap=None, masks=True):
    out = rle.polygon_rle(polys.to(device), (h, w), cap=cap)
    counts = out.counts.cpu().numpy().view(np.uint32)
    m, area = out.m.cpu().numpy(), out.area.cpu().numpy().view(np.uint32)
    strings = out.strings()
    dec = out.decode().cpu().numpy() if masks else None
    for i, p in enumerate(polys.numpy()):
        want = rle_ref.fr_poly(p, h, w)
        assert m[i] == want.size, (i, m[i], want.size)
        assert np.array_equal(counts[i, :m[i]], want), i
        assert int(area[i]) == rle_ref.area(want), i
        assert strings[i] == rle_ref.to_string(want), i
        if masks:
            assert np.array_equal(dec[i], rle_ref.decode(want, h, w)), i
    return out


@pytest.mark.parametrize("v", [3, 4, 7, 128, 196, 256, 512])
def test_star_polygons(device, v):
    h, w = 200, 304
    check_against_oracle(star_polygons(24, v, h, w, seed=v), h, w, device)


@pytest.mark.parametrize("shape", [(37, 53), (480, 640), (800, 1216)])
def test_polygons_leaving_the_image(device, shape):
    h, w = shape
    check_against_oracle(star_polygons(32, 64, h, w, seed=h, spill=0.4), h, w, device, masks=h < 500)


def test_multi_band_polygons(device):
    """Wider than one 327,680-position LDS bitmap: processed in column bands."""
    for h, w in ((800, 1216), (1024, 2048)):
        big = torch.tensor([[[3.3, 5.1], [w - 2.7, 9.4], [w - 4.2, h - 3.3], [w / 2.0, h / 2.0], [6.6, h - 1.5]]])
        full = torch.tensor([[[0.0, 0.0], [float(w), 0.0], [float(w), float(h)], [w / 3.0, float(h)], [0.0, float(h)]]])
        zig = torch.stack([torch.linspace(0, w, 128), torch.where(torch.arange(128) % 2 == 0, 10.0, h - 10.0)], 1)[None]
        for p in (big, full):
            check_against_oracle(p.float(), h, w, device, masks=False)
        check_against_oracle(zig.float(), h, w, device, masks=False)
        check_against_oracle(star_polygons(6, 128, h, w, seed=3, spill=0.2) * 1.0, h, w, device, masks=False)


def test_degenerate_and_tie_cases(device):
    h, w = 60, 80
    cases = [
        [[10, 10], [20, 10], [20, 20], [10, 20]],                  # integer rectangle
        [[0, 0], [w, 0], [w, h], [0, h]],                          # whole image: y == h, x == w - 1 -> position h*w
        [[5.3, 5.3]] * 4,                                          # a point
        [[5, 5], [30, 5], [5, 5], [30, 5]],                        # a segment traced twice
        [[5, 5], [30, 25], [55, 45], [30, 25]],                    # collinear, there and back
        [[10.5, 10.5], [30.5, 10.5], [30.5, 30.5], [10.5, 30.5]],  # pixel centres
        [[10.1, 10.1], [30.3, 10.7], [30.9, 30.1], [10.7, 30.3]],  # 5x+.5 exactly integral: ties in the rounding
        [[10, 10], [40, 40], [40, 10], [10, 40]],                  # bow tie (self-intersection)
        [[-50, -50], [200, -50], [200, 200], [-50, 200]],          # encloses the image
        [[-30, 10], [-10, 10], [-10, 30], [-30, 30]],              # entirely left of the image
        [[100, 10], [120, 10], [120, 30], [100, 30]],              # entirely right of it
        [[10, -30], [30, -30], [30, 90], [10, 90]],                # crosses top and bottom
        [[79, 0], [80, 0], [80, 60], [79, 60]],                    # last column
    ]
    polys = torch.tensor(cases, dtype=torch.float32)
    out = check_against_oracle(polys, h, w, device)
    assert out.area.cpu().tolist()[:3] == [100, h * w, 0]
    # the same ring traced twice: every crossing pairs up, the mask is empty
    ring = star_polygons(1, 64, h, w, seed=9)
    twice = torch.cat([ring, ring], dim=1)
    out = check_against_oracle(twice, h, w, device)
    assert out.m.cpu().tolist() == [1] and out.area.cpu().tolist() == [0]
    # pentagram: even-odd filling
    k = torch.arange(5) * 2
    star = torch.stack([40 + 25 * torch.cos(k * 2 * math.pi / 5), 30 + 25 * torch.sin(k * 2 * math.pi / 5)], 1)[None]
    check_against_oracle(star.float(), h, w, device)


def test_capacity_overflow_is_reported_and_retried(device):
    h, w = 120, 160
    polys = star_polygons(8, 96, h, w, seed=21)
    n, v, _ = polys.shape
    p = polys.to(device)
    cap = 4
    counts = torch.full((n, cap), -1, dtype=torch.int32, device=device)
    m = torch.empty((n,), dtype=torch.int32, device=device)
    status = torch.empty((n,), dtype=torch.int32, device=device)
    check(lib().dance_polygon_rle(ptr(p), None, n, v, h, w, ptr(counts), cap, ptr(m), None, ptr(status),
                                  current_stream()), "polygon_rle")
    for i in range(n):
        want = rle_ref.fr_poly(polys[i].numpy(), h, w)
        assert int(m[i]) == want.size and int(status[i]) == (1 if want.size > cap else 0)
        assert np.array_equal(counts[i].cpu().numpy().view(np.uint32)[:min(cap, want.size)], want[:cap])
    check_against_oracle(polys, h, w, device, cap=4)          # the wrapper re-runs with room


def test_per_polygon_image_sizes(device):
    shapes = [(60, 80), (200, 304), (37, 53), (800, 1216)]
    polys = torch.cat([star_polygons(1, 32, h, w, seed=h, spill=0.1) for h, w in shapes])
    sizes = torch.tensor(shapes, dtype=torch.int32, device=device)
    n, v, cap = len(shapes), 32, 4096
    counts = torch.empty((n, cap), dtype=torch.int32, device=device)
    m = torch.empty((n,), dtype=torch.int32, device=device)
    area = torch.empty((n,), dtype=torch.int32, device=device)
    p = polys.to(device)
    check(lib().dance_polygon_rle(ptr(p), ptr(sizes), n, v, 0, 0, ptr(counts), cap, ptr(m), ptr(area), None,
                                  current_stream()), "polygon_rle")
    for i, (h, w) in enumerate(shapes):
        want = rle_ref.fr_poly(polys[i].numpy(), h, w)
        assert int(m[i]) == want.size
        assert np.array_equal(counts[i].cpu().numpy().view(np.uint32)[:want.size], want)
        assert int(area[i]) == rle_ref.area(want)


def test_get_polygon_rles_matches_the_reference_call_shape(device):
    """The reference's get_polygon_rles(polygons.flatten(), (h, w)) -> [{"size", "counts"}]
    (coco_evaluation.py:311-318) on deformed contours of the synthetic workload."""
    from dance_amd import ops, synthetic
    h, w = 800, 1216
    boxes = synthetic.make_boxes(64, h, w, 5)
    contours, _ = ops.uniform_upsample(boxes.to(device), 128, kind=ops.POLY_BOX)
    g = torch.Generator().manual_seed(1)
    contours = contours + 6.0 * torch.randn(contours.shape, generator=g).to(device)
    contours[..., 0].clamp_(0, w)
    contours[..., 1].clamp_(0, h)
    got = rle.get_polygon_rles(contours.flatten(1), (h, w))
    want = rle_ref.get_polygon_rles(contours.flatten(1).cpu().numpy(), (h, w))
    assert got == want
    assert rle.get_polygon_rles(contours[:0].flatten(1), (h, w)) == []


def test_wild_coordinates_terminate(device):
    """NaN / inf / absurd coordinates (a collapsed contour is NaN in the reference's `derange`)
    are clamped: the kernel finishes and still returns a valid run list."""
    h, w = 60, 80
    polys = torch.tensor([[[10.0, 10.0], [float("nan"), 20.0], [30.0, float("inf")], [5.0, 40.0]],
                          [[-1e30, 5.0], [1e30, 5.0], [1e30, 50.0], [-1e30, 50.0]],
                          [[float("nan")] * 2] * 4], dtype=torch.float32)
    out = rle.polygon_rle(polys.to(device), (h, w))
    counts, m = out.counts.cpu().numpy().view(np.uint32), out.m.cpu().numpy()
    for i in range(3):
        assert m[i] >= 1 and int(counts[i, :m[i]].sum()) == h * w
    # rows 5..49 of every column: the second polygon spans the whole width
    assert int(out.area[1]) == 45 * w


def test_errors(device):
    with pytest.raises(RuntimeError, match="GPU"):
        rle.polygon_rle(torch.zeros((1, 4, 2)), (10, 10))
    with pytest.raises(RuntimeError, match="vertices"):
        rle.polygon_rle(torch.zeros((1, 513, 2), device=device), (10, 10))
    with pytest.raises(RuntimeError, match="float32"):
        rle.polygon_rle(torch.zeros((1, 4, 2), device=device, dtype=torch.float64), (10, 10))
    with pytest.raises(RuntimeError, match="image shape"):
        rle.polygon_rle(torch.zeros((1, 4, 2), device=device), (0, 10))


def test_detector_postprocess_makes_the_masks_on_the_gpu(device):
    """dsnake_baseline/postprocessing.py:24-76 through dance_amd.postprocessing: boxes and contours
    scaled to the output resolution, clipped, empty boxes dropped, pred_masks = RLE dicts."""
    from _util import FakeBoxes, FakeInstances
    from dance_amd import postprocessing
    from dance_amd.structures import PolygonPoints
    h, w, oh, ow = 200, 304, 480, 640
    polys = star_polygons(9, 64, h, w, seed=4, spill=0.2)
    boxes = torch.cat([polys.min(1)[0], polys.max(1)[0]], 1)
    boxes[2] = torch.tensor([50.0, 60.0, 50.0, 90.0])                      # empty after clipping -> dropped
    scores = torch.arange(9, dtype=torch.float32)
    inst = FakeInstances((h, w), pred_boxes=FakeBoxes(boxes.to(device)), scores=scores.to(device),
                         pred_polys=PolygonPoints(polys.to(device)))
    out = postprocessing.detector_postprocess(inst, oh, ow)
    keep = [i for i in range(9) if i != 2]
    assert out.image_size == (oh, ow) and out.scores.cpu().tolist() == [float(i) for i in keep]
    want = polys[keep].clone()
    want[..., 0] *= ow / w
    want[..., 1] *= oh / h
    want[..., 0].clamp_(0, ow)
    want[..., 1].clamp_(0, oh)
    assert torch.equal(out.pred_polys.tensor.cpu(), want)
    assert out.pred_masks == rle_ref.get_polygon_rles(want.flatten(1).numpy(), (oh, ow))
    # the DANCE variant: no clipping of the contours, no masks, optional boxes from the contours
    inst = FakeInstances((h, w), pred_boxes=FakeBoxes(boxes.to(device)), scores=scores.to(device),
                         pred_polys=PolygonPoints(polys.to(device)))
    out = postprocessing.dance_detector_postprocess(False, 0.0, "poly", inst, oh, ow, 0, False, True)
    raw = polys[keep].clone()
    raw[..., 0] *= ow / w
    raw[..., 1] *= oh / h
    assert torch.equal(out.pred_polys.tensor.cpu(), raw) and not out.has("pred_masks")
    assert torch.equal(out.pred_boxes.tensor.cpu(), torch.cat([raw.min(1)[0], raw.max(1)[0]], 1))
