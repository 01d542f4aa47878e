"""GPU: the four extreme_utils ops and the fused resampler against the C oracle and the
golden vectors -- integer / index outputs and resampled coordinates bit-exact."""
import pytest
import torch

from _util import golden
from dance_amd import extreme_utils as ext
from dance_amd import ops
from oracle import extreme_utils_ref as cref
from test_oracle_golden import RESAMPLER_CASES

pytestmark = pytest.mark.gpu


def _collect_inputs(b, n, h, w, seed, int_scores=False):
    g = torch.Generator().manual_seed(seed)
    hm = torch.rand((b, 4, h, w), generator=g)
    if int_scores:                      # many exact ties -> exercises first-max-in-scan-order
        hm = torch.floor(hm * 4)
    vote = torch.randn((b, h, w, 2), generator=g)
    x0 = torch.randint(0, w // 2, (b, n), generator=g)
    y0 = torch.randint(0, h // 2, (b, n), generator=g)
    x1 = x0 + torch.randint(0, w // 2, (b, n), generator=g)
    y1 = y0 + torch.randint(0, h // 2, (b, n), generator=g)
    bbox = torch.stack([x0, y0, x1, y1], dim=2).long()
    radius = torch.randint(0, 6, (b, n), generator=g).long()
    ct = torch.stack([(x0 + x1) / 2.0, (y0 + y1) / 2.0], dim=2).float()
    # votes: half of the pixels point exactly at some centre, the rest are random
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    to_c = torch.stack([ct[:, 0, 0][:, None, None] - xs, ct[:, 0, 1][:, None, None] - ys], dim=-1)
    mask = torch.rand((b, h, w, 1), generator=g) < 0.5
    vote = torch.where(mask, to_c, vote).contiguous()
    return hm, bbox, radius, vote, ct


@pytest.mark.parametrize("shape", [(1, 1, 5, 5), (2, 7, 40, 56), (3, 33, 100, 152)])
@pytest.mark.parametrize("ties", [False, True])
def test_collect_extreme_point(device, shape, ties):
    b, n, h, w = shape
    inp = _collect_inputs(b, n, h, w, seed=b * 100 + n, int_scores=ties)
    want = cref.collect_extreme_point(*inp)
    got = ext.collect_extreme_point(*[t.to(device) for t in inp])
    assert got.dtype == torch.int64 and got.shape == (b, n, 4, 2)
    assert torch.equal(got.cpu(), want)


def test_collect_extreme_point_known_answer_and_errors(device):
    h = w = 5
    hm = torch.zeros((1, 4, h, w))
    hm[0, 0, 0, 3], hm[0, 1, 1, 0], hm[0, 2, 4, 1], hm[0, 3, 2, 4] = 2.0, 3.0, 5.0, 1.0
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    vote = torch.stack([2.0 - xs, 2.0 - ys], dim=-1)[None].contiguous()
    bbox = torch.tensor([[[0, 0, 4, 4]]], dtype=torch.int64)
    radius = torch.tensor([[1]], dtype=torch.int64)
    ct = torch.tensor([[[2.0, 2.0]]])
    args = [t.to(device) for t in (hm, bbox, radius, vote, ct)]
    assert ext.collect_extreme_point(*args).cpu().tolist() == [[[[3, 0], [0, 1], [1, 4], [4, 2]]]]
    args[3] = -args[3]
    assert ext.collect_extreme_point(*args).cpu().tolist() == [[[[2, 0], [0, 2], [2, 4], [4, 2]]]]
    with pytest.raises(RuntimeError, match="ext_hm must be a CUDA tensor"):
        ext.collect_extreme_point(hm, *args[1:])
    with pytest.raises(RuntimeError, match="same batch size"):
        ext.collect_extreme_point(args[0], args[1].repeat(2, 1, 1), *args[2:])
    # empty: n = 0
    out = ext.collect_extreme_point(args[0], args[1][:, :0], args[2][:, :0], args[3], args[4][:, :0])
    assert out.shape == (1, 0, 4, 2)


@pytest.mark.parametrize("E,p", [(4, 128), (12, 128), (12, 40), (7, 196)])
def test_edge_num_and_wnp(device, E, p):
    g = torch.Generator().manual_seed(E * 1000 + p)
    b, n = 2, 257
    length = torch.rand((b, n, E), generator=g) ** 3 + 1e-3
    en = torch.round(length * p / length.sum(2, keepdim=True)).long().clamp(min=1)
    en[0, 0] = 1                                   # big deficit
    en[0, 1] = p                                   # big surplus
    sort = torch.argsort(en, dim=2, descending=True, stable=True)
    want = en.clone()
    cref.calculate_edge_num(want, want.sum(2), sort, p)
    got = en.clone().to(device)
    ret = ext.calculate_edge_num(got, got.sum(2), sort.to(device), p)
    assert ret is None
    assert torch.equal(got.cpu(), want)
    assert torch.all(want.sum(2) == p)
    start = torch.cumsum(want, 2) - want
    w_want, i_want = cref.calculate_wnp(want, start, p)
    w_got, i_got = ext.calculate_wnp(got, start.to(device), p)
    assert w_got.shape == (b, n, p, 1) and i_got.shape == (b, n, p, 2)
    assert torch.equal(w_got.cpu(), w_want) and torch.equal(i_got.cpu(), i_want)
    with pytest.raises(RuntimeError, match="edge_num must be contiguous"):
        ext.calculate_edge_num(got.transpose(1, 2).contiguous().transpose(1, 2), got.sum(2), sort.to(device), p)
    with pytest.raises(RuntimeError, match="edge_num must be a CUDA tensor"):
        ext.calculate_edge_num(en, en.sum(2), sort, p)


@pytest.mark.parametrize("shape", [(1, 6, 2), (3, 128, 2), (5, 37, 9), (2, 1, 1)])
def test_roll_array(device, shape):
    b, n, d = shape
    g = torch.Generator().manual_seed(n)
    a = torch.randn(shape, generator=g)
    step = torch.randint(-3 * n, 3 * n, (b,), generator=g).long()
    want = cref.roll_array(a, step)
    got = ext.roll_array(a.to(device), step.to(device))
    assert torch.equal(got.cpu(), want)
    for i in range(b):
        assert torch.equal(want[i], torch.roll(a[i], int(step[i]), 0))
    with pytest.raises(RuntimeError, match="step must have the same batch size"):
        ext.roll_array(a.to(device), step[:0].to(device))


def test_fused_resampler_golden_bit_exact(device):
    g = golden("g1_resampler")
    for name, ps in RESAMPLER_CASES:
        for p in ps:
            pts, start = ops.uniform_upsample(g[name + "_poly"].to(device), p)
            assert torch.equal(start.cpu(), g["%s_p%d_start" % (name, p)]), (name, p)
            assert torch.equal(pts.cpu(), g["%s_p%d_points" % (name, p)]), (name, p)
    # polygon builders fused in: octagon from extreme points, boxes, quadrangles
    pts, start = ops.uniform_upsample(g["octagon_ext"].to(device), 128, kind=ops.POLY_OCTAGON)
    assert torch.equal(pts.cpu(), g["octagon_p128_points"]) and torch.equal(start.cpu(), g["octagon_p128_start"])


def test_fused_resampler_builders_and_large(device, monkeypatch):
    monkeypatch.delenv("DANCE_SYNC_CHECKS", raising=False)    # the immediate assert is what is tested
    from dance_amd import synthetic
    from oracle import snake_ref
    boxes = synthetic.make_boxes(3000, 800, 1216, seed=5)
    for kind, poly, p in ((ops.POLY_BOX, snake_ref.box_polygon(boxes), 128),
                          (ops.POLY_BOX, snake_ref.box_polygon(boxes), 196),
                          (ops.POLY_QUAD, snake_ref.quadrangle(boxes), 40)):
        want_pts, want_start, status = cref.uniform_upsample_fused(poly, p)
        assert int(status.sum()) == 0
        pts, start = ops.uniform_upsample(boxes.to(device), p, kind=kind)
        assert torch.equal(pts.cpu(), want_pts) and torch.equal(start.cpu(), want_start)
    # reference-style call through the 4-D (1,N,E,2) layout + empty input
    pts, start = ops.uniform_upsample(snake_ref.box_polygon(boxes)[None].to(device), 128)
    assert pts.shape == (3000, 128, 2)
    pts, start = ops.uniform_upsample(boxes[:0].to(device), 128, kind=ops.POLY_BOX)
    assert pts.shape == (0, 128, 2)
    # E > p_num cannot be satisfied: flagged instead of the reference's out-of-bounds walk
    with pytest.raises(AssertionError):
        ops.uniform_upsample(torch.rand((2, 12, 2)).to(device), 5)


def test_contour_upsample_2x_golden_bit_exact(device):
    g = golden("g7_upsample2x")
    for i in range(3):
        got = ops.contour_upsample_2x(g["loc%d" % i].to(device))
        assert torch.equal(got.cpu(), g["out%d" % i])
    big = torch.rand((3000, 128, 2)) * 1000
    from oracle import snake_ref
    assert torch.equal(ops.contour_upsample_2x(big.to(device)).cpu(), snake_ref.contour_upsample_2x(big))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 300, 2000])
@pytest.mark.parametrize("thresh", [0.3, 0.6])
def test_nms_vs_c_oracle(device, n, thresh):
    """kept index set bit-exact against the C restatement of nms.cu (+1 IoU, strict >)"""
    g = torch.Generator().manual_seed(n)
    ctr = torch.rand((n, 2), generator=g) * 300
    wh = torch.rand((n, 2), generator=g) * 120 + 4
    dets = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    scores = torch.rand((n,), generator=g)
    want = cref.nms(dets, scores, thresh)
    got = ext.nms(dets.to(device), scores.to(device), thresh)
    assert got.is_cuda and got.dtype == torch.int64
    assert torch.equal(got.cpu(), want)
    assert 1 <= len(want) <= n


def test_nms_known_answers(device):
    # identical boxes: only the best-scored survives; disjoint boxes: all survive
    same = torch.tensor([[10., 10., 50., 50.]]).repeat(5, 1)
    sc = torch.tensor([0.1, 0.9, 0.5, 0.3, 0.7])
    assert ext.nms(same.to(device), sc.to(device), 0.5).tolist() == [1]
    apart = torch.tensor([[0., 0., 9., 9.], [20., 0., 29., 9.], [0., 20., 9., 29.]])
    assert ext.nms(apart.to(device), sc[:3].to(device), 0.5).tolist() == [0, 1, 2]
    # "+1" convention: boxes [0,9] and [5,14] overlap 5 px of 10 -> IoU = 50/(100+100-50) = 1/3
    pair = torch.tensor([[0., 0., 9., 9.], [5., 0., 14., 9.]])
    assert ext.nms(pair.to(device), sc[:2].to(device), 0.34).tolist() == [0, 1]
    assert ext.nms(pair.to(device), sc[:2].to(device), 0.33).tolist() == [1]
    empty = ext.nms(torch.zeros((0, 4), device=device), torch.zeros((0,), device=device), 0.5)
    assert empty.numel() == 0 and not empty.is_cuda          # the reference's CPU-empty quirk


@pytest.mark.parametrize("n", [1, 63, 64, 65, 300, 1500])
def test_ml_nms_against_the_oracle(device, n):
    """dance_amd.ml_nms (the reference's core/layers/ml_nms.py:8-31 over detectron2's batched_nms)
    against the C restatement of torchvision's NMS: kept indices identical, in score order."""
    from _util import FakeBoxes, FakeInstances
    from dance_amd import ml_nms
    g = torch.Generator().manual_seed(n)
    ctr = torch.rand((n, 2), generator=g) * 300
    wh = torch.rand((n, 2), generator=g) * 80 + 4
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    scores = torch.rand((n,), generator=g)
    classes = torch.randint(0, 3, (n,), generator=g)
    for thr in (0.3, 0.6):
        want = cref.batched_nms(boxes, scores, classes, thr)
        got = ml_nms.batched_nms(boxes.to(device), scores.to(device), classes.to(device), thr)
        assert got.dtype == torch.int64 and torch.equal(got.cpu(), want)
    inst = FakeInstances((400, 400), pred_boxes=FakeBoxes(boxes.to(device)), scores=scores.to(device),
                         pred_classes=classes.to(device))
    out = ml_nms.ml_nms(inst, 0.6, max_proposals=10)
    want = cref.batched_nms(boxes, scores, classes, 0.6)[:10]
    assert torch.equal(out.scores.cpu(), scores[want]) and torch.equal(out.pred_boxes.tensor.cpu(), boxes[want])
    assert ml_nms.ml_nms(inst, 0.0) is inst
    assert ml_nms.batched_nms(boxes[:0].to(device), scores[:0].to(device), classes[:0].to(device), 0.5).numel() == 0


def test_batched_nms_many_equals_one_call_per_list(device):
    """The segmented NMS (one call for all images) returns, list by list, exactly what
    batched_nms returns for that list alone and what the C oracle returns: sizes around the
    64-box tile edge, an empty list, heavy overlap, repeated scores (ties keep input order)."""
    from dance_amd import ml_nms
    g = torch.Generator().manual_seed(17)
    boxes, scores, idxs = [], [], []
    for n in (300, 0, 1, 64, 65, 1000, 129):
        xy = torch.rand(n, 2, generator=g) * 200
        wh = torch.rand(n, 2, generator=g) * 80 + 4
        boxes.append(torch.cat([xy, xy + wh], 1))
        s = torch.rand(n, generator=g)
        if n >= 64:
            s[::7] = s[3]                                  # ties
        scores.append(s)
        idxs.append(torch.randint(0, 3, (n,), generator=g))
    many = ml_nms.batched_nms_many([b.to(device) for b in boxes], [s.to(device) for s in scores],
                                   [i.to(device) for i in idxs], 0.5)
    assert len(many) == len(boxes)
    for b, s, i, got in zip(boxes, scores, idxs, many):
        one = ml_nms.batched_nms(b.to(device), s.to(device), i.to(device), 0.5)
        assert torch.equal(got, one)
        assert torch.equal(got.cpu(), cref.batched_nms(b, s, i, 0.5))
    assert ml_nms.batched_nms_many([], [], [], 0.5) == []
    empty = ml_nms.batched_nms_many([boxes[1].to(device)] * 2, [scores[1].to(device)] * 2, [idxs[1].to(device)] * 2, 0.5)
    assert [e.numel() for e in empty] == [0, 0]


def test_octagon_upsample_clamp_and_override_in_one_launch(device):
    """ops.octagon_upsample = DsnakeHead.forward's clamp of the init snake's extreme points to the
    image (dsnake_head.py:800-805), the ground-truth override (:811-817) and get_octagon (:808-810)
    in one launch, reading `pred[:, ::10]` in place: bit-exact against torch.clamp / torch.where +
    the plain octagon resampler, NaN and out-of-image points included."""
    g = torch.Generator().manual_seed(5)
    n, h, w = 300, 400, 608
    ring = torch.rand((n, 40, 2), generator=g) * torch.tensor([w + 80.0, h + 80.0]) - 40.0   # in and out of the image
    # plausible extreme points: top, left, bottom, right of a box around a centre
    c = torch.rand((n, 2), generator=g) * torch.tensor([w * 1.0, h * 1.0])
    r = torch.rand((n, 2), generator=g) * 60 + 5
    ring[:, 0] = c - torch.stack([torch.zeros(n), r[:, 1]], 1)
    ring[:, 10] = c - torch.stack([r[:, 0], torch.zeros(n)], 1)
    ring[:, 20] = c + torch.stack([torch.zeros(n), r[:, 1]], 1)
    ring[:, 30] = c + torch.stack([r[:, 0], torch.zeros(n)], 1)
    ring[7, 20, 0] = float("nan")
    ring = ring.to(device)
    view = ring[:, ::10]
    gt = (torch.rand((n, 4, 2), generator=g) * 300 + 20).to(device)
    mask = (torch.rand(n, generator=g) < 0.3).to(device)
    lo = torch.zeros(2, device=device)
    hi = torch.tensor([w - 1.0, h - 1.0], device=device)
    for use_gt in (False, True):
        want_ext = torch.clamp(view, min=lo, max=hi)
        if use_gt:
            want_ext = torch.where(mask[:, None, None], gt, want_ext)
        want_pts, want_start = ops.uniform_upsample(want_ext.contiguous(), 128, kind=ops.POLY_OCTAGON, check_status=False)
        pts, used, start = ops.octagon_upsample(view, 128, clamp_max=(w - 1.0, h - 1.0),
                                                override=gt if use_gt else None,
                                                override_mask=mask if use_gt else None, check_status=False)
        assert torch.equal(torch.isnan(used), torch.isnan(want_ext))
        assert torch.equal(torch.nan_to_num(used), torch.nan_to_num(want_ext))
        assert torch.equal(start, want_start)
        assert torch.equal(torch.isnan(pts), torch.isnan(want_pts))
        assert torch.equal(torch.nan_to_num(pts), torch.nan_to_num(want_pts))
    # no clamp, dense input: the plain resampler
    dense = view.contiguous()
    a, used, s = ops.octagon_upsample(dense, 128, check_status=False)
    b, s2 = ops.uniform_upsample(dense, 128, kind=ops.POLY_OCTAGON, check_status=False)
    assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)) and torch.equal(s, s2)
    assert torch.equal(torch.nan_to_num(used), torch.nan_to_num(dense))
