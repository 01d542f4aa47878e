"""GPU parity of the FCOSE box decode (csrc/fcose.hip + dance_amd/fcose_outputs.py) against the
golden fixture made by the reference's own FCOSEOutputs (tests/golden/g11_fcose.npz) and against
the CPU restatement oracle/fcose_ref.py.  Boxes, locations, classes and extreme points are
bit-exact; scores (sigmoid, sqrt) within 1e-6 relative."""
import pytest
import torch

from _util import golden

pytestmark = pytest.mark.gpu

LEVELS = (((12, 16), 8), ((6, 8), 16), ((3, 4), 32))
CASES = (("plain", 0.05, True, 1000, 0.6, 100), ("topk", 0.05, False, 20, 0.6, 10),
         ("sparse", 0.45, True, 1000, 0.5, 100))
SCORE_TOL = 1e-6


class Images:
    def __init__(self, sizes):
        self.image_sizes = sizes

    def __len__(self):
        return len(self.image_sizes)


def _outputs(locations, logits, reg, ext, ctr, strides, th, with_ctr, pre, nms_th, post, sizes):
    from dance_amd.fcose_outputs import FCOSEOutputs
    return FCOSEOutputs(Images(sizes), locations, logits, reg, ext, ctr, 0.25, 2.0, None, None, True, None,
                        strides, 1.5, logits[0].size(1), th, pre, nms_th, post, with_ctr)


def _score_close(a, b):
    return float(((a - b).abs() / b.abs().clamp_min(1e-3)).max()) <= SCORE_TOL if a.numel() else True


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_predict_proposals_golden(case):
    from dance_amd.fcose_outputs import compute_locations
    tag, th, with_ctr, pre, nms_th, post = case
    g = golden("g11_fcose")
    dev = torch.device("cuda:0")
    t = lambda k: g[k].to(dev)
    logits = [t("%s_logits%d" % (tag, l)) for l in range(3)]
    reg = [t("%s_reg%d" % (tag, l)) for l in range(3)]
    ext = [t("%s_ext%d" % (tag, l)) for l in range(3)]
    ctr = [t("%s_ctr%d" % (tag, l)) for l in range(3)]
    strides = [s for _, s in LEVELS]
    locations = compute_locations(logits, strides)
    res = _outputs(locations, logits, reg, ext, ctr, strides, th, with_ctr, pre, nms_th, post,
                   [(96, 128), (90, 120)]).predict_proposals()
    assert len(res) == 2
    for i, r in enumerate(res):
        want = {k: g["%s_im%d_%s" % (tag, i, k)] for k in ("boxes", "scores", "classes",
                                                                              "locations", "ext")}
        assert len(r) == want["scores"].numel()
        assert r.image_size == [(96, 128), (90, 120)][i]
        assert torch.equal(r.pred_classes.cpu(), want["classes"])
        assert torch.equal(r.pred_boxes.tensor.cpu(), want["boxes"])
        assert torch.equal(r.locations.cpu(), want["locations"])
        assert torch.equal(r.ext_points.tensor.cpu(), want["ext"])
        assert _score_close(r.scores.cpu(), want["scores"])


def _random_head(seed, n, c, shapes, thresh, with_ctr):
    """Head outputs whose scores keep a 1e-5 margin to the threshold (logits too close are pushed
    down), so that the candidate set does not hang on the last bit of exp()."""
    g = torch.Generator().manual_seed(seed)
    logits, reg, ext, ctr = [], [], [], []
    for h, w in shapes:
        lg = torch.randn(n, c, h, w, generator=g) * 1.5 - 2.5
        ct = torch.randn(n, 1, h, w, generator=g)
        s = torch.sigmoid(lg)
        v = s * torch.sigmoid(ct) if with_ctr else s
        lg[(v - thresh).abs() < 1e-5] = -12.0
        logits.append(lg)
        ctr.append(ct)
        reg.append(torch.rand(n, 4, h, w, generator=g) * 6 + 0.25)
        ext.append(torch.randn(n, 4, h, w, generator=g) * 0.3)
    return logits, reg, ext, ctr


@pytest.mark.parametrize("with_ctr,with_ext", [(True, True), (False, True), (True, False)])
def test_decode_candidates_vs_oracle(with_ctr, with_ext):
    """Every candidate of every level and image, before top-k and NMS, in the reference's
    `nonzero` order: five levels of a 200 x 304 image, 80 classes, ragged level sizes."""
    from dance_amd.fcose_outputs import decode_candidates
    from oracle import fcose_ref
    shapes, strides = [(25, 38), (13, 19), (7, 10), (4, 5), (2, 3)], [8, 16, 32, 64, 128]
    n, c, th = 3, 80, 0.05
    logits, reg, ext, ctr = _random_head(7, n, c, shapes, th, with_ctr)
    locations = fcose_ref.compute_locations(shapes, strides)
    want = fcose_ref.predict_proposals(locations, logits, reg, ext if with_ext else None, ctr, strides, th, with_ctr,
                                       10 ** 9, 0.0, 0, before_nms=True)
    dev = torch.device("cuda:0")
    up = lambda lst: [x.to(dev) for x in lst]
    fields, bounds = decode_candidates(up(locations), up(logits), up(reg), up(ext) if with_ext else None, up(ctr),
                                       strides, th, with_ctr)
    assert (fields["ext"] is None) == (not with_ext)
    for i in range(n):
        lo, hi = bounds[i][0], bounds[i][-1]
        assert hi - lo == want[i]["scores"].numel() and hi > lo
        assert torch.equal(fields["classes"][lo:hi].cpu(), want[i]["classes"])
        assert torch.equal(fields["locations"][lo:hi].cpu(), want[i]["locations"])
        assert torch.equal(fields["boxes"][lo:hi].cpu(), want[i]["boxes"])
        if with_ext:
            assert torch.equal(fields["ext"][lo:hi].cpu(), want[i]["ext"])
        assert _score_close(fields["scores"][lo:hi].cpu(), want[i]["scores"])
        assert _score_close(fields["raw"][lo:hi].cpu(), want[i]["raw"])
    assert bounds[-1][-1] == fields["scores"].numel()


def test_top_k_and_cut_vs_oracle():
    """Per-level top-k (unsorted in the reference) and the post-NMS cut: same detections as the
    restatement, compared in the NMS's score order."""
    from oracle import fcose_ref
    shapes, strides = [(20, 24), (10, 12), (5, 6)], [8, 16, 32]
    n, c, th, pre, nms_th, post = 2, 8, 0.05, 50, 0.6, 30
    logits, reg, ext, ctr = _random_head(11, n, c, shapes, th, True)
    locations = fcose_ref.compute_locations(shapes, strides)
    want = fcose_ref.predict_proposals(locations, logits, reg, ext, ctr, strides, th, True, pre, nms_th, post)
    dev = torch.device("cuda:0")
    up = lambda lst: [x.to(dev) for x in lst]
    res = _outputs(up(locations), up(logits), up(reg), up(ext), up(ctr), strides, th, True, pre, nms_th, post,
                   [(160, 192)] * n).predict_proposals()
    for r, w in zip(res, want):
        assert len(r) == w["scores"].numel() == post
        assert torch.equal(r.pred_classes.cpu(), w["classes"])
        assert torch.equal(r.pred_boxes.tensor.cpu(), w["boxes"])
        assert torch.equal(r.ext_points.tensor.cpu(), w["ext"])
        assert _score_close(r.scores.cpu(), w["scores"])


def test_no_candidates_and_single_level_api():
    from dance_amd.fcose_outputs import compute_locations
    dev = torch.device("cuda:0")
    logits, reg, ext, ctr = [[x.to(dev) for x in lst] for lst in _random_head(3, 2, 4, [(6, 8), (3, 4)], 0.05, True)]
    strides = [8, 16]
    locations = compute_locations(logits, strides)
    o = _outputs(locations, logits, reg, ext, ctr, strides, 0.9999, True, 1000, 0.6, 100, [(48, 64)] * 2)
    res = o.predict_proposals()
    assert [len(r) for r in res] == [0, 0]
    assert res[0].pred_boxes.tensor.shape == (0, 4) and res[0].ext_points.tensor.shape == (0, 4, 2)
    # one level through the reference's per-level method, without extreme points
    o = _outputs(locations, logits, reg, None, ctr, strides, 0.05, True, 1000, 0.6, 100, [(48, 64)] * 2)
    per_image = o.forward_for_single_feature_map(locations[0], logits[0], reg[0] * 8, None, ctr[0], o.image_sizes)
    from oracle import fcose_ref
    want = fcose_ref.level_candidates(locations[0].cpu(), logits[0].cpu(), reg[0].cpu() * 8, None, ctr[0].cpu(), 0.05,
                                      True, 1000)
    for r, w in zip(per_image, want):
        assert not r.has("ext_points")
        assert torch.equal(r.pred_boxes.tensor.cpu(), w["boxes"]) and torch.equal(r.pred_classes.cpu(), w["classes"])
    with pytest.raises(RuntimeError):
        _outputs([x.cpu() for x in locations], [x.cpu() for x in logits], [x.cpu() for x in reg], None,
                 [x.cpu() for x in ctr], strides, 0.05, True, 1000, 0.6, 100, [(48, 64)] * 2).predict_proposals()
    with pytest.raises(NotImplementedError):
        o.losses()


def test_full_size_counts_and_boxes():
    """BASELINE-size head outputs (16 images of 800 x 1216, five levels, 80 classes): the number of
    candidates per image equals torch's own count of the thresholded scores, and every decoded
    box equals the reference's formula evaluated with torch ops at the candidates' locations."""
    from dance_amd.fcose_outputs import compute_locations, decode_candidates
    dev = torch.device("cuda:0")
    shapes, strides = [(100, 152), (50, 76), (25, 38), (13, 19), (7, 10)], [8, 16, 32, 64, 128]
    n, c, th = 16, 80, 0.05
    g = torch.Generator(device=dev).manual_seed(5)
    logits = [torch.randn(n, c, h, w, generator=g, device=dev) * 1.5 - 5.0 for h, w in shapes]
    ctr = [torch.randn(n, 1, h, w, generator=g, device=dev) for h, w in shapes]
    reg = [torch.rand(n, 4, h, w, generator=g, device=dev) * 6 + 0.25 for h, w in shapes]
    ext = [torch.randn(n, 4, h, w, generator=g, device=dev) * 0.3 for h, w in shapes]
    locations = compute_locations(logits, strides)
    fields, bounds = decode_candidates(locations, logits, reg, ext, ctr, strides, th, True)
    total = 0
    for l in range(5):
        v = torch.sigmoid(logits[l]) * torch.sigmoid(ctr[l])
        near = int(((v - th).abs() < 1e-6).sum())
        per_image = (v > th).flatten(1).sum(1).tolist()
        for i in range(n):
            assert abs((bounds[i][l + 1] - bounds[i][l]) - per_image[i]) <= near
        total += sum(per_image)
    assert abs(fields["scores"].numel() - total) <= 5 and total > 1000
    # box arithmetic at the candidates: loc -+ reg * stride, level by level for image 3
    for l in range(5):
        lo, hi = bounds[3][l], bounds[3][l + 1]
        loc = fields["locations"][lo:hi]
        w = shapes[l][1]
        cell = ((loc[:, 1] - strides[l] // 2) / strides[l]).long() * w + ((loc[:, 0] - strides[l] // 2) / strides[l]).long()
        r = (reg[l][3] * strides[l]).flatten(1)[:, cell].t()
        box = torch.stack([loc[:, 0] - r[:, 0], loc[:, 1] - r[:, 1], loc[:, 0] + r[:, 2], loc[:, 1] + r[:, 3]], 1)
        assert torch.equal(fields["boxes"][lo:hi], box)
        assert bool((fields["scores"][lo:hi] ** 2 > th * (1 - 1e-5)).all())


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_single_image_path_golden(case):
    """One image per call (the reference's inference mode) takes the per-image NMS branch: the
    fixture's images one at a time give the fixture's results."""
    from dance_amd.fcose_outputs import compute_locations
    tag, th, with_ctr, pre, nms_th, post = case
    g = golden("g11_fcose")
    dev = torch.device("cuda:0")
    strides = [s for _, s in LEVELS]
    sizes = [(96, 128), (90, 120)]
    for i in range(2):
        t = lambda k: g[k][i:i + 1].to(dev)
        logits = [t("%s_logits%d" % (tag, l)) for l in range(3)]
        res = _outputs(compute_locations(logits, strides), logits, [t("%s_reg%d" % (tag, l)) for l in range(3)],
                       [t("%s_ext%d" % (tag, l)) for l in range(3)], [t("%s_ctr%d" % (tag, l)) for l in range(3)],
                       strides, th, with_ctr, pre, nms_th, post, sizes[i:i + 1]).predict_proposals()
        assert len(res) == 1
        r = res[0]
        assert torch.equal(r.pred_classes.cpu(), g["%s_im%d_classes" % (tag, i)])
        assert torch.equal(r.pred_boxes.tensor.cpu(), g["%s_im%d_boxes" % (tag, i)])
        assert torch.equal(r.ext_points.tensor.cpu(), g["%s_im%d_ext" % (tag, i)])
        assert _score_close(r.scores.cpu(), g["%s_im%d_scores" % (tag, i)])
