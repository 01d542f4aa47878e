"""GPU: dance_amd.targets (one workgroup per instance) against the reference's own outputs
(tests/golden/g12_targets.npz) and against oracle/targets_ref.py on random polygons -- indices
and float64 coordinates bit for bit (the NumPy half of
RefineNet.compute_targets_for_polys, snake_head.py:982-1254)."""
import numpy as np
import pytest
import torch

from _util import golden
from dance_amd import targets
from oracle import snake_ref, targets_ref

pytestmark = pytest.mark.gpu


def _g():
    return {k: t.numpy() for k, t in golden("g12_targets").items()}


def test_uniform_sample_golden(device):
    g = _g()
    n = int(g["n_polys"])
    polys, newp, want = [], [], []
    for i in range(n):
        for tag in ("rm", "eq", "add"):
            polys.append(g["poly%d" % i])
            newp.append(int(g["us_%s_n%d" % (tag, i)]))
            want.append(g["us_%s_%d" % (tag, i)])
    got = targets.uniform_sample(polys, newp, device)
    for a, b in zip(got, want):
        assert a.cpu().numpy().dtype == b.dtype and np.array_equal(a.cpu().numpy(), b)


def test_dense_targets_and_matching_golden(device):
    g = _g()
    n, v, up = int(g["n_polys"]), int(g["v"]), int(g["up"])
    polys = [g["poly%d" % i] for i in range(n)]
    sampled = np.stack([g["sampled%d" % i] for i in range(n)])
    eidx = np.stack([g["edge_idx%d" % i] for i in range(n)])
    dense = targets.dense_targets(polys, sampled[:, 0], v * up, device)
    want_dense = np.stack([g["dense%d" % i] for i in range(n)])
    assert np.array_equal(dense.cpu().numpy(), want_dense)
    re, ext = targets.get_aux_extreme_points(dense)
    for i in range(n):
        assert np.array_equal(re[i].cpu().numpy(), g["aux_pts%d" % i])
        assert np.array_equal(ext[i].cpu().numpy(), g["aux_idx%d" % i])
    out, status = targets.single_uniform_segment_matching(dense, torch.from_numpy(sampled), torch.from_numpy(eidx), up)
    assert status.tolist() == [0] * n
    for i in range(n):
        assert np.array_equal(out[i].cpu().numpy(), g["match%d" % i])
    # the shipped configuration's form: box contour rolled to the first target + multisegment matching
    box = np.stack([g["box_pts%d" % i] for i in range(n)])
    rolled, mout, mstatus = targets.single_uniform_multisegment_matching(re, torch.from_numpy(box), re.ext_idx_padded,
                                                                         re.ext_count, up)
    assert mstatus.tolist() == [0] * n
    for i in range(n):
        assert np.array_equal(rolled[i].cpu().numpy(), g["box_rolled%d" % i])
        assert np.array_equal(mout[i].cpu().numpy(), g["mmatch%d" % i])
    # the reference's fall-back (collapsed ground truth)
    flat = torch.from_numpy(np.tile(np.array([[10.0, 20.0]]), (1, v * up, 1))).to(device)
    out, status = targets.single_uniform_segment_matching(flat, torch.from_numpy(sampled[-1:]),
                                                          torch.from_numpy(eidx[-1:]), up)
    assert status.tolist() == [1] and np.array_equal(out[0].cpu().numpy(), g["match_flat"])


def test_uniform_sample_1d_golden(device):
    g = _g()
    lines, ms, want = [], [], []
    for i in range(int(g["n_polys"])):
        for k in range(6):
            a, b, m = g["s1d_arg_%d_%d" % (i, k)]
            lines.append(g["dense%d" % i][a:b])
            ms.append(int(m))
            want.append(g["s1d_%d_%d" % (i, k)])
    got, status = targets.uniform_sample_1d(lines, ms, device)
    assert int(status.abs().sum()) == 0
    for a, b in zip(got, want):
        assert np.array_equal(a.cpu().numpy(), b)
    # a poly-line of zero length: the reference raises, status 1
    _, status = targets.uniform_sample_1d([np.zeros((5, 2))], [3], device)
    assert status.tolist() == [1]


def _star(seed, p, cx, cy, r0):
    g = np.random.default_rng(seed)
    ang = np.sort(g.random(p) * 2 * np.pi)
    rad = r0 * (0.5 + 0.5 * g.random(p))
    return np.stack([cx + rad * np.cos(ang), cy + 0.7 * rad * np.sin(ang)], 1).astype(np.float32)


@pytest.mark.parametrize("v", [32, 128])
def test_whole_chain_vs_oracle_random(device, v):
    """Random ground-truth polygons through the whole chain at the reference's sizes (128 points,
    up-rate 5): dense targets -> auxiliary extreme points -> segment matching, every stage bit
    for bit against the CPU restatement."""
    up, n = 5, 24
    rng = np.random.default_rng(v)
    polys = [_star(100 + i, int(rng.integers(4, 300)), 200 + 30 * i, 300 + 10 * i, float(rng.integers(15, 250)))
             for i in range(n)]
    ex = np.stack([np.stack([c[np.argmin(c[:, 1])], c[np.argmin(c[:, 0])], c[np.argmax(c[:, 1])], c[np.argmax(c[:, 0])]])
                   for c in polys])
    octs, eidx = snake_ref.get_octagon(torch.from_numpy(ex), v)
    sampled = octs.numpy()
    dense = targets.dense_targets(polys, sampled[:, 0], v * up, device)
    re, ext = targets.get_aux_extreme_points(dense)
    out, status = targets.single_uniform_segment_matching(dense, octs, eidx, up)
    for i, c in enumerate(polys):
        big = targets_ref.uniform_sample(c, len(c) * v * up)
        tt = int(np.argmin(np.power(big - sampled[i, 0], 2).sum(axis=1)))
        want = np.roll(big, -tt, axis=0)[::len(c)]
        assert np.array_equal(dense[i].cpu().numpy(), want), i
        re_w, ext_w = targets_ref.get_aux_extreme_points(want)
        assert np.array_equal(re[i].cpu().numpy(), re_w) and np.array_equal(ext[i].cpu().numpy(), ext_w), i
        m_w, st_w = targets_ref.single_uniform_segment_matching(want, sampled[i], eidx[i].numpy(), up, v)
        assert int(status[i]) == st_w and np.array_equal(out[i].cpu().numpy(), m_w), i
    # box contours through the multisegment form (INITIAL = box)
    boxes = torch.tensor([[float(c[:, 0].min()), float(c[:, 1].min()), float(c[:, 0].max()), float(c[:, 1].max())]
                          for c in polys])
    bpts = snake_ref.uniform_upsample(snake_ref.box_polygon(boxes)[None], v)[0]
    rolled, mout, mstatus = targets.single_uniform_multisegment_matching(re, bpts, re.ext_idx_padded, re.ext_count, up)
    for i in range(n):
        re_w, ext_w = re[i].cpu().numpy(), ext[i].cpu().numpy()
        r_w = targets_ref.roll_to_first_target(bpts[i].numpy(), re_w)
        assert np.array_equal(rolled[i].cpu().numpy(), r_w), i
        m_w, st_w = targets_ref.single_uniform_multisegment_matching(re_w, r_w, ext_w, up, v)
        assert int(mstatus[i]) == st_w and np.array_equal(mout[i].cpu().numpy(), m_w), i


def test_uniform_sample_random_vs_oracle(device):
    rng = np.random.default_rng(5)
    polys, newp = [], []
    for i in range(40):
        p = int(rng.integers(3, 400))
        polys.append(_star(i, p, 100.0, 100.0, 60.0))
        newp.append(int(rng.integers(max(1, p - 20), 6 * p)))
    got = targets.uniform_sample(polys, newp, device)
    for c, k, a in zip(polys, newp, got):
        w = targets_ref.uniform_sample(c, k)
        assert a.cpu().numpy().dtype == w.dtype and np.array_equal(a.cpu().numpy(), w)


def test_compute_targets_for_polys_golden(device):
    """dance_amd.targets.compute_targets_for_polys against the reference's own driver (g13): three
    images, a reversed ring, a fragmented and a zero-area instance, an image without ground truth."""
    from _util import targets_driver_fixture
    polys, boxes, sizes, v, want = targets_driver_fixture()
    got = targets.compute_targets_for_polys(polys, boxes, sizes, v, device=device)
    assert sorted(got) == sorted(want)
    for k in want:
        a = got[k].cpu()
        assert a.dtype == want[k].dtype and a.shape == want[k].shape, k
        assert torch.equal(a, want[k]), k


def test_compute_targets_for_polys_vs_oracle_random(device):
    """200 random star polygons over 8 images at the shipped 128 x 5 sampling, some given
    counter-clockwise, against the CPU oracle; plus the error paths."""
    rng = np.random.default_rng(3)
    polys, boxes, sizes = [], [], []
    for i in range(8):
        im, bx = [], []
        for j in range(25):
            c = _star(100 * i + j, int(rng.integers(6, 120)), 300 + 20 * j, 250 + 10 * i, int(rng.integers(15, 140)))
            if j % 4 == 0:
                c = c[::-1].copy()
            im.append([c.reshape(-1)])
            bx.append([float(c[:, 0].min()), float(c[:, 1].min()), float(c[:, 0].max()), float(c[:, 1].max())])
        polys.append(im)
        boxes.append(torch.tensor(bx, dtype=torch.float32))
        sizes.append((640, 900))
    v = 128
    got = targets.compute_targets_for_polys(polys, boxes, sizes, v, device=device)
    want = targets_ref.compute_targets_for_polys_box(polys, boxes, sizes, v)
    for k in want:
        assert torch.equal(got[k].cpu(), want[k]), k
    assert got["sample_locs"].shape == (200, v, 2)
    assert float(got["sample_targets"][..., 0].max()) <= 899.0 and float(got["sample_targets"][..., 1].max()) <= 639.0
    with pytest.raises(RuntimeError):                   # nothing left: the reference's torch.stack([]) fails too
        targets.compute_targets_for_polys([[]], [torch.zeros((0, 4))], [(10, 10)], v, device=device)
    with pytest.raises(RuntimeError):
        targets.compute_targets_for_polys([polys[0]], [boxes[0][:3]], [sizes[0]], v, device=device)
