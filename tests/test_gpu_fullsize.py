"""GPU: BASELINE.json configs 3, 4 and 5 at their FULL sizes, checked through properties that
do not need a full-size CPU run: a random sample of contours is recomputed by the oracle on
just its own image (every contour depends only on its image's features -- SURVEY 8e), results
are independent of the chunking over contours, and permuting the contours permutes the
output.  Per-iteration tolerance 1e-4 norm-wise (iteration 0 from identical inputs; later
iterations are compared on the oracle's own chain for the sampled contours, on band-limited
features where the chain is well conditioned)."""
import pytest
import torch

from _util import (REL_TOL, FakeBoxes, FakeInstances, assert_coords_close, coord_rel_err, dance_state_dict,
                   dsnake_state_dict)
from dance_amd import modules, ops, snake, synthetic
from oracle import snake_ref

pytestmark = pytest.mark.gpu
HF, WF = 200, 304


def _features(b, c, seed, device, attention=True):
    """(b, [1+]c, HF, WF) band-limited features, generated image by image on the host."""
    out = torch.empty((b, c + (1 if attention else 0), HF, WF), dtype=torch.float32, device=device)
    # 8 distinct maps, cycled: bounds host time, every image still differs from its neighbours
    maps = [synthetic.make_smooth_features(1, c, HF, WF, seed + k, attention=attention)[0].to(device)
            for k in range(min(b, 8))]
    for i in range(b):
        out[i] = maps[i % 8]
    return out


def _dance_inputs(b, per, v, seed, device):
    n = b * per
    boxes = synthetic.make_boxes(n, HF * 4, WF * 4, seed)
    whs = torch.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1)
    init, _ = ops.uniform_upsample(boxes.to(device), v, kind=ops.POLY_BOX)
    idx = torch.arange(b).repeat_interleave(per).float()
    return boxes, whs, init, idx


def _check_sample(d, preds, sd, feats, init, idx, whs, sample):
    """oracle on the sampled contours, each with its own image only: the free-running chain
    (1e-4 norm-wise; band-limited features) and every iteration on identical inputs -- the
    oracle's previous contours fed to the HIP iteration -- to 1e-4 norm-wise and 0.02 px"""
    for j in sample:
        img = int(idx[j])
        f1 = feats[img:img + 1].cpu()
        want = snake_ref.dance_deform(sd, f1, init[j:j + 1].cpu(), torch.zeros(1), [(HF * 4, WF * 4)],
                                      whs[j:j + 1].cpu())
        f = d.prepare_features(feats[img:img + 1])
        prev = init[j:j + 1]
        for it in range(3):
            assert coord_rel_err(preds[it][j:j + 1], want[it]) < REL_TOL, (j, it)
            one = d.evolve(it, *f, prev.to(feats.device), torch.zeros(1, dtype=torch.int32, device=feats.device),
                           whs[j:j + 1].to(feats.device), use_att=it > 0, clip_first=False, clip_max=(0, 0))
            assert_coords_close(one, want[it], "contour %d iteration %d" % (j, it))
            prev = want[it]


def test_config3_dance_32x100_attention(device):
    """BASELINE config 3: batch 32 images x 100 instances, attentive deformation, all iterations."""
    b, per, v = 32, 100, 128
    cfg, sd, _ = dance_state_dict(1234, v)
    nets = [snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer%d." % i), device) for i in range(3)]
    feats = _features(b, 256, 500, device)
    boxes, whs, init, idx = _dance_inputs(b, per, v, 77, device)
    d = snake.ContourDeformer(nets)
    preds = d.deform(feats, init, idx.to(device), [(HF * 4, WF * 4)] * b, whs.to(device))
    assert all(p.shape == (b * per, v, 2) and bool(torch.isfinite(p).all()) for p in preds)
    # chunk invariance (bit-exact) and permutation equivariance (bit-exact)
    d2 = snake.ContourDeformer(nets)
    d2.MAX_CHUNK = 1000
    again = d2.deform(feats, init, idx.to(device), [(HF * 4, WF * 4)] * b, whs.to(device))
    perm = torch.randperm(b * per, generator=torch.Generator().manual_seed(3)).to(device)
    shuffled = d.deform(feats, init[perm], idx.to(device)[perm], [(HF * 4, WF * 4)] * b, whs.to(device)[perm])
    for it in range(3):
        assert torch.equal(preds[it], again[it])
        assert torch.equal(preds[it][perm], shuffled[it])
    # two HIP streams (the RefineNet module's default from 1024 contours on): kernels of the two halves
    # share CUs; three runs, every iteration bit-identical to the one-stream result
    for _ in range(3):
        two = d.deform(feats, init, idx.to(device), [(HF * 4, WF * 4)] * b, whs.to(device), streams=2)
        for it in range(3):
            assert torch.equal(preds[it], two[it])
    _check_sample(d, preds, sd, feats, init, idx, whs, [0, 1234, 3199])


def test_config4_dsnake_32x100(device):
    """BASELINE config 4: Dsnake R50, batch 32 images x 100 instances: init (V=40) ->
    extreme points -> octagon -> 3 evolves, through the drop-in module."""
    b, per = 32, 100
    cfg, sd, _ = dsnake_state_dict(1234)
    head = modules.DsnakeHead(cfg)
    head.load_state_dict(sd)
    head = head.to(device).eval()
    feats = _features(b, 256, 900, device, attention=False)
    boxes = synthetic.make_boxes(b * per, HF * 4, WF * 4, 78).view(b, per, 4)
    insts = [FakeInstances((HF * 4, WF * 4), pred_boxes=FakeBoxes(boxes[i].to(device))) for i in range(b)]
    with torch.no_grad():
        res, _ = head(feats, insts, None)
    got = torch.stack([r.pred_polys.tensor for r in res])             # (b, per, 128, 2)
    assert got.shape == (b, per, 128, 2) and bool(torch.isfinite(got).all())
    # independence: two images alone give the same contours as inside the batch of 32
    for i in (0, 31):
        alone = [FakeInstances((HF * 4, WF * 4), pred_boxes=FakeBoxes(boxes[i].to(device)))]
        with torch.no_grad():
            r1, _ = head(feats[i:i + 1], alone, None)
        assert torch.equal(r1[0].pred_polys.tensor, got[i])
    # oracle on a sample of instances of one image (its own feature map only)
    sample = [0, 50, 99]
    want = snake_ref.dsnake_deform(sd, feats[5:6].cpu(), [boxes[5][sample]])
    assert coord_rel_err(got[5][sample], want["preds"][-1]) < REL_TOL


def test_config5_large_sweep_256v(device):
    """BASELINE config 5, one GPU's view of the sweep at FULL size: 256 images x 300 instances
    x 256 vertices (76,800 contours; 16 GB of features), chunked over contours."""
    b, per, v = 256, 300, 256
    cfg, sd, _ = dance_state_dict(1234, v)
    nets = [snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer%d." % i), device) for i in range(3)]
    feats = _features(b, 256, 1300, device)
    boxes, whs, init, idx = _dance_inputs(b, per, v, 79, device)
    d = snake.ContourDeformer(nets)
    f = d.prepare_features(feats)
    preds = d.deform(f, init, idx.to(device), [(HF * 4, WF * 4)] * b, whs.to(device), layout="nhwc")
    assert all(p.shape == (b * per, v, 2) and bool(torch.isfinite(p).all()) for p in preds)
    # the first and the last image's contours, recomputed alone, are bit-identical
    for img in (0, b - 1):
        sl = slice(img * per, (img + 1) * per)
        alone = d.deform(feats[img:img + 1], init[sl], torch.zeros(per, device=device), [(HF * 4, WF * 4)],
                         whs[sl].to(device))
        for it in range(3):
            assert torch.equal(alone[it], preds[it][sl]), (img, it)
    _check_sample(d, preds, sd, feats, init, idx, whs, [7, 40000, 76799])


def test_dsnake_two_streams_equal_one_stream(device, monkeypatch):
    """Dsnake at 32 x 100 with the batch's halves on two HIP streams (DANCE_STREAMS=2) against one
    stream, three times over: every stage bit-identical.  With two streams
    kernels of different kinds share CUs -- the resampler beside the snake's MFMA kernels is where a
    packed-fp32 lerp once came out wrong in lanes 48..63 (dance_amd/build.py, NOPK)."""
    b, per = 32, 100
    cfg, sd, _ = dsnake_state_dict(4321)
    head = modules.DsnakeHead(cfg)
    head.load_state_dict(sd)
    head = head.to(device).eval()
    feats = _features(b, 256, 901, device, attention=False)
    boxes = synthetic.make_boxes(b * per, HF * 4, WF * 4, 80).view(b, per, 4)

    def run(streams):
        monkeypatch.setenv("DANCE_STREAMS", str(streams))
        insts = [FakeInstances((HF * 4, WF * 4), pred_boxes=FakeBoxes(boxes[i].to(device))) for i in range(b)]
        with torch.no_grad():
            head(feats, insts, None)
        st = head.last_stages
        return [st["exts_raw"].clone(), st["exts"].clone(), st["octagons"].clone()] + [p.clone() for p in st["preds"]]

    ref = run(1)
    for _ in range(3):
        for a, r in zip(run(2), ref):
            assert torch.equal(a, r)


def test_peeled_remainder_equals_one_stream(device, monkeypatch):
    """BASELINE config 2's count (16 x 100 = 1600 contours = 6.25 x 256 CUs): deform() hands n mod CUs
    contours to a side HIP stream (ContourDeformer.peel_count) so that the main stream's kernels run
    whole rounds of workgroups; three runs, every iteration bit-identical to the one-stream result
    (DANCE_PEEL=0), and stage events still arrive on the main stream."""
    monkeypatch.delenv("DANCE_PEEL", raising=False)
    b, per, v = 16, 100, 128
    cfg, sd, _ = dance_state_dict(1235, v)
    nets = [snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer%d." % i), device) for i in range(3)]
    feats = _features(b, 256, 501, device)
    boxes, whs, init, idx = _dance_inputs(b, per, v, 81, device)
    d = snake.ContourDeformer(nets)
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    with torch.cuda.device(device):
        side_n = d.peel_count(b * per, torch.device(device))
    assert side_n == (b * per) % cus
    if side_n == 0:
        pytest.skip("1600 is a multiple of this GPU's CU count")
    f = d.prepare_features(feats)
    runs = [d.deform(f, init, idx.to(device), [(HF * 4, WF * 4)] * b, whs.to(device), layout="nhwc") for _ in range(3)]
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(snake.EVOLVE_EVENTS)] for _ in range(3)]
    for it_ev in ev:
        for e in it_ev:
            e.record()                                          # creates the handle the library records into
    timed = d.deform(f, init, idx.to(device), [(HF * 4, WF * 4)] * b, whs.to(device), layout="nhwc", stage_events=ev)
    torch.cuda.synchronize()
    assert ev[0][0].elapsed_time(ev[2][-1]) > 0
    monkeypatch.setenv("DANCE_PEEL", "0")
    with torch.cuda.device(device):
        assert d.peel_count(b * per, torch.device(device)) == 0
    one = d.deform(f, init, idx.to(device), [(HF * 4, WF * 4)] * b, whs.to(device), layout="nhwc")
    for it in range(3):
        assert bool(torch.isfinite(one[it]).all())
        for r in runs + [timed]:
            assert torch.equal(r[it], one[it]), it
    # a short batch, a multiple of the CU count and a chunked batch are not peeled
    monkeypatch.delenv("DANCE_PEEL")
    with torch.cuda.device(device):
        assert d.peel_count(2 * cus - 1, torch.device(device)) == 0
        assert d.peel_count(6 * cus, torch.device(device)) == 0
        assert d.peel_count(d.MAX_CHUNK + 1, torch.device(device)) == 0


def test_config1_single_image_100x128_every_contour(device):
    """BASELINE config 1 at FULL size -- the reference's own CPU-runnable case and its published
    inference mode (one image, Dance.single_test dance.py:52): (1, 257, 200, 304) features, 100 x 128,
    image_idx None, so clip_locations' quirk is live (only contour 0 is clamped, snake_head.py:1718).
    ALL 100 contours against the oracle: the free-running chain (band-limited features) and every
    iteration on identical inputs, 1e-4 norm-wise and 0.02 px."""
    n, v = 100, 128
    cfg, sd, _ = dance_state_dict(1236, v)
    nets = [snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer%d." % i), device) for i in range(3)]
    feats = _features(1, 256, 1500, device)
    boxes = synthetic.make_boxes(n, HF * 4, WF * 4, 82)
    boxes[0] = torch.tensor([-30.0, -20.0, 90.0, 70.0])       # contour 0 starts outside: the clamp acts
    boxes[1] = torch.tensor([WF * 4 - 60.0, HF * 4 - 50.0, WF * 4 + 25.0, HF * 4 + 30.0])   # contour 1: it must not
    whs = torch.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1)
    init_ref, _ = snake_ref.uniform_upsample(snake_ref.box_polygon(boxes)[None], v)
    init, _ = ops.uniform_upsample(boxes.to(device), v, kind=ops.POLY_BOX)
    assert torch.equal(init.cpu(), init_ref)
    want = snake_ref.dance_deform(sd, feats.cpu(), init_ref, None, (HF * 4, WF * 4), whs)
    d = snake.ContourDeformer(nets)
    got = d.deform(feats, init, None, (HF * 4, WF * 4), whs.to(device))
    f = d.prepare_features(feats)
    prev = init
    for it in range(3):
        assert got[it].shape == (n, v, 2)
        assert coord_rel_err(got[it], want[it]) < REL_TOL, ("chain", it)
        one = d.evolve(it, *f, prev, None, whs.to(device), use_att=it > 0, clip_first=True,
                       clip_max=(WF * 4 - 1, HF * 4 - 1))
        assert_coords_close(one, want[it], "config 1, iteration %d, all contours" % it)
        prev = want[it].to(device)
    # the quirk at full size, after iteration 0: contour 0 clamped into the image (and onto its border),
    # contour 1 free to leave it.  (Contour 0 then collapses onto y = 0 with these weights, so the
    # reference's eps-free derange makes it NaN from iteration 1 on -- here too: coord_rel_err above
    # demands the NaNs at identical places.)
    first, w0 = got[0].cpu(), want[0]
    assert float(first[0].min()) >= 0.0 and float(first[0, :, 0].max()) <= WF * 4 - 1 and bool((first[0] == 0).any())
    out1 = (w0[1, :, 0] > WF * 4 - 1) | (w0[1, :, 1] > HF * 4 - 1)
    assert bool(out1.any()) and torch.equal((first[1, :, 0] > WF * 4 - 1) | (first[1, :, 1] > HF * 4 - 1), out1)
    assert bool(torch.isnan(want[2][0]).all()) and bool(torch.isfinite(want[2][1:]).all())
