"""Dynamic range of the default fast precision mode (fp16x3: fp16 hi/lo pairs, csrc/p16.h).

A pair carries 22 bits only while 2^-3 <= |x| < 65504, so everything that is split is first
multiplied by a power of two -- measured for the features (per image) and the de_location
values (per contour), estimated from the BatchNorm parameters for the state slots when a net is
packed (dance_amd/snake.py SnakeNetWeights._pack16).  These tests move the network's internal
scales by exact powers of two, far outside what the He-scaled synthetic fixtures exercise, and
hold every iteration on identical inputs to the same gate as everywhere else (1e-4 norm-wise
AND 0.02 px against the fp32 oracle run on the SAME rescaled weights and features).  Where a
rescaling is an exact invariance of the fp32 network the oracle's own result is also checked
to be unchanged.  Beyond the representable range the result must be NaN (loud), never a finite
wrong number.  reference arithmetic: fp32 throughout, snake_head.py:92-153, 220-284."""
import os

import pytest
import torch

from _util import ABS_TOL_PX, assert_coords_close, dance_state_dict, dsnake_state_dict
from dance_amd import modules, ops, snake, synthetic
from oracle import snake_ref

pytestmark = pytest.mark.gpu
HF, WF, N, V = 40, 56, 24, 128
BLOCKS = ["head"] + ["res%d" % i for i in range(7)]


def _case(seed=7):
    feats = synthetic.make_smooth_features(2, 256, HF, WF, seed)
    boxes = synthetic.make_boxes(N, HF * 4, WF * 4, seed + 1, r_min=8, r_max=60)
    whs = torch.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1)
    init, _ = snake_ref.uniform_upsample(snake_ref.box_polygon(boxes)[None], V)
    idx = (torch.arange(N) % 2).float()
    return feats, whs, init, idx


def _oracle(sd, feats, whs, init, idx):
    return snake_ref.dance_deform(sd, feats, init.clone(), idx, [(HF * 4, WF * 4)] * 2, whs)


def _fast_iterations(device, sd, feats, whs, idx, inputs):
    """every iteration of the fast mode on the given input contours -> list of (N, V, 2) on the host"""
    nets = [snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer%d." % i), device) for i in range(3)]
    d = snake.ContourDeformer(nets, precision="fp16x3")
    f = d.prepare_features(feats.to(device))
    out = []
    for it in range(3):
        one = d.evolve(it, *f, inputs[it].to(device).contiguous(), idx.to(torch.int32).to(device), whs.to(device),
                       use_att=it > 0, clip_first=False, clip_max=(0.0, 0.0))
        out.append(one.cpu())
    return out, nets


def _gate(device, sd, feats, whs, init, idx, what, unchanged=None):
    want = _oracle(sd, feats, whs, init, idx)
    assert all(bool(torch.isfinite(w).all()) for w in want), what
    if unchanged is not None:               # an exact invariance of the fp32 network
        for a, b in zip(want, unchanged):
            assert float((a - b).abs().max()) <= 1e-3 * ABS_TOL_PX, what
    got, nets = _fast_iterations(device, sd, feats, whs, idx, [init] + want[:2])
    for it in range(3):
        assert_coords_close(got[it], want[it], "%s, iteration %d" % (what, it))
    return want, nets


def _scale_features(sd, feats, k):
    """features * 2^-k, the head block's feature columns * 2^k (attention channel untouched)"""
    sd = {key: t.clone() for key, t in sd.items()}
    feats = feats.clone()
    feats[:, 1:] *= 2.0 ** -k
    for i in range(3):
        sd["deformer%d.head.conv.fc.weight" % i][:, :256] *= 2.0 ** k
    return sd, feats


def _scale_stream(sd, k):
    """the whole residual stream * 2^-k: every block's BatchNorm gamma, beta * 2^-k, every consumer of
    a state slot (residual convolutions, fusion, prediction[0]'s state columns) * 2^k"""
    sd = {key: t.clone() for key, t in sd.items()}
    for i in range(3):
        p = "deformer%d." % i
        for b in BLOCKS:
            sd[p + b + ".norm.weight"] *= 2.0 ** -k
            sd[p + b + ".norm.bias"] *= 2.0 ** -k
        for j in range(7):
            sd[p + "res%d.conv.fc.weight" % j] *= 2.0 ** k
        sd[p + "fusion.weight"] *= 2.0 ** k
        sd[p + "prediction.0.weight"][:, 256:] *= 2.0 ** k
    return sd


@pytest.mark.parametrize("k", [7, 10, 13, -7, -13, -30])
def test_feature_scale(device, k):
    """features at 2^-k of the fixtures' O(1) scale (k < 0: larger), compensated in the head
    convolution: the per-image measured maximum puts them back into the pairs' range"""
    cfg, sd, _ = dance_state_dict(4321, V)
    feats, whs, init, idx = _case()
    base = _oracle(sd, feats, whs, init, idx)
    sd2, feats2 = _scale_features(sd, feats, k)
    _gate(device, sd2, feats2, whs, init, idx, "features * 2^%d" % -k, unchanged=base)


@pytest.mark.parametrize("k", [7, 10, 13, -7, -13, -30])
def test_residual_stream_scale(device, k):
    """all eight state slots at 2^-k of the fixtures' scale: the slot exponents follow the
    BatchNorm parameters exactly"""
    cfg, sd, _ = dance_state_dict(4322, V)
    feats, whs, init, idx = _case(11)
    base = _oracle(sd, feats, whs, init, idx)
    sd2 = _scale_stream(sd, k)
    want, nets = _gate(device, sd2, feats, whs, init, idx, "residual stream * 2^%d" % -k, unchanged=base)
    ref_nets = [snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer%d." % i), device) for i in range(3)]
    for a, b in zip(nets, ref_nets):        # the stored values are the same numbers as without the rescaling
        assert [x + k for x in b.act_exps["slots"]] == a.act_exps["slots"]


@pytest.mark.parametrize("k", [7, 13])
@pytest.mark.parametrize("block", [0, 3, 7])
def test_one_block_small(device, k, block):
    """ONE block's BatchNorm gamma, beta * 2^-k (block 0 = head), nothing compensating (the residual sum
    makes a per-block rescaling no invariance of the network, and compensating it in the next
    convolution would feed that one inputs 2^k above its statistics -- an ill-conditioned net, whatever
    the arithmetic): a layer whose contribution sits 2^-k below the stream it is added to; the oracle
    runs on the same modified net"""
    cfg, sd, _ = dance_state_dict(4323, V)
    feats, whs, init, idx = _case(13)
    sd = {key: t.clone() for key, t in sd.items()}
    for i in range(3):
        p = "deformer%d." % i
        sd[p + BLOCKS[block] + ".norm.weight"] *= 2.0 ** -k
        sd[p + BLOCKS[block] + ".norm.bias"] *= 2.0 ** -k
    _gate(device, sd, feats, whs, init, idx, "block %d BatchNorm * 2^%d" % (block, -k))


@pytest.mark.parametrize("k", [10, -10])
def test_prediction_scale(device, k):
    """prediction[0]'s output * 2^-k (its weights and bias; ReLU is positively homogeneous), undone in
    prediction[2]: the stored block's exponent follows the weight norms"""
    cfg, sd, _ = dance_state_dict(4328, V)
    feats, whs, init, idx = _case(37)
    base = _oracle(sd, feats, whs, init, idx)
    sd = {key: t.clone() for key, t in sd.items()}
    for i in range(3):
        p = "deformer%d." % i
        sd[p + "prediction.0.weight"] *= 2.0 ** -k
        sd[p + "prediction.0.bias"] *= 2.0 ** -k
        sd[p + "prediction.2.weight"] *= 2.0 ** k
    _gate(device, sd, feats, whs, init, idx, "prediction[0] * 2^%d" % -k, unchanged=base)


def test_everything_small(device):
    """features at 2^-13 AND the stream at 2^-10 together"""
    cfg, sd, _ = dance_state_dict(4324, V)
    feats, whs, init, idx = _case(17)
    base = _oracle(sd, feats, whs, init, idx)
    sd2, feats2 = _scale_features(_scale_stream(sd, 10), feats, 13)
    _gate(device, sd2, feats2, whs, init, idx, "features * 2^-13, stream * 2^-10", unchanged=base)


def _evolve_ref(sd, feats, inputs, idx, whs, dtype):
    """the oracle's iterations on the given input contours, in fp32 or fp64"""
    out = []
    for it in range(3):
        sub = {k: (t.to(dtype) if t.is_floating_point() else t)
               for k, t in snake_ref.sub_state_dict(sd, "deformer%d." % it).items()}
        out.append(snake_ref.dance_evolve(sub, feats.to(dtype), inputs[it].to(dtype).clone(), idx,
                                          [(HF * 4, WF * 4)] * 2, whs.to(dtype), att=it > 0))
    return out


def test_overflow_edge_is_loud(device):
    """Activations that outgrow what the BatchNorm parameters promise: the head convolution (weights
    and bias) * 2^j with nothing compensating, so the whole residual stream sits ~2^j above its
    estimate.  Up to the pairs' headroom every contour is as close to an fp64 evaluation as the fp32
    oracle itself up to the pairs' 22 bits against fp32's 24 (such a net amplifies rounding noise, the
    oracle's included: the gate is the parity gate plus 16 times the oracle's own distance from fp64 --
    4 for the two bits, 4 for the spread of a maximum over a few thousand coordinates); beyond the
    headroom a contour is either still that close or NaN -- never finite and wrong."""
    cfg, sd, _ = dance_state_dict(4325, V)
    feats, whs, init, idx = _case(19)
    first_nan = None
    for j in (0, 4, 8, 10, 12, 14, 16, 20, 24):
        sd2 = {key: t.clone() for key, t in sd.items()}
        for i in range(3):
            sd2["deformer%d.head.conv.fc.weight" % i] *= 2.0 ** j
            sd2["deformer%d.head.conv.fc.bias" % i] *= 2.0 ** j
        want = _oracle(sd2, feats, whs, init, idx)
        assert all(bool(torch.isfinite(w).all()) for w in want)
        inputs = [init] + want[:2]
        ref64 = _evolve_ref(sd2, feats, inputs, idx, whs, torch.float64)
        got, _ = _fast_iterations(device, sd2, feats, whs, idx, inputs)
        for it in range(3):
            noise = float((want[it].double() - ref64[it]).abs().max())
            bad = torch.isnan(got[it]).flatten(1).any(1)
            if bool(bad.any()) and first_nan is None:
                first_nan = j
            assert bool(torch.isfinite(got[it][~bad]).all())
            if bool((~bad).any()):
                err = float((got[it][~bad].double() - ref64[it][~bad]).abs().max())
                assert err < ABS_TOL_PX + 16 * noise, (j, it, err, noise)
    assert first_nan is not None, "the sweep never reached the overflow edge"
    assert first_nan >= 8, "2^8 above the estimate must still be inside the headroom (snake.ACT_TARGET_EXP)"


def test_feature_overflow_is_loud(device):
    """features beyond what any power of two in the kernels' exponent range (2^-40) brings back: NaN"""
    cfg, sd, _ = dance_state_dict(4326, V)
    feats, whs, init, idx = _case(23)
    sd2, feats2 = _scale_features(sd, feats, -70)
    want = _oracle(sd2, feats2, whs, init, idx)
    got, _ = _fast_iterations(device, sd2, feats2, whs, idx, [init] + want[:2])
    for it in range(3):
        bad = torch.isnan(got[it]).flatten(1).any(1)
        assert bool(bad.any())
        if bool((~bad).any()):
            assert float((got[it][~bad] - want[it][~bad]).abs().max()) < ABS_TOL_PX


@pytest.mark.parametrize("k", [10, -10])
def test_dsnake_feature_scale(device, k):
    """Dsnake (zeros padding, demin de_location in pixels, the one-launch init stage): features * 2^-k
    compensated in `fuse` and in the deformers' head convolutions; every stage on the oracle's inputs"""
    cfg, sd, _ = dsnake_state_dict(4327)
    sd = {key: t.clone() for key, t in sd.items()}
    feats = synthetic.make_smooth_features(2, 256, HF, WF, 29, attention=False)
    base_feats = feats.clone()
    boxes = synthetic.make_boxes(N, HF * 4, WF * 4, 31, r_min=8, r_max=60).view(2, N // 2, 4)
    base = snake_ref.dsnake_deform(sd, base_feats, [boxes[0], boxes[1]])
    feats = feats * 2.0 ** -k
    sd["fuse.weight"] *= 2.0 ** k
    for i in range(3):
        sd["deformer%d.head.conv.fc.weight" % i][:, :256] *= 2.0 ** k
    want = snake_ref.dsnake_deform(sd, feats, [boxes[0], boxes[1]])
    for a, b in zip(want["preds"], base["preds"]):
        assert float((a - b).abs().max()) <= 1e-3 * ABS_TOL_PX
    head = modules.DsnakeHead(cfg)
    head.load_state_dict(sd)
    head = head.to(device).eval()
    d = head._get_deformer()
    if os.environ.get("DANCE_PRECISION", "fp16x3") != "fp16x3":
        pytest.skip("the module follows DANCE_PRECISION; this test is about the fast mode's range")
    assert d.precision == "fp16x3"
    nhwc, _ = ops.nchw_to_nhwc(feats.to(device), 0)
    idx = want["image_idx"].to(torch.int32).to(device)
    exts = head.init_stage(d, nhwc, want["init"].to(device), idx)
    raw = snake_ref.dsnake_init(sd, feats, want["init"], want["image_idx"])
    assert_coords_close(exts, raw, "Dsnake init, features * 2^%d" % -k)
    prev = want["octagons"]
    for it in range(3):
        one = d.evolve(1 + it, nhwc, None, prev.to(device).contiguous(), idx, None, use_att=False,
                       clip_first=False, clip_max=(0.0, 0.0))
        assert_coords_close(one, want["preds"][it], "Dsnake iteration %d, features * 2^%d" % (it, -k))
        prev = want["preds"][it]
