"""GPU: layout change and bilinear gather + de_location against the golden vectors and the oracle."""
import pytest
import torch

from _util import golden, rel_err
from dance_amd import ops, synthetic
from oracle import snake_ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 257, 50, 76), (1, 256, 13, 7), (3, 9, 5, 6), (2, 65, 8, 8)])
def test_nchw_to_nhwc(device, shape):
    b, ct, hf, wf = shape
    att = 1 if ct % 2 else 0
    x = torch.randn(shape, generator=torch.Generator().manual_seed(1))
    feat, a = ops.nchw_to_nhwc(x.to(device), att)
    assert torch.equal(feat.cpu(), x[:, att:].permute(0, 2, 3, 1).contiguous())
    if att:
        assert torch.equal(a.cpu(), x[:, 0])


def _gather(device, feats, loc, idx, mode, deloc):
    # the golden map has 5 channels: pad to 8 (kernel wants C % 4 == 0), compare the first 5
    c = feats.size(1)
    cp = (c + 3) // 4 * 4
    f = torch.zeros((feats.size(0), cp) + tuple(feats.shape[2:]))
    f[:, :c] = feats
    nhwc, _ = ops.nchw_to_nhwc(f.to(device), 0)
    x, _ = ops.gather_features(nhwc, None, loc.to(device), None if idx is None else idx.to(device),
                               ops.PAD_REFLECTION if mode == "reflection" else ops.PAD_ZEROS, deloc)
    return x.cpu(), c, cp


def test_gather_golden(device):
    g = golden("g2_gather")
    f, loc, idx = g["features"], g["locations"], g["image_idx"]
    for mode, feats, ii, key in (("reflection", f, idx, "reflection_batched"),
                                 ("reflection", f[:1], None, "reflection_single"),
                                 ("zeros", f, idx, "zeros_batched")):
        x, c, cp = _gather(device, feats, loc, ii, mode, "derange")
        assert x.shape == (loc.size(0), cp + 2, loc.size(1))
        assert rel_err(x[:, :c], g[key]) < 1e-5, key
        want = snake_ref.de_location(loc, "derange").permute(0, 2, 1)
        assert rel_err(x[:, cp:], want) < 1e-6


@pytest.mark.parametrize("deloc", ["derange", "demean", "demin", None])
@pytest.mark.parametrize("mode", ["reflection", "zeros"])
def test_gather_full_width_vs_oracle(device, mode, deloc):
    b, c, hf, wf, n, v = 2, 256, 50, 76, 7, 196
    feats = synthetic.make_features(b, c, hf, wf, seed=3, attention=True)
    boxes = synthetic.make_boxes(n, hf * 4, wf * 4, seed=4, r_min=8, r_max=60)
    loc = synthetic.make_circle_contours(boxes, v)
    loc[0] += 150.0                       # partly outside the image
    loc[1, :, 0] = 17.25                  # degenerate in x: derange divides 0/0 -> NaN
    idx = torch.tensor([0, 1, 1, 0, 1, 0, 0], dtype=torch.float32)
    nhwc, att = ops.nchw_to_nhwc(feats.to(device), 1)
    x, a = ops.gather_features(nhwc, att, loc.to(device), idx.to(device),
                               ops.PAD_REFLECTION if mode == "reflection" else ops.PAD_ZEROS, deloc)
    want = snake_ref.get_locations_feature(feats, loc, idx, mode)
    assert rel_err(x[:, :c], want[:, 1:]) < 2e-5
    assert rel_err(a, want[:, 0]) < 2e-5
    if deloc is not None:
        assert rel_err(x[:, c:], snake_ref.de_location(loc, deloc).permute(0, 2, 1)) < 1e-6
    else:
        assert x.shape[1] == c


def test_gather_single_point_and_empty(device):
    feats = synthetic.make_features(2, 8, 6, 9, seed=9, attention=False)
    nhwc, _ = ops.nchw_to_nhwc(feats.to(device), 0)
    loc = torch.tensor([[[3.0, 4.0]], [[20.5, 11.25]], [[-7.0, 100.0]]])
    idx = torch.tensor([0.0, 1.0, 1.0])
    x, _ = ops.gather_features(nhwc, None, loc.to(device), idx.to(device), ops.PAD_ZEROS, None)
    assert rel_err(x, snake_ref.get_locations_feature(feats, loc, idx, "zeros")) < 1e-5
    x, _ = ops.gather_features(nhwc, None, loc[:0].to(device), idx[:0].to(device), ops.PAD_ZEROS, "demin")
    assert x.shape == (0, 10, 1)


@pytest.mark.parametrize("shape", [(3, 257, 40, 56), (2, 65, 7, 9), (1, 5, 3, 5), (2, 256, 25, 38)])
def test_absmax_words(device, shape):
    """the per-image max |feature| words the fast mode scales the head block's input by (csrc/p16.h):
    from the layout pass and from dance_feature_absmax, against torch; order-independent, hence
    bit-reproducible; a NaN wins; an all-zero image gives zeros"""
    from dance_amd._lib import ABSMAX_WORDS
    b, ct, hf, wf = shape
    att = 1 if ct % 2 else 0
    x = torch.randn(shape, generator=torch.Generator().manual_seed(ct)) * 3
    x[0, att:] *= 1e-6 if b > 1 else 1.0
    if b > 1:
        x[1, att:] = 0
        x[1, att + 1, 1, 2] = -7.5e3
    xd = x.to(device)
    nhwc, _ = ops.nchw_to_nhwc(xd, att)
    want = xd[:, att:].abs().flatten(1).max(1)[0]
    for words in (ops.absmax_of(nhwc), ops.feature_absmax(nhwc.clone())):
        assert words.shape == (b, ABSMAX_WORDS) and words.dtype == torch.int32
        got = words.view(torch.float32).max(1)[0]
        assert torch.equal(got, want)
    again, _ = ops.nchw_to_nhwc(xd, att)
    assert torch.equal(ops.absmax_of(again).view(torch.float32).max(1)[0], want)
    # a NaN anywhere in an image: its words say so (bit pattern above every finite value)
    xd[0, att, 0, 0] = float("nan")
    nhwc, _ = ops.nchw_to_nhwc(xd, att)
    for words in (ops.absmax_of(nhwc), ops.feature_absmax(nhwc.clone())):
        assert int((words[0] & 0x7fffffff).max()) > 0x7f800000
        if b > 1:
            assert torch.equal(words[1:].view(torch.float32).max(1)[0], want[1:])
    z, _ = ops.nchw_to_nhwc(torch.zeros(shape, device=device), att)
    assert int(ops.absmax_of(z).abs().max()) == 0
