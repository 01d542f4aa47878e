"""dance_gather_head16w (gather + de_location + head block on 16-wide vertex tiles, the
de_location channels as one im2col K step -- csrc/head16.hip) against plain torch fp32 from the
same weights, at every instantiated tile count, with and without the input scale -- reference:
RefineNet.evolve snake_head.py:1639-1652, _SnakeNet.head :233."""
import ctypes

import pytest
import torch

from dance_amd import ops, snake, synthetic
from dance_amd._lib import PolicyStruct, check, current_stream, lib, ptr
from _fast_ref import head_ref, true_state, w32
from _util import dance_state_dict

pytestmark = pytest.mark.gpu


def _run(net, nhwc, att, loc, idx, n, v, pad, deloc, terms, use_att=True, scaled=True, policy=None, head=None):
    """-> (head block output in true units (n, 128, v), sampled attention (n, v)).
    head: (w16, bias_scaled, w_inv_scale) replacing the net's packed head weights"""
    s16 = net.struct16
    head_w, head_bs, head_winv = (s16.head_w, s16.head_bs, s16.head_winv) if head is None else \
        (head[0].data_ptr(), head[1].data_ptr(), head[2])
    dev = nhwc.device
    b, hf, wf, c = nhwc.shape
    y = torch.zeros((n, 8, 2, 16, v, 8), dtype=torch.float16, device=dev)
    att_o = torch.zeros((n, v), device=dev)
    words = ops.absmax_of(nhwc) if scaled else None
    with torch.cuda.device(dev):
        check(lib().dance_gather_head16w(ptr(nhwc), ptr(att) if use_att else None, ptr(loc), ptr(idx), head_w,
                                         s16.head_wd, head_bs, head_winv, s16.head_wd_winv, s16.head_scale, s16.head_shift,
                                         ptr(y), y.stride(0), ptr(att_o) if use_att else None, n, v, b, hf, wf, c,
                                         pad, deloc, terms, ptr(words),
                                         None if policy is None else ctypes.byref(policy), current_stream()),
              "dance_gather_head16w")
    return true_state(net, y[:, :1].contiguous()), att_o


def _inputs(device, v, n=29, seed=5):
    b, hf, wf = 2, 40, 56
    feats = synthetic.make_features(b, 256, hf, wf, seed).to(device)
    nhwc, att = ops.nchw_to_nhwc(feats, 1)
    boxes = synthetic.make_boxes(n, hf * 4, wf * 4, v, r_min=6, r_max=50)
    loc = synthetic.make_circle_contours(boxes, v).to(device)
    loc[3] += 400.0                                             # leaves the image
    idx = (torch.arange(n) % b).to(torch.int32).to(device)
    return nhwc, att, loc, idx


@pytest.mark.parametrize("v", [16, 20, 32, 36, 40, 48, 64, 68, 80, 96, 100, 112, 128, 132, 160, 164, 192, 196, 208,
                               209, 212, 224, 225, 240, 241, 250, 256])
def test_head16w_vs_torch(device, v):
    """every instantiated tile count (above 128 vertices, and here -- 29 contours -- from 65 on, the ring
    runs as two half rings in two workgroups, each gathering 4 halo vertices either side), three
    de_location types, both paddings, both term counts: torch fp32 grid_sample-equivalent gather ->
    de_location -> circular Conv1d k = 9 -> ReLU -> BatchNorm (eval) from the same weights"""
    cfg, sd, _ = dance_state_dict(77, 128)
    sub = snake.sub_state_dict(sd, "deformer2.")
    net = snake.SnakeNetWeights(sub, device)
    w = w32(sub, device)
    nhwc, att, loc, idx = _inputs(device, v)
    n = loc.shape[0]
    for terms, pad, deloc in [(3, ops.PAD_REFLECTION, "derange"), (1, ops.PAD_REFLECTION, "derange"),
                              (3, ops.PAD_ZEROS, "demin"), (3, ops.PAD_REFLECTION, "demean")]:
        got, att_a = _run(net, nhwc, att, loc, idx, n, v, pad, ops.DELOC[deloc], terms)
        want, att_b = head_ref(w, nhwc, att, loc, idx, pad, deloc)
        scale = float(want.abs().max())
        assert scale > 0 and bool(torch.isfinite(got).all())
        tol = 1e-5 if terms == 3 else 5e-3                      # (one term: plain fp16 operands)
        assert float((got - want).abs().max()) <= tol * scale, (v, terms, deloc)
        assert torch.equal(att_a, att_b)


@pytest.mark.parametrize("v", [40, 128, 196])
def test_head16w_unscaled_input_agrees(device, v):
    """feat_absmax NULL (the input split as it is) against the scaled default: O(1) features lose
    nothing either way, so the two agree to summation-order level"""
    cfg, sd, _ = dance_state_dict(84, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer0."), device)
    nhwc, att, loc, idx = _inputs(device, v, n=300)
    for deloc, pad in (("derange", ops.PAD_REFLECTION), ("demin", ops.PAD_ZEROS)):
        a, _ = _run(net, nhwc, att, loc, idx, 300, v, pad, ops.DELOC[deloc], 3)
        b, _ = _run(net, nhwc, att, loc, idx, 300, v, pad, ops.DELOC[deloc], 3, scaled=False)
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


@pytest.mark.parametrize("v", [65, 68, 80, 81, 96, 97, 100, 112, 113, 128])
def test_head16w_small_n_half_rings(device, v):
    """65..128-vertex rings with at most DANCE_FEW_HEAD (128) contours run as two half rings (the one
    image's-detections case): the whole-ring kernel (policy.few_contour_forms = 0) and the half-ring one
    compared directly -- same K order per output, same per-contour input scale: bit-identical"""
    cfg, sd, _ = dance_state_dict(82, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer1."), device)
    nhwc, att, loc, idx = _inputs(device, v, n=37)
    for terms, deloc in [(3, "derange"), (1, "demin")]:
        whole, att_w = _run(net, nhwc, att, loc, idx, 37, v, ops.PAD_REFLECTION, ops.DELOC[deloc], terms,
                            policy=PolicyStruct(few_contour_forms=0))
        halves, att_h = _run(net, nhwc, att, loc, idx, 37, v, ops.PAD_REFLECTION, ops.DELOC[deloc], terms)
        assert torch.equal(halves, whole)
        assert torch.equal(att_h, att_w)


def test_head16w_no_deloc_and_no_attention(device):
    cfg, sd, _ = dance_state_dict(79, 128)
    sub = snake.sub_state_dict(sd, "deformer1.")
    net = snake.SnakeNetWeights(sub, device)
    v = 100
    nhwc, att, loc, idx = _inputs(device, v)
    n = loc.shape[0]
    # without de_location the kernel expects weights packed for the feature channels alone
    w = w32(sub, device)
    w16, sc = ops.pack_conv_weight16(w["head.conv.fc.weight"][:, :256].contiguous())
    bias_s = (w["head.conv.fc.bias"] * sc).contiguous()
    a, _ = _run(net, nhwc, att, loc, idx, n, v, ops.PAD_ZEROS, ops.DELOC[None], 3, False,
                head=(w16, bias_s, 1.0 / sc))
    b, _ = head_ref(w, nhwc, None, loc, idx, ops.PAD_ZEROS, None)
    assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


def test_head16w_argument_checks(device):
    """Channel counts that are not a multiple of 64, rings outside 16..256 and a missing de_location
    K-step tensor are refused before any launch; n = 0 is a no-op."""
    cfg, sd, _ = dance_state_dict(81, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer0."), device)
    s16 = net.struct16
    nhwc, att, loc, idx = _inputs(device, 128, n=5)
    y = torch.zeros((5, 8, 2, 16, 128, 8), dtype=torch.float16, device=device)

    def call(n, v, c, wd, deloc):
        with torch.cuda.device(device):
            return lib().dance_gather_head16w(ptr(nhwc), ptr(att), ptr(loc), ptr(idx), s16.head_w, wd, s16.head_bs,
                                              s16.head_winv, s16.head_wd_winv, s16.head_scale, s16.head_shift, ptr(y), y.stride(0),
                                              None, n, v, 2, 40, 56, c, ops.PAD_ZEROS, deloc, 3, None, None,
                                              current_stream())

    assert call(0, 128, 256, s16.head_wd, ops.DELOC["demin"]) == 0
    for bad in [(5, 128, 96, s16.head_wd, ops.DELOC["demin"]), (5, 8, 256, s16.head_wd, ops.DELOC["demin"]),
                (5, 260, 256, s16.head_wd, ops.DELOC["demin"]), (5, 128, 256, None, ops.DELOC["demin"])]:
        assert call(*bad) != 0
    assert call(5, 128, 256, None, ops.DELOC[None]) == 0          # no de_location: no K-step tensor needed
    torch.cuda.synchronize()
