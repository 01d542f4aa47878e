"""dance_gather_head16w (gather + de_location + head block on 16-wide vertex tiles, the
de_location channels as one im2col K step -- csrc/head16.hip) against dance_gather_head16 (32-wide
tiles, same arithmetic up to summation order) and against plain torch fp32 -- reference:
RefineNet.evolve snake_head.py:1639-1652, _SnakeNet.head :233."""
import pytest
import torch

from dance_amd import ops, snake, synthetic
from dance_amd._lib import check, current_stream, lib, ptr
from _util import dance_state_dict

pytestmark = pytest.mark.gpu


def _run(fn_name, s16, nhwc, att, loc, idx, n, v, pad, deloc, terms, use_att=True, wd32=False):
    dev = nhwc.device
    b, hf, wf, c = nhwc.shape
    y = torch.zeros((n, 8, 2, 16, v, 8), dtype=torch.float16, device=dev)
    att_o = torch.zeros((n, v), device=dev)
    with torch.cuda.device(dev):
        st = current_stream()
        extra = (s16.head_wd if fn_name == "dance_gather_head16w" or wd32 else None,)
        check(getattr(lib(), fn_name)(ptr(nhwc), ptr(att) if use_att else None, ptr(loc), ptr(idx), s16.head_w, *extra,
                                      s16.head_bs, s16.head_winv, s16.head_scale, s16.head_shift, ptr(y), y.stride(0),
                                      ptr(att_o) if use_att else None, n, v, b, hf, wf, c, pad, deloc, terms, st),
              fn_name)
    return ops.from_p16(y[:, :1].contiguous()), att_o


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
def test_head16w_equals_head16(device, v):
    """every instantiated tile count; above 208 vertices the ring runs as two half rings (tiles split
    7+7, 8+7, 8+8) in two workgroups, each gathering 4 halo vertices either side"""
    cfg, sd, _ = dance_state_dict(77, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer2."), device)
    s16 = net.struct16
    nhwc, att, loc, idx = _inputs(device, v)
    n = loc.shape[0]
    for terms, pad, deloc in [(3, ops.PAD_REFLECTION, "derange"), (1, ops.PAD_REFLECTION, "derange"),
                              (3, ops.PAD_ZEROS, "demin"), (3, ops.PAD_REFLECTION, "demean")]:
        a, att_a = _run("dance_gather_head16w", s16, nhwc, att, loc, idx, n, v, pad, ops.DELOC[deloc], terms)
        b, att_b = _run("dance_gather_head16", s16, nhwc, att, loc, idx, n, v, pad, ops.DELOC[deloc], terms)
        scale = float(b.abs().max())
        assert scale > 0 and bool(torch.isfinite(a).all())
        tol = 1e-5 if terms == 3 else 5e-3                      # summation order (one term: fp16 operands)
        assert float((a - b).abs().max()) <= tol * scale, (v, terms, deloc)
        assert torch.equal(att_a, att_b)


@pytest.mark.parametrize("v", [65, 68, 80, 81, 96, 97, 100, 112, 113, 128])
def test_head16w_small_n_half_rings(device, v, monkeypatch):
    """65..128-vertex rings with at most DANCE_HEAD_SMALL_N (128) contours run as two half rings (the one
    image's-detections case); test_head16w_equals_head16 above covers them too (29 contours) -- here the
    whole-ring kernel (threshold 0) and the half-ring one are compared directly"""
    cfg, sd, _ = dance_state_dict(82, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer1."), device)
    nhwc, att, loc, idx = _inputs(device, v, n=37)
    for terms, deloc in [(3, "derange"), (1, "demin")]:
        monkeypatch.setenv("DANCE_HEAD_SMALL_N", "0")
        whole, att_w = _run("dance_gather_head16w", net.struct16, nhwc, att, loc, idx, 37, v, ops.PAD_REFLECTION,
                            ops.DELOC[deloc], terms)
        monkeypatch.delenv("DANCE_HEAD_SMALL_N")
        halves, att_h = _run("dance_gather_head16w", net.struct16, nhwc, att, loc, idx, 37, v, ops.PAD_REFLECTION,
                             ops.DELOC[deloc], terms)
        assert torch.equal(halves, whole)                       # same K order per output: bit-identical
        assert torch.equal(att_h, att_w)


@pytest.mark.parametrize("v", [129, 132, 160, 164, 192, 196, 208])
def test_head16w_mid_half_rings(device, v, monkeypatch):
    """129..208-vertex rings run as two half rings of 5, 6 or 7 tiles (default) -- against the
    one-workgroup-per-contour kernels (DANCE_HEAD_HALVES_MID=0): same K order per output, bit-identical"""
    cfg, sd, _ = dance_state_dict(83, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer2."), device)
    nhwc, att, loc, idx = _inputs(device, v, n=21)
    for terms, deloc in [(3, "derange"), (1, "demean")]:
        monkeypatch.setenv("DANCE_HEAD_HALVES_MID", "0")
        whole, att_w = _run("dance_gather_head16w", net.struct16, nhwc, att, loc, idx, 21, v, ops.PAD_REFLECTION,
                            ops.DELOC[deloc], terms)
        monkeypatch.delenv("DANCE_HEAD_HALVES_MID")
        halves, att_h = _run("dance_gather_head16w", net.struct16, nhwc, att, loc, idx, 21, v, ops.PAD_REFLECTION,
                             ops.DELOC[deloc], terms)
        assert torch.equal(halves, whole)
        assert torch.equal(att_h, att_w)


def test_head16w_vs_torch(device):
    """torch fp32: grid_sample (reflection) -> derange de_location -> circular Conv1d k = 9 -> ReLU ->
    BatchNorm (eval), from the same weights."""
    cfg, sd, _ = dance_state_dict(78, 128)
    sub = snake.sub_state_dict(sd, "deformer0.")
    net = snake.SnakeNetWeights(sub, device)
    v = 196
    nhwc, att, loc, idx = _inputs(device, v, n=13, seed=9)
    n = loc.shape[0]
    got, _ = _run("dance_gather_head16w", net.struct16, nhwc, att, loc, idx, n, v, ops.PAD_REFLECTION,
                  ops.DELOC["derange"], 3)
    sampled, _ = ops.gather_features(nhwc, att, loc, idx, ops.PAD_REFLECTION, "derange")      # (n, 258, v) exact fp32
    w = {k: t.to(device).float() for k, t in sub.items()}
    x = torch.cat([sampled[:, :, -4:], sampled, sampled[:, :, :4]], dim=2)
    y = torch.relu(torch.nn.functional.conv1d(x, w["head.conv.fc.weight"], w["head.conv.fc.bias"]))
    want = torch.nn.functional.batch_norm(y, w["head.norm.running_mean"], w["head.norm.running_var"],
                                          w["head.norm.weight"], w["head.norm.bias"], False, 0.0, 1e-5)
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_head16w_no_deloc_and_no_attention(device):
    cfg, sd, _ = dance_state_dict(79, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer1."), device)
    v = 100
    nhwc, att, loc, idx = _inputs(device, v)
    n = loc.shape[0]
    a, _ = _run("dance_gather_head16w", net.struct16, nhwc, att, loc, idx, n, v, ops.PAD_ZEROS, ops.DELOC[None], 3, False)
    b, _ = _run("dance_gather_head16", net.struct16, nhwc, att, loc, idx, n, v, ops.PAD_ZEROS, ops.DELOC[None], 3, False)
    assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


@pytest.mark.parametrize("v", [40, 128, 212, 224, 256])
def test_head16_de_location_k_step_equals_padded_chunk(device, v):
    """The 32-wide kernel with the de_location channels as one K step (wd16) against the same kernel
    with the zero-padded chunk through the 9 ring taps."""
    cfg, sd, _ = dance_state_dict(80, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer0."), device)
    nhwc, att, loc, idx = _inputs(device, v, n=11)
    n = loc.shape[0]
    for terms in (3, 1):
        a, _ = _run("dance_gather_head16", net.struct16, nhwc, att, loc, idx, n, v, ops.PAD_REFLECTION,
                    ops.DELOC["derange"], terms, wd32=True)
        b, _ = _run("dance_gather_head16", net.struct16, nhwc, att, loc, idx, n, v, ops.PAD_REFLECTION,
                    ops.DELOC["derange"], terms)
        tol = 1e-5 if terms == 3 else 5e-3
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()), (v, terms)


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
                                              s16.head_winv, s16.head_scale, s16.head_shift, ptr(y), y.stride(0),
                                              None, n, v, 2, 40, 56, c, ops.PAD_ZEROS, deloc, 3, current_stream())

    assert call(0, 128, 256, s16.head_wd, ops.DELOC["demin"]) == 0
    for bad in [(5, 128, 96, s16.head_wd, ops.DELOC["demin"]), (5, 8, 256, s16.head_wd, ops.DELOC["demin"]),
                (5, 260, 256, s16.head_wd, ops.DELOC["demin"]), (5, 128, 256, None, ops.DELOC["demin"])]:
        assert call(*bad) != 0
    assert call(5, 128, 256, None, ops.DELOC[None]) == 0          # no de_location: no K-step tensor needed
    torch.cuda.synchronize()
