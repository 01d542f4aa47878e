"""GPU: RefineNet.bottom_out (conv3x3 + GroupNorm + ReLU) on the HIP path,
against the golden output of the reference's own module and against the torch-CPU oracle."""
import pytest
import torch

from _util import golden, rel_err
from dance_amd import config, modules, ops, synthetic
from oracle import snake_ref

pytestmark = pytest.mark.gpu


def _state_dict(seed):
    cfg = config.dance_r50(num_sampling=128, num_convs=2)
    net = modules.RefineNet(cfg)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    return cfg, sd, synthetic.fill_state_dict_(sd, seed)


def _layers(sd, device, n=2):
    out = []
    for i in range(n):
        w16, s = ops.pack_conv3x3_weight16(sd["bottom_out.%d.weight" % i].to(device))
        out.append((w16, s, sd["bottom_out.%d.norm.weight" % i].to(device).contiguous(),
                    sd["bottom_out.%d.norm.bias" % i].to(device).contiguous()))
    return out


def test_bottom_out_golden(device):
    g = golden("g8_bottom_out")
    cfg, sd, ck = _state_dict(int(g["weight_seed"]))
    assert abs(ck - float(g["weight_checksum"])) < 1e-6 * ck
    feats = synthetic.make_smooth_features(2, 256, 9, 70, int(g["feat_seed"]), attention=False, cell=4)
    got = ops.bottom_out16(feats.to(device), 0, _layers(sd, device))
    want = g["out"].permute(0, 2, 3, 1)
    assert got.shape == want.shape
    assert rel_err(got, want) < 1e-4


@pytest.mark.parametrize("shape", [(1, 2, 64), (2, 7, 65), (1, 5, 130), (3, 4, 3)])
def test_bottom_out_shapes_vs_oracle(device, shape):
    """partial tiles in both directions, odd heights, attention channel skipped, one layer and two"""
    b, h, w = shape
    cfg, sd, _ = _state_dict(5)
    feats = synthetic.make_features(b, 256, h, w, 100 + w, attention=True)
    for n in (1, 2):
        sub = {k: v for k, v in sd.items() if k.startswith("bottom_out.") and int(k.split(".")[1]) < n}
        want = snake_ref.bottom_out(sub, feats[:, 1:]).permute(0, 2, 3, 1)
        got = ops.bottom_out16(feats.to(device), 1, _layers(sd, device, n))
        assert rel_err(got, want) < 1e-4, (shape, n)
        # the output carries its per-image max |value| words (the deformer's input scale), measured on the way out
        words = ops.absmax_of(got)
        assert torch.equal(words.view(torch.float32).max(1)[0], got.abs().flatten(1).max(1)[0])
        # the same tensor in channels_last memory is read in place (dance_nhwc_to_p16img):
        # 257 channels with the attention plane skipped (scalar loads), 256 (16-byte loads)
        cl = feats.to(device).contiguous(memory_format=torch.channels_last)
        assert torch.equal(ops.bottom_out16(cl, 1, _layers(sd, device, n)), got)
        cl = feats[:, 1:].to(device).contiguous(memory_format=torch.channels_last)
        if h * w > 1:
            assert not cl.is_contiguous()
        assert torch.equal(ops.bottom_out16(cl, 0, _layers(sd, device, n)), got)


def test_module_bottom_out_feeds_the_gather(device):
    """the drop-in module runs its bottom_out on the HIP path and hands NHWC to the deformer"""
    cfg, sd, _ = _state_dict(7)
    net = modules.RefineNet(cfg)
    net.load_state_dict(sd)
    net = net.to(device).eval()
    feats = synthetic.make_smooth_features(2, 256, 24, 40, 9)
    d, nhwc, att = net._features(feats.to(device))
    want = snake_ref.bottom_out(sd, feats[:, 1:]).permute(0, 2, 3, 1)
    assert nhwc.shape == want.shape and nhwc.is_contiguous()
    assert rel_err(nhwc, want) < 1e-4
    assert torch.equal(att.cpu(), feats[:, 0])


def test_dsnake_module_with_bottom_out(device):
    """DsnakeHead end to end with NUM_CONVS = 2 (dsnake_head.py:788-830): HIP feature prep ->
    init -> octagon -> 3 evolves, against the oracle's composition of the same stages."""
    from _util import REL_TOL, FakeBoxes, FakeInstances, coord_rel_err
    cfg = config.dsnake_r50(num_convs=2)
    head = modules.DsnakeHead(cfg)
    sd = {k: v.clone() for k, v in head.state_dict().items()}
    synthetic.fill_state_dict_(sd, 21)
    head.load_state_dict(sd)
    head = head.to(device).eval()
    hf, wf = 40, 56
    feats = synthetic.make_smooth_features(2, 256, hf, wf, 5, attention=False)
    boxes = synthetic.make_boxes(10, hf * 4, wf * 4, 6, r_min=10, r_max=50).view(2, 5, 4)
    insts = [FakeInstances((hf * 4, wf * 4), pred_boxes=FakeBoxes(boxes[i].to(device))) for i in range(2)]
    with torch.no_grad():
        res, _ = head(feats.to(device), insts, None)
    got = torch.cat([r.pred_polys.tensor for r in res])
    prepped = snake_ref.bottom_out(sd, feats)
    want = snake_ref.dsnake_deform(sd, prepped, [boxes[0], boxes[1]])
    assert coord_rel_err(got, want["preds"][-1]) < REL_TOL


@pytest.mark.parametrize("shape", [(1, 256, 25, 38), (2, 128, 7, 5), (1, 8, 1, 1), (3, 12, 3, 100)])
def test_upsample2x_add_vs_torch(device, shape):
    """acc + bilinear x2 (align_corners=False) in one launch against torch's own two ops on the
    same GPU: 1e-6 (the four weights are exact; only the summation's contraction differs)."""
    import torch.nn.functional as F
    b, c, h, w = shape
    g = torch.Generator().manual_seed(h * w)
    src = torch.randn(b, c, h, w, generator=g).to(device).contiguous(memory_format=torch.channels_last)
    acc = torch.randn(b, c, 2 * h, 2 * w, generator=g).to(device).contiguous(memory_format=torch.channels_last)
    src, acc = src.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2), acc.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    up = F.interpolate(src, scale_factor=2, mode="bilinear", align_corners=False)
    got = ops.upsample2x_add(src)
    assert got.shape == up.shape and float((got - up).abs().max()) <= 1e-6 * float(up.abs().max())
    want = acc + up
    got = ops.upsample2x_add(src, acc)
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())
    keep = acc.clone()
    out = ops.upsample2x_add(src, acc, inplace=True)
    assert out.data_ptr() == acc.data_ptr() and torch.equal(out, got) and not torch.equal(acc, keep)
    with pytest.raises(RuntimeError):
        ops.upsample2x_add(src.contiguous() if h * w > 1 and c > 1 else src.cpu())


def test_bottom_out_is_bit_reproducible(device):
    """the GroupNorm statistics are per-workgroup partial sums folded in a fixed order (no atomics):
    repeated runs over a map of a few thousand workgroups give identical bits, statistics included"""
    cfg, sd, _ = _state_dict(11)
    feats = synthetic.make_features(2, 256, 100, 152, 123, attention=True).to(device)
    layers = _layers(sd, device)
    first = ops.bottom_out16(feats, 1, layers)
    for _ in range(4):
        assert torch.equal(ops.bottom_out16(feats, 1, layers), first)
    assert torch.equal(ops.absmax_of(ops.bottom_out16(feats, 1, layers)), ops.absmax_of(first))


@pytest.mark.parametrize("shape", [(2, 256, 48, 64, 1), (1, 128, 5, 7, 3), (3, 260, 9, 4, 2), (1, 4, 1, 1, 1)])
def test_conv1x1_nhwc_vs_torch(device, shape):
    """the edge predictor's last layer (edge_det.py:197) as a fixed-order kernel: against F.conv2d, with and
    without bias, and bit-identical from run to run"""
    b, c, h, w, k = shape
    g = torch.Generator().manual_seed(c + k)
    x = torch.randn((b, c, h, w), generator=g).to(device).contiguous(memory_format=torch.channels_last)
    if not ops.is_nhwc(x):          # (1x1 maps: every layout is dense)
        x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    wt = (torch.randn((k, c, 1, 1), generator=g) / c ** 0.5).to(device)
    bias = torch.randn((k,), generator=g).to(device)
    for bs in (bias, None):
        got = ops.conv1x1_nhwc(x, wt, bs)
        want = torch.nn.functional.conv2d(x.double(), wt.double(), None if bs is None else bs.double())
        assert got.shape == (b, k, h, w) and got.is_contiguous()
        assert float((got.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
        assert torch.equal(ops.conv1x1_nhwc(x, wt, bs), got)
