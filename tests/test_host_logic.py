"""CPU: host-side logic -- state-dict compatibility, weight packing, config, loud failures."""
import types

import pytest
import torch

from dance_amd import config, modules, ops, snake, synthetic


def test_state_dict_keys_match_reference_layout():
    net = modules.RefineNet(config.dance_r50())
    keys = list(net.state_dict().keys())
    assert keys[:3] == ["bottom_out.0.weight", "bottom_out.0.norm.weight", "bottom_out.0.norm.bias"]
    assert "bottom_out.1.norm.bias" in keys and "bottom_out.0.bias" not in keys
    for i in range(3):
        p = "deformer%d." % i
        for k in ("head.conv.fc.weight", "head.conv.fc.bias", "head.norm.weight", "head.norm.bias",
                  "head.norm.running_mean", "head.norm.running_var", "head.norm.num_batches_tracked",
                  "res6.conv.fc.weight", "fusion.weight", "fusion.bias", "prediction.0.weight",
                  "prediction.2.bias", "prediction.4.weight"):
            assert p + k in keys
    sd = net.state_dict()
    assert sd["deformer0.head.conv.fc.weight"].shape == (128, 258, 9)
    assert sd["deformer2.prediction.0.weight"].shape == (256, 1280, 1)
    assert sum(p.numel() for p in net.parameters()) == 5818182 + 2 * (256 * 256 * 9 + 512)
    d = modules.DsnakeHead(config.dsnake_r50())
    dk = list(d.state_dict().keys())
    assert "fuse.weight" in dk and "init_snake.head.conv.fc.weight" in dk and "deformer2.res0.norm.bias" in dk
    assert dk.index("fuse.weight") < dk.index("init_snake.head.conv.fc.weight") < dk.index("deformer0.head.conv.fc.weight")
    assert d.state_dict()["fuse.weight"].shape == (256, 512, 1)


def test_weight_packing():
    w = torch.randn(128, 258, 9)
    p = ops.pack_conv_weight(w)
    assert p.shape == (9, 272, 128)
    assert torch.equal(p[3, 17], w[:, 17, 3]) and float(p[:, 258:].abs().max()) == 0.0
    net = modules._SnakeNet(128, 258, (1, 1, 1, 2, 2, 4, 4))
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    synthetic.fill_state_dict_(sd, 3)
    packed = snake.SnakeNetWeights(sd, "cpu")
    t = packed.tensors
    assert t["pred0_wg"].shape == (256, 256) and t["pred0_ws"].shape == (1, 1024, 256)
    assert torch.equal(t["pred0_wg"][5], sd["prediction.0.weight"][:, 5, 0])
    assert torch.equal(t["pred0_ws"][0, 7], sd["prediction.0.weight"][:, 256 + 7, 0])
    assert t["pred2_w"].shape == (64, 2) and t["fusion_w"].shape == (1, 1024, 256)
    # folded BatchNorm == F.batch_norm in eval
    x = torch.randn(4, 128, 9)
    want = torch.nn.functional.batch_norm(x, sd["head.norm.running_mean"], sd["head.norm.running_var"],
                                          sd["head.norm.weight"], sd["head.norm.bias"], False, 0.0, 1e-5)
    got = x * t["head_scale"][None, :, None] + t["head_shift"][None, :, None]
    assert float((got - want).abs().max()) < 1e-5
    assert packed.struct.feat_channels == 256 and list(packed.struct.dilation) == [1, 1, 1, 2, 2, 4, 4]


def test_config_from_cfgnode_like():
    NS = types.SimpleNamespace
    sh = NS(NUM_SAMPLING=196, INITIAL="box", ATTENTION=True, INDIVIDUAL_SCALE=True, DETACH=True)
    cfg = NS(MODEL=NS(EDGE_HEAD=NS(CONVS_DIM=256, NORM="GN"), SNAKE_HEAD=sh))
    c = config.SnakeHeadConfig.from_cfg(cfg)
    assert (c.num_sampling, c.initial, c.attention, c.individual_scale, c.convs_dim) == (196, "box", True, True, 256)
    assert c.de_loc_type == "derange" and len(c.num_iter) == 3
    c.validate()
    sh.PRE_OFFSET = True
    with pytest.raises(NotImplementedError, match="PRE_OFFSET"):
        config.SnakeHeadConfig.from_cfg(cfg).validate()
    assert config.dance_r50().num_sampling == 196 and config.dsnake_r50().num_sampling == 128


def test_no_cpu_fallback():
    """the product path fails loudly without a GPU instead of computing on the CPU"""
    net = modules.RefineNet(config.dance_r50(num_convs=0)).eval()
    with pytest.raises(RuntimeError, match="GPU"):
        net.deform(torch.zeros(1, 257, 4, 4), torch.zeros(1, 128, 2), None, (16, 16), torch.ones(1, 2))
    with pytest.raises(RuntimeError, match="GPU"):
        ops.uniform_upsample(torch.zeros(2, 4), 128, kind=ops.POLY_BOX)
    from dance_amd import extreme_utils
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        extreme_utils.roll_array(torch.zeros(1, 2, 2), torch.zeros(1, dtype=torch.int64))


def test_product_never_imports_oracle():
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dance_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f


def test_new_wrappers_refuse_cpu_tensors_and_bad_shapes():
    """No CPU / torch fallback anywhere: the host wrappers of the widened rows raise."""
    import pytest
    import torch
    from dance_amd import modules, rle
    with pytest.raises(RuntimeError, match="GPU"):
        rle.polygon_rle(torch.zeros((2, 8, 2)), (32, 32))
    with pytest.raises(RuntimeError, match="GPU"):
        rle.get_polygon_rles(torch.zeros((2, 16)), (32, 32))
    with pytest.raises(RuntimeError, match="GPU"):
        modules.Attender()(torch.zeros((1, 1, 8, 8)))
    att = modules.Attender()
    assert list(att.state_dict()) == ["0.weight", "0.norm.weight", "0.norm.bias", "1.weight", "1.norm.weight",
                                      "1.norm.bias", "2.weight", "2.bias"]


def test_edge_head_refuses_unbuilt_options():
    import pytest
    from _util import edge_head_cfg
    from dance_amd import modules
    cfg, shapes = edge_head_cfg()
    cfg.MODEL.SNAKE_HEAD.SELECTIVE_REFINE = True
    with pytest.raises(NotImplementedError, match="SELECTIVE_REFINE"):
        modules.EdgeSnakeFPNHead(cfg, shapes)
    cfg, shapes = edge_head_cfg()
    cfg.TEST.GT_IN.ON = True
    with pytest.raises(NotImplementedError, match="GT_IN"):
        modules.EdgeSnakeFPNHead(cfg, shapes)
    cfg, shapes = edge_head_cfg()
    head = modules.EdgeSnakeFPNHead(cfg, shapes)
    keys = list(head.state_dict())
    assert keys[0] == "attender.0.weight" and "p5.4.norm.bias" in keys and "predictor.1.bias" in keys
    assert keys.index("p2.0.weight") < keys.index("predictor.0.weight") < keys.index("refine_head.bottom_out.0.weight")


def test_fcose_locations_and_structures():
    """compute_locations equals the restatement of fcose.py:129-142; Instances / Boxes behave as
    the detector's decode and ml_nms need (fields, indexing, len, cat); CPU tensors are refused."""
    import pytest
    from dance_amd import fcose_outputs
    from dance_amd.structures import Boxes, ExtremePoints, Instances
    from oracle import fcose_ref
    shapes, strides = [(5, 7), (3, 4), (1, 1)], [8, 16, 32]
    feats = [torch.zeros(2, 3, h, w) for h, w in shapes]
    for a, b in zip(fcose_outputs.compute_locations(feats, strides), fcose_ref.compute_locations(shapes, strides)):
        assert torch.equal(a, b)
    a = Instances((10, 20), pred_boxes=Boxes(torch.arange(8.0).view(2, 4)), scores=torch.tensor([0.5, 0.25]),
                  ext_points=ExtremePoints(torch.zeros(2, 4, 2)))
    b = Instances((10, 20), pred_boxes=Boxes(torch.zeros(0, 4)), scores=torch.zeros(0),
                  ext_points=ExtremePoints(torch.zeros(0, 4, 2)))
    c = Instances.cat([a, b, a])
    assert len(c) == 4 and c.image_size == (10, 20) and c.has("scores") and not c.has("pred_classes")
    assert isinstance(c.pred_boxes, Boxes) and isinstance(c.ext_points, ExtremePoints)
    d = c[torch.tensor([3, 0])]
    assert torch.equal(d.scores, torch.tensor([0.25, 0.5])) and len(d.pred_boxes) == 2
    assert torch.equal(c.pred_boxes.area(), torch.tensor([4.0, 4.0, 4.0, 4.0]))
    with pytest.raises(AssertionError):
        a.pred_classes = torch.zeros(3)
    with pytest.raises(AttributeError):
        a.missing
    with pytest.raises(RuntimeError):
        fcose_outputs.decode_candidates([torch.zeros(4, 2)], [torch.zeros(1, 2, 2, 2)], [torch.zeros(1, 4, 2, 2)],
                                        None, [torch.zeros(1, 1, 2, 2)], [8], 0.05, True)
