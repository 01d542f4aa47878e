"""GPU: the drop-in modules behind the reference's forward(features, pred_instances, targets)."""
import pytest
import torch

from _util import REL_TOL, FakeBoxes, FakeInstances, coord_rel_err, dance_state_dict, golden
from dance_amd import config, modules, synthetic
from dance_amd.structures import ExtremePoints
from test_oracle_golden import dance_case_inputs

pytestmark = pytest.mark.gpu


def _module(cfg, sd, device):
    net = modules.RefineNet(cfg)
    net.load_state_dict(sd)
    return net.to(device).eval()


def test_refinenet_single_test_box(device):
    g, cfg, sd, feats, image_idx, sizes = dance_case_inputs("smooth_single_box_v128")
    net = _module(cfg, sd, device)
    inst = FakeInstances(sizes, pred_boxes=FakeBoxes(g["boxes"][0].to(device)))
    with torch.no_grad():
        res, losses = net(feats.to(device), [inst], None)
    assert losses == {} and len(res) == 1
    assert res[0].pred_polys.tensor.shape == g["pred2"].shape
    assert coord_rel_err(res[0].pred_polys.tensor, g["pred2"]) < REL_TOL
    with pytest.raises(NotImplementedError):
        net.train()(feats.to(device), [inst], None)


def test_refinenet_single_test_octagon(device):
    g, cfg, sd, feats, image_idx, sizes = dance_case_inputs("smooth_single_octagon_v128")
    cfg.initial = "octagon"
    net = _module(cfg, sd, device)
    ext = ExtremePoints(g["ext_points"][0].clone().to(device))
    inst = FakeInstances(sizes, pred_boxes=FakeBoxes(g["boxes"][0].to(device)), ext_points=ext)
    with torch.no_grad():
        res, _ = net(feats.to(device), [inst], None)
    assert coord_rel_err(res[0].pred_polys.tensor, g["pred2"]) < REL_TOL     # NaN rows included
    # the in-place fit_to_box pulled the stray point back inside its box
    assert float(ext.tensor[1, 0, 0]) >= float(ext.tensor[1, 1, 0])
    # an image without detections returns the instances untouched
    empty = FakeInstances(sizes, pred_boxes=FakeBoxes(torch.zeros((0, 4), device=device)))
    cfg.initial = "box"
    res, _ = _module(cfg, sd, device)(feats.to(device), [empty], None)
    assert res[0] is empty and not empty.has("pred_polys")


def test_refinenet_batched(device):
    g, cfg, sd, feats, image_idx, sizes = dance_case_inputs("smooth_batched_box_v196")
    cfg.vis_path = True
    net = _module(cfg, sd, device)
    insts = [FakeInstances(sizes[i], pred_boxes=FakeBoxes(g["boxes"][i].to(device))) for i in range(3)]
    with torch.no_grad():
        res, _ = net(feats.to(device), insts, None)
    got = torch.cat([r.pred_polys.tensor for r in res])
    assert coord_rel_err(got, g["pred2"]) < REL_TOL
    assert res[0].pred_path.shape == (3, 4, 196, 2)


def test_refinenet_batched_many_instances_stream_policies(device, monkeypatch):
    """The batched module path at sizes where its stream policy switches (modules.RefineNet.deform):
    6 x 100 contours (one stream, 88 peeled to the side stream on a 256-CU GPU), 8 x 128 = 1024 (whole
    rounds: two streams) -- each against DANCE_STREAMS=1 with DANCE_PEEL=0, bit for bit, box and octagon
    initial contours."""
    g, cfg, sd, feats, image_idx, sizes = dance_case_inputs("smooth_batched_box_v196")
    cfg.vis_path = True
    for initial in ("box", "octagon"):
        cfg.initial = initial
        net = _module(cfg, sd, device)
        for b, per in ((6, 100), (8, 128)):
            fb = feats[:1].repeat(b, 1, 1, 1).to(device)
            fb = fb + 0.01 * torch.arange(b, device=device).view(b, 1, 1, 1)
            h, w = sizes[0]
            boxes = synthetic.make_boxes(b * per, h, w, 31 + b).view(b, per, 4).to(device)

            def run():
                insts = []
                for i in range(b):
                    kw = {}
                    if initial == "octagon":
                        bx = boxes[i]
                        cx, cy = (bx[:, 0] + bx[:, 2]) / 2, (bx[:, 1] + bx[:, 3]) / 2
                        ext = torch.stack([torch.stack([cx, bx[:, 1]], 1), torch.stack([bx[:, 0], cy], 1),
                                           torch.stack([cx, bx[:, 3]], 1), torch.stack([bx[:, 2], cy], 1)], 1)
                        kw["ext_points"] = ExtremePoints(ext.clone())
                    insts.append(FakeInstances((h, w), pred_boxes=FakeBoxes(boxes[i].clone()), **kw))
                with torch.no_grad():
                    res, _ = net(fb, insts, None)
                return torch.cat([r.pred_polys.tensor for r in res])

            monkeypatch.delenv("DANCE_STREAMS", raising=False)
            monkeypatch.delenv("DANCE_PEEL", raising=False)
            got = run()
            monkeypatch.setenv("DANCE_STREAMS", "1")
            monkeypatch.setenv("DANCE_PEEL", "0")
            want = run()
            assert got.shape == (b * per, 196, 2)
            assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(want, nan=-7.0)), (initial, b, per)


def test_bottom_out_channels_last_feeds_the_gather(device):
    """with bottom_out in place (NUM_CONVS=2) the torch convs run channels_last and their
    output is consumed without a layout pass; result = the oracle on the same features."""
    from oracle import snake_ref
    cfg = config.dance_r50(num_sampling=128, num_convs=2)
    net = modules.RefineNet(cfg)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    synthetic.fill_state_dict_(sd, 99)
    assert "bottom_out.0.norm.weight" in sd and "bottom_out.1.weight" in sd and "bottom_out.0.bias" not in sd
    net.load_state_dict(sd)
    net = net.to(device).eval()
    feats = synthetic.make_smooth_features(1, 256, 24, 32, 5).to(device)
    boxes = synthetic.make_boxes(4, 96, 128, 6, r_min=6, r_max=30)
    inst = FakeInstances((96, 128), pred_boxes=FakeBoxes(boxes.clone().to(device)))
    with torch.no_grad():
        res, _ = net(feats, [inst], None)
        x = feats[:, 1:]
        for conv in net.bottom_out:
            x = conv(x)
        full = torch.cat([feats[:, :1], x], dim=1).cpu()
    init, _ = snake_ref.uniform_upsample(snake_ref.box_polygon(boxes)[None], 128)
    b = boxes.clone()
    whs = torch.stack([b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1)
    want = snake_ref.dance_deform(sd, full, init, None, (96, 128), whs)
    assert coord_rel_err(res[0].pred_polys.tensor, want[-1]) < REL_TOL


def test_timer_scopes_match_the_reference_names(device):
    """dance_amd.timer records device time under the reference's scope names (timer.py,
    snake_head.py:1786-1864) when enabled, and nothing when disabled."""
    from dance_amd import timer
    g, cfg, sd, feats, image_idx, sizes = dance_case_inputs("smooth_single_box_v128")
    net = _module(cfg, sd, device)
    inst = FakeInstances(sizes, pred_boxes=FakeBoxes(g["boxes"][0].to(device)))
    timer.reset()
    with torch.no_grad():
        net(feats.to(device), [inst], None)
    assert timer.total_times() == {}
    timer.enable_all()
    try:
        with torch.no_grad():
            net(feats.to(device), [inst], None)
        stats = timer.total_times()
    finally:
        timer.disable_all()
        timer.reset()
    assert set(stats) == {"contour_sampling", "snake_feat_prep", "snake_deformation", "snake_postprocess"}
    assert all(ms >= 0.0 and n == 1 for ms, n in stats.values())
    assert stats["snake_deformation"][0] > stats["contour_sampling"][0]


def test_edge_snake_fpn_head_end_to_end(device):
    """The reference's EdgeSnakeFPNHead.forward (edge_det.py:187-395) on one image: stock torch
    scale heads + predictor, HIP attender, HIP feature prep + contour path; against the golden
    output of the reference's own module (tests/golden/g10_edge_head.npz)."""
    from _util import coord_rel_err, edge_head_cfg, edge_head_inputs, golden
    g = golden("g10_edge_head")
    cfg, shapes = edge_head_cfg()
    head = modules.EdgeSnakeFPNHead(cfg, shapes)
    sd = {k: v.clone() for k, v in head.state_dict().items()}
    ck = synthetic.fill_state_dict_(sd, int(g["weight_seed"]))
    assert abs(ck - float(g["weight_checksum"])) < 1e-6 * ck
    head.load_state_dict(sd)
    head = head.to(device).eval()
    feats, boxes = edge_head_inputs(int(g["input_seed"]))
    inst = FakeInstances((192, 256), pred_boxes=FakeBoxes(boxes.clone().to(device)))
    with torch.no_grad():
        att, losses, new = head({k: v.to(device) for k, v in feats.items()}, [inst], None)
    assert losses == {} and att.shape == (1, 1, 48, 64)
    assert float((att.cpu() - g["att_map"]).abs().max()) < 1e-4
    polys = new[0].pred_polys.tensor
    assert polys.shape == (6, 128, 2)
    assert coord_rel_err(polys, g["pred_polys"]) < 1e-4
    # masks of the result, the reference evaluator's next step (coco_evaluation.py:352-357)
    from dance_amd import rle
    from oracle import rle_ref
    new[0].pred_polys.clip((192, 256))
    got = rle.get_polygon_rles(new[0].pred_polys.flatten(), (192, 256))
    want = rle_ref.get_polygon_rles(new[0].pred_polys.flatten().cpu().numpy(), (192, 256))
    assert got == want and len(got) == 6


def test_snake_fpn_head_is_the_dsnake_caller(device):
    """dsnake_head.py:182-194: SnakeFPNHead hands features["p2"] to DsnakeHead."""
    cfg = config.dsnake_r50(num_convs=0)
    head = modules.SnakeFPNHead(cfg).to(device).eval()
    assert [k for k in head.state_dict()][0].startswith("refine_head.")
    with pytest.raises(NotImplementedError):
        head.train()({"p2": None}, None, None)


def test_detector_decode_to_contours_to_masks(device):
    """The three widened steps compose: FCOSE head outputs -> FCOSEOutputs.predict_proposals
    (dance_amd Instances) -> RefineNet.forward -> get_polygon_rles.  The contours equal those
    of the same boxes handed over in the duck-typed test instances, bit for bit."""
    from dance_amd import rle
    from dance_amd.fcose_outputs import FCOSEOutputs, compute_locations
    g, cfg, sd, feats, image_idx, sizes = dance_case_inputs("smooth_batched_box_v196")
    cfg.vis_path = True                                   # the batched branch of forward()
    net = _module(cfg, sd, device)
    b, hf, wf = feats.size(0), feats.size(2), feats.size(3)
    gen = torch.Generator().manual_seed(21)
    shapes, strides = [(hf // 2, wf // 2), (hf // 4, wf // 4)], [8, 16]
    logits = [(torch.randn(b, 3, h, w, generator=gen) * 1.5 - 3.0).to(device) for h, w in shapes]
    ctr = [torch.randn(b, 1, h, w, generator=gen).to(device) for h, w in shapes]
    reg = [(torch.rand(b, 4, h, w, generator=gen) * 3 + 1).to(device) for h, w in shapes]
    ext = [(torch.rand(b, 4, h, w, generator=gen) - 0.5).to(device) for h, w in shapes]

    class Images:
        image_sizes = list(sizes)

        def __len__(self):
            return b

    out = FCOSEOutputs(Images(), compute_locations(logits, strides), logits, reg, ext, ctr, 0.25, 2.0, None, None,
                       True, None, strides, 1.5, 3, 0.05, 1000, 0.6, 6, True)
    insts = out.predict_proposals()
    assert [len(i) for i in insts] == [6] * b
    plain = [FakeInstances(sizes[i], pred_boxes=FakeBoxes(insts[i].pred_boxes.tensor.clone())) for i in range(b)]
    with torch.no_grad():
        res, _ = net(feats.to(device), insts, None)
        want, _ = net(feats.to(device), plain, None)
    for r, w, size in zip(res, want, sizes):
        assert r.pred_polys.tensor.shape == (6, 196, 2)
        assert torch.equal(r.pred_polys.tensor, w.pred_polys.tensor)
        assert r.has("ext_points") and r.has("scores")
        polys = r.pred_polys.tensor.clone()
        ok = torch.isfinite(polys).all(-1).all(-1)
        masks = rle.get_polygon_rles(polys[ok].reshape(int(ok.sum()), -1), tuple(size))
        assert len(masks) == int(ok.sum()) and all(m["size"] == list(size) for m in masks)
