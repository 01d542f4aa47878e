"""CPU: the oracle (C restatement + torch restatement) against the golden vectors that
oracle/gen_golden.py produced with the reference's own code."""
import pytest
import torch

from _util import CHAIN_TOL_WHITE, dance_state_dict, dsnake_state_dict, golden, rel_err
from dance_amd import synthetic
from oracle import extreme_utils_ref as cref
from oracle import snake_ref

RESAMPLER_CASES = (("box", (128, 196, 256)), ("box_int", (128, 196)), ("quad", (40,)),
                   ("octagon", (128, 196)), ("ngon12", (128, 40)), ("ngon7", (128,)))


def test_c_resampler_bit_exact():
    g = golden("g1_resampler")
    for name, ps in RESAMPLER_CASES:
        for p in ps:
            pts, start, status = cref.uniform_upsample_fused(g[name + "_poly"], p)
            assert int(status.sum()) == 0
            assert torch.equal(start, g["%s_p%d_start" % (name, p)]), (name, p)
            assert torch.equal(pts, g["%s_p%d_points" % (name, p)]), (name, p)


def test_torch_resampler_restatement_bit_exact():
    g = golden("g1_resampler")
    for name, ps in RESAMPLER_CASES:
        for p in ps:
            pts, start = snake_ref.uniform_upsample(g[name + "_poly"][None].clone(), p)
            assert torch.equal(start, g["%s_p%d_start" % (name, p)])
            assert torch.equal(pts, g["%s_p%d_points" % (name, p)])
    assert torch.equal(snake_ref.octagon_polygon(g["octagon_ext"]), g["octagon_poly"])


def test_gather_restatement():
    g = golden("g2_gather")
    f, loc, idx = g["features"], g["locations"], g["image_idx"]
    assert rel_err(snake_ref.get_locations_feature(f, loc, idx, "reflection"), g["reflection_batched"]) < 1e-6
    assert rel_err(snake_ref.get_locations_feature(f[:1], loc, None, "reflection"), g["reflection_single"]) < 1e-6
    assert rel_err(snake_ref.get_locations_feature(f, loc, idx, "zeros"), g["zeros_batched"]) < 1e-6
    for i in range(loc.size(0)):
        for mode, key in (("reflection", "reflection_batched"), ("zeros", "zeros_batched")):
            cf = snake_ref.bilinear_sample_closed_form(f[int(idx[i])], loc[i], mode)
            assert rel_err(cf, g[key][i]) < 1e-5


def _block_sd(cin, seed):
    from dance_amd import modules
    blk = modules.SnakeBlock(cin, 128)
    sd = {k: v.clone() for k, v in blk.state_dict().items()}
    return sd, synthetic.fill_state_dict_(sd, seed)


def test_snake_block_restatement():
    g = golden("g3_snake_block")
    for tag, cin, d in (("c128_d1_v128", 128, 1), ("c128_d2_v196", 128, 2), ("c128_d4_v40", 128, 4),
                        ("c258_d1_v128", 258, 1), ("c128_d4_v256", 128, 4)):
        sd, ck = _block_sd(cin, int(g[tag + "_seed"]))
        assert abs(ck - float(g[tag + "_checksum"])) < 1e-6 * ck
        y = snake_ref.snake_block(g[tag + "_x"], sd, "", d)
        assert rel_err(y, g[tag + "_y"]) < 1e-5


def test_snake_net_restatement():
    from dance_amd import modules
    g = golden("g4_snake_net")
    for tag in ("v128", "v196", "v40"):
        net = modules._SnakeNet(128, 258, (1, 1, 1, 2, 2, 4, 4))
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        ck = synthetic.fill_state_dict_(sd, int(g[tag + "_seed"]))
        assert abs(ck - float(g[tag + "_checksum"])) < 1e-6 * ck
        y, state = snake_ref.snake_net(g[tag + "_x"], sd, return_states=True)
        assert rel_err(y, g[tag + "_y"]) < 1e-5
        assert rel_err(state.pow(2).sum(dim=2).sqrt(), g[tag + "_state_l2"]) < 1e-5


DANCE_CASES = ("single_box_v196", "batched_box_v128", "single_octagon_v128", "plain_demean_v128",
               "smooth_single_box_v128", "smooth_single_octagon_v128", "smooth_batched_box_v196")


def dance_case_inputs(tag):
    g = golden("g5_dance_" + tag)
    att, ind = bool(int(g["attention"])), bool(int(g["individual_scale"]))
    over = {} if att else dict(attention=False, individual_scale=False, de_loc_type="demean")
    cfg, sd, ck = dance_state_dict(int(g["weight_seed"]), int(g["num_sampling"]), **over)
    assert abs(ck - float(g["checksum"])) < 1e-6 * ck, "weights differ from the fixture's"
    hf, wf, batch = int(g["hf"]), int(g["wf"]), int(g["batch"])
    make = synthetic.make_smooth_features if int(g["smooth"]) else synthetic.make_features
    feats = make(batch, 256, hf, wf, int(g["feat_seed"]), attention=att)
    image_idx = g["image_idx"] if g["image_idx"].numel() else None
    sizes = (hf * 4, wf * 4) if image_idx is None else [(hf * 4, wf * 4)] * batch
    return g, cfg, sd, feats, image_idx, sizes


def test_dance_chain_restatement():
    for tag in DANCE_CASES:
        g, cfg, sd, feats, image_idx, sizes = dance_case_inputs(tag)
        kw = dict(attention=cfg.attention, individual_scale=cfg.individual_scale,
                  de_loc_type=cfg.de_loc_type)
        # every iteration on IDENTICAL inputs (the fixture's previous contours): 1e-5
        for i in range(3):
            prev = g["init"] if i == 0 else g["pred%d" % (i - 1)]
            one = snake_ref.dance_evolve(snake_ref.sub_state_dict(sd, "deformer%d." % i), feats,
                                         prev.clone(), image_idx, sizes, g["whs"],
                                         att=(cfg.attention and i > 0), **kw)
            assert rel_err(one, g["pred%d" % i]) < 1e-5, (tag, i)
        # the whole chain: white-noise fixtures are chaotic (the host CPU's own summation
        # order moves iteration 2 by ~1e-4, _util.CHAIN_TOL_WHITE), band-limited ones are not
        preds = snake_ref.dance_deform(sd, feats, g["init"], image_idx, sizes, g["whs"], **kw)
        tol = (1e-5, 1e-5, 1e-5) if int(g["smooth"]) else CHAIN_TOL_WHITE
        for i in range(3):
            assert rel_err(preds[i], g["pred%d" % i]) < tol[i], (tag, i)


def test_dsnake_restatement():
    g = golden("g6_dsnake")
    cfg, sd, ck = dsnake_state_dict(int(g["weight_seed"]))
    assert abs(ck - float(g["checksum"])) < 1e-6 * ck
    feats = synthetic.make_smooth_features(2, 256, int(g["hf"]), int(g["wf"]), int(g["feat_seed"]), attention=False)
    out = snake_ref.dsnake_deform(sd, feats, [g["boxes"][0], g["boxes"][1]])
    assert torch.equal(out["init"], g["init"])
    assert rel_err(out["exts"], g["exts"]) < 1e-5
    assert rel_err(out["octagons"], g["octagons"]) < 1e-5
    for i in range(3):
        assert rel_err(out["preds"][i], g["pred%d" % i]) < 1e-5


def test_attender_restatement():
    """oracle/snake_ref.attender against the output of the reference's own
    EdgeSnakeFPNHead.attender (edge_det.py:66-100; fixture written by oracle/gen_golden.py)."""
    from dance_amd import modules
    g = golden("g9_attender")
    sd = {k: v.clone() for k, v in modules.Attender().state_dict().items()}
    ck = synthetic.fill_state_dict_(sd, int(g["weight_seed"]))
    assert abs(ck - float(g["weight_checksum"])) < 1e-6 * ck
    seed = int(g["input_seed"])
    a = synthetic.make_smooth_features(2, 1, 50, 76, seed, attention=False, cell=4)
    b = synthetic.make_smooth_features(1, 1, 37, 53, seed + 1, attention=False, cell=2)
    for i, x in enumerate([1 - torch.sigmoid(3.0 * a), 1 - torch.sigmoid(3.0 * b)]):
        assert float((snake_ref.attender(sd, x, prefix="") - g["out%d" % i]).abs().max()) < 1e-6


def test_edge_head_restatement():
    """oracle/snake_ref.edge_head_forward against the reference's own EdgeSnakeFPNHead.forward
    (edge_det.py:187-395, eval, one image; fixture written by oracle/gen_golden.py)."""
    from _util import coord_rel_err, edge_head_cfg, edge_head_inputs
    from dance_amd import modules
    g = golden("g10_edge_head")
    cfg, shapes = edge_head_cfg()
    sd = {k: v.clone() for k, v in modules.EdgeSnakeFPNHead(cfg, shapes).state_dict().items()}
    ck = synthetic.fill_state_dict_(sd, int(g["weight_seed"]))
    assert abs(ck - float(g["weight_checksum"])) < 1e-6 * ck, "state-dict keys/order differ from the reference's"
    feats, boxes = edge_head_inputs(int(g["input_seed"]))
    att, preds = snake_ref.edge_head_forward(sd, feats, boxes, (192, 256), 128)
    assert float((att - g["att_map"]).abs().max()) < 1e-5
    assert coord_rel_err(preds[-1], g["pred_polys"]) < 1e-4


def test_upsample2x_restatement():
    g = golden("g7_upsample2x")
    for i in range(3):
        assert torch.equal(snake_ref.contour_upsample_2x(g["loc%d" % i]), g["out%d" % i])


def test_extreme_utils_known_answers():
    # 5x5 map, one box, hand-computed: votes all point at the centre (2,2)
    h = w = 5
    hm = torch.zeros((1, 4, h, w))
    hm[0, 0, 0, 3] = 2.0      # top: best in the band rows 0..1
    hm[0, 1, 1, 0] = 3.0      # left
    hm[0, 2, 4, 1] = 5.0      # bottom
    hm[0, 3, 2, 4] = 1.0      # right: sits on the initial mid-point -> not a strict improvement
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    vote = torch.stack([2.0 - xs, 2.0 - ys], dim=-1)[None].contiguous()
    bbox = torch.tensor([[[0, 0, 4, 4]]], dtype=torch.int64)
    radius = torch.tensor([[1]], dtype=torch.int64)
    ct = torch.tensor([[[2.0, 2.0]]])
    out = cref.collect_extreme_point(hm, bbox, radius, vote, ct)
    assert out.tolist() == [[[[3, 0], [0, 1], [1, 4], [4, 2]]]]
    # votes pointing away: nothing qualifies, mid-points come back
    out = cref.collect_extreme_point(hm, bbox, radius, -vote, ct)
    assert out.tolist() == [[[[2, 0], [0, 2], [2, 4], [4, 2]]]]
    # roll
    a = torch.arange(12, dtype=torch.float32).view(1, 6, 2)
    r = cref.roll_array(a, torch.tensor([-2]))
    assert torch.equal(r[0, :, 0], torch.tensor([4., 6., 8., 10., 0., 2.]))
    assert torch.equal(cref.roll_array(a, torch.tensor([8])), torch.roll(a, 2, 1))
    # edge_num: surplus walks the sorted edges; deficit goes to the largest
    en = torch.tensor([[[5, 1, 3, 1]]])
    cref.calculate_edge_num(en, en.sum(2), torch.argsort(en, dim=2, descending=True, stable=True), 7)
    assert en.tolist() == [[[2, 1, 3, 1]]]
    en = torch.tensor([[[2, 1, 2, 1]]])
    cref.calculate_edge_num(en, en.sum(2), torch.argsort(en, dim=2, descending=True, stable=True), 3)
    assert en.tolist() == [[[1, 1, 1, 1]]]      # cannot go below one point per edge
    en = torch.tensor([[[2, 1, 2, 1]]])
    cref.calculate_edge_num(en, en.sum(2), torch.argsort(en, dim=2, descending=True, stable=True), 9)
    assert en.tolist() == [[[5, 1, 2, 1]]]
    w_, ind = cref.calculate_wnp(torch.tensor([[[2, 1]]]), torch.tensor([[[0, 2]]]), 4)
    assert w_.flatten().tolist() == [0.0, 0.5, 0.0, 0.0]
    assert ind[0, 0].tolist() == [[0, 1], [0, 1], [1, 0], [0, 0]]


def test_nms_oracle_known_answers():
    """C restatement of nms.cu on hand-computed cases ("+1" IoU, strict >, result ascending)."""
    sc = torch.tensor([0.1, 0.9, 0.5, 0.3, 0.7])
    same = torch.tensor([[10., 10., 50., 50.]]).repeat(5, 1)
    assert cref.nms(same, sc, 0.5).tolist() == [1]
    pair = torch.tensor([[0., 0., 9., 9.], [5., 0., 14., 9.]])
    assert cref.nms(pair, sc[:2], 0.34).tolist() == [0, 1]       # IoU = 50 / 150
    assert cref.nms(pair, sc[:2], 0.33).tolist() == [1]
    assert cref.nms(torch.zeros((0, 4)), torch.zeros((0,)), 0.5).numel() == 0


def test_batched_nms_oracle_known_answers():
    """C restatement of torchvision's NMS (no "+1", strict >) behind detectron2.layers.batched_nms,
    the call of the reference's ml_nms (core/layers/ml_nms.py:25).  PARITY UNPINNED: hand-computed."""
    pair = torch.tensor([[0., 0., 10., 10.], [5., 0., 15., 10.]])               # IoU = 50 / 150
    sc = torch.tensor([0.9, 0.8])
    same = torch.zeros(2, dtype=torch.int64)
    assert cref.batched_nms(pair, sc, same, 0.34).tolist() == [0, 1]
    assert cref.batched_nms(pair, sc, same, 0.33).tolist() == [0]
    assert cref.batched_nms(pair, sc, torch.tensor([0, 1]), 0.33).tolist() == [0, 1]   # other class: kept
    assert cref.batched_nms(pair, torch.tensor([0.1, 0.8]), same, 0.33).tolist() == [1]  # score order
    assert cref.batched_nms(torch.zeros((0, 4)), torch.zeros((0,)), torch.zeros((0,), dtype=torch.int64), 0.5).numel() == 0


FCOSE_LEVELS = (((12, 16), 8), ((6, 8), 16), ((3, 4), 32))
FCOSE_CASES = (("plain", 0.05, True, 1000, 0.6, 100), ("topk", 0.05, False, 20, 0.6, 10),
               ("sparse", 0.45, True, 1000, 0.5, 100))


@pytest.mark.parametrize("case", FCOSE_CASES, ids=[c[0] for c in FCOSE_CASES])
def test_fcose_restatement(case):
    """oracle/fcose_ref.py against the reference's own FCOSEOutputs.predict_proposals
    (fcose_outputs.py:419-542; fixture g11, inputs stored in it): classes bit-exact, boxes /
    locations / extreme points bit-exact, scores 1e-6."""
    from oracle import fcose_ref
    tag, th, with_ctr, pre, nms_th, post = case
    g = golden("g11_fcose")
    t = lambda k: g[k]
    strides = [s for _, s in FCOSE_LEVELS]
    locations = fcose_ref.compute_locations([hw for hw, _ in FCOSE_LEVELS], strides)
    got = fcose_ref.predict_proposals(locations, [t("%s_logits%d" % (tag, l)) for l in range(3)],
                                      [t("%s_reg%d" % (tag, l)) for l in range(3)],
                                      [t("%s_ext%d" % (tag, l)) for l in range(3)],
                                      [t("%s_ctr%d" % (tag, l)) for l in range(3)], strides, th, with_ctr, pre,
                                      nms_th, post)
    for i, d in enumerate(got):
        for k in ("classes", "boxes", "locations", "ext"):
            assert torch.equal(d[k], t("%s_im%d_%s" % (tag, i, k))), (tag, i, k)
        w = t("%s_im%d_scores" % (tag, i))
        assert float(((d["scores"] - w).abs() / w).max()) <= 1e-6
