"""ORACLE -- TEST INFRASTRUCTURE ONLY; runs in the BUILD CONTAINER only.

Generates tests/golden/*.npz by running the REFERENCE'S OWN Python
(/root/reference/core/modeling/edge_snake/snake_head.py, .../dsnake_head.py; loaded by
oracle/ref_loader.py) on seeded inputs.  A fixture is data only: inputs (or the seed that
regenerates them through dance_amd/synthetic.py), the configuration, and the reference's
outputs.  Weights and feature maps are never stored -- they are regenerated from the seed
and guarded by a checksum kept in the fixture.

    python -m oracle.gen_golden            # from the repo root, in the build container

The script also cross-checks oracle/snake_ref.py and the C resampler against the
reference on every case and fails if they disagree (ints bit-exact, floats 1e-5).
"""
import os
import sys

import numpy as np
from types import SimpleNamespace as NS
import torch

from dance_amd import synthetic
from oracle import extreme_utils_ref, ref_loader, snake_ref, targets_ref

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
WEIGHT_SEED = 1234


def _save(name, **arrays):
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print("wrote %-28s %7.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024))


def _close(a, b, tol=1e-5, what=""):
    # NaNs are part of the reference's behaviour (derange divides by max-min with no
    # epsilon, snake_head.py:1588-1593): they must sit at the same places.
    assert torch.equal(torch.isnan(a), torch.isnan(b)), what + ": NaN pattern differs"
    a, b = torch.nan_to_num(a, nan=0.0), torch.nan_to_num(b, nan=0.0)
    err = float(((a - b).abs() / b.abs().clamp_min(1.0)).max()) if a.numel() else 0.0
    assert err <= tol, "%s: oracle vs reference rel err %.3g" % (what, err)
    return err


# ------------------------------------------------------------------------------- G1
def resampler_polygons():
    g = torch.Generator().manual_seed(7)
    cases = {}
    # random float boxes (the DANCE initial contour, snake_head.py:911-924)
    boxes = synthetic.make_boxes(40, 800, 1216, seed=11)
    cases["box"] = snake_ref.box_polygon(boxes)
    # integer boxes: many exact .5 ties in len*p/sum, degenerate and sliver boxes
    ib = torch.tensor([[0, 0, 1, 127], [10, 10, 74, 74], [5, 5, 5, 5], [3, 9, 3, 200],
                       [0, 0, 1215, 799], [100, 50, 103, 51], [7, 7, 8, 8], [20, 30, 220, 31],
                       [1, 2, 4, 130], [50, 60, 51, 60.001]], dtype=torch.float32)
    cases["box_int"] = snake_ref.box_polygon(ib)
    # quadrangles (Dsnake initial contour, dsnake_head.py:501-517)
    cases["quad"] = snake_ref.quadrangle(boxes[:20])
    # octagons from extreme points (snake_head.py:796-832): random + collapsed ones
    cx, cy = (boxes[:30, 0] + boxes[:30, 2]) / 2, (boxes[:30, 1] + boxes[:30, 3]) / 2
    w, h = boxes[:30, 2] - boxes[:30, 0], boxes[:30, 3] - boxes[:30, 1]
    j = (torch.rand((30, 4), generator=g) - 0.5)
    ex = torch.stack([cx + j[:, 0] * w, boxes[:30, 1], boxes[:30, 0], cy + j[:, 1] * h,
                      cx + j[:, 2] * w, boxes[:30, 3], boxes[:30, 2], cy + j[:, 3] * h],
                     dim=1).view(-1, 4, 2)
    ex[3] = ex[3, 0]                  # all four extreme points identical
    ex[4, :, 1] = ex[4, 0, 1]         # zero height
    ex[5, 0, 0] = ex[5, 1, 0]         # top point in the left corner
    cases["octagon"] = snake_ref.octagon_polygon(ex)
    cases["octagon_ext"] = ex
    # random 12-gons and 7-gons far from any structure
    cases["ngon12"] = torch.rand((25, 12, 2), generator=g) * 300
    cases["ngon7"] = torch.rand((10, 7, 2), generator=g) * 50
    return cases


def gen_resampler(sh):
    polys = resampler_polygons()
    out = {}
    for name, p_nums in (("box", (128, 196, 256)), ("box_int", (128, 196)), ("quad", (40,)),
                         ("octagon", (128, 196)), ("ngon12", (128, 40)), ("ngon7", (128,))):
        poly = polys[name].clone()
        out[name + "_poly"] = poly
        for p in p_nums:
            pts, start = sh.RefineNet.uniform_upsample(poly.clone()[None], p)
            out["%s_p%d_points" % (name, p)] = pts
            out["%s_p%d_start" % (name, p)] = start
            o_pts, o_start = snake_ref.uniform_upsample(poly.clone()[None], p)
            assert torch.equal(o_start, start), (name, p)
            assert torch.equal(o_pts, pts), (name, p)
            c_pts, c_start, c_status = extreme_utils_ref.uniform_upsample_fused(poly, p)
            assert int(c_status.sum()) == 0
            assert torch.equal(c_start, start), ("C resampler start", name, p)
            assert torch.equal(c_pts, pts), ("C resampler points", name, p,
                                            float((c_pts - pts).abs().max()))
    # get_octagon straight from extreme points (reference method, needs an instance)
    ex = polys["octagon_ext"]
    net = sh.RefineNet(ref_loader.dance_cfg(num_convs=0))
    pts, start = net.get_octagon(ex.clone(), 128)
    assert torch.equal(pts, out["octagon_p128_points"])
    out["octagon_ext"] = ex
    # NumPy twin of the resampler (snake_head.py:661-725): non-authoritative cross-check
    agree = 0
    for i in range(polys["ngon12"].size(0)):
        tw = sh.RefineNet.uniform_sample(polys["ngon12"][i].numpy(), 128)
        agree += int(np.abs(tw - out["ngon12_p128_points"][i].numpy()).max() < 1e-3)
    print("numpy twin uniform_sample agrees on %d/%d random 12-gons" % (agree, polys["ngon12"].size(0)))
    _save("g1_resampler", **out)


# ------------------------------------------------------------------------------- G2
def gen_gather(sh, dh):
    g = torch.Generator().manual_seed(21)
    feats = torch.randn((2, 5, 7, 9), generator=g)
    H, W = 28, 36
    n, v = 6, 16
    loc = torch.rand((n, v, 2), generator=g) * torch.tensor([W, H]) * 1.6 - torch.tensor([W, H]) * 0.3
    # exact pixel centres, borders and far outside points
    loc[0, :8] = torch.tensor([[2.0, 2.0], [0.0, 0.0], [W - 1.0, H - 1.0], [W, H], [-0.5, 5.0],
                               [6.0, 6.0], [-40.0, 90.0], [35.999, 27.999]])
    idx = torch.tensor([0, 1, 1, 0, 1, 0], dtype=torch.float32)
    net = sh.RefineNet(ref_loader.dance_cfg(num_convs=0, convs_dim=4))
    dnet = dh.DsnakeHead(ref_loader.dsnake_cfg(num_convs=0, convs_dim=4))
    refl_b = net.get_locations_feature(feats, loc, idx)
    refl_1 = net.get_locations_feature(feats[:1], loc, None)
    zero_b = dnet.get_locations_feature(feats, loc, idx)
    for a, b, mode, fi, ii in ((refl_b, None, "reflection", feats, idx),
                               (refl_1, None, "reflection", feats[:1], None),
                               (zero_b, None, "zeros", feats, idx)):
        o = snake_ref.get_locations_feature(fi, loc, ii, mode)
        assert torch.equal(o, a)
    # closed form vs grid_sample
    for i in range(n):
        cf = snake_ref.bilinear_sample_closed_form(feats[int(idx[i])], loc[i], "reflection")
        _close(cf, refl_b[i], 1e-5, "closed-form reflection")
        cf = snake_ref.bilinear_sample_closed_form(feats[int(idx[i])], loc[i], "zeros")
        _close(cf, zero_b[i], 1e-5, "closed-form zeros")
    _save("g2_gather", features=feats, locations=loc, image_idx=idx,
          reflection_batched=refl_b, reflection_single=refl_1, zeros_batched=zero_b)


# --------------------------------------------------------------------------- G3 / G4
def gen_blocks(sh):
    out = {}
    g = torch.Generator().manual_seed(33)
    for tag, cin, d, v in (("c128_d1_v128", 128, 1, 128), ("c128_d2_v196", 128, 2, 196),
                           ("c128_d4_v40", 128, 4, 40), ("c258_d1_v128", 258, 1, 128),
                           ("c128_d4_v256", 128, 4, 256)):
        blk = sh.SnakeBlock(cin, 128, n_adj=4, dilation=d).eval()
        sd = blk.state_dict()
        ck = synthetic.fill_state_dict_(sd, WEIGHT_SEED + d + v)
        x = torch.randn((2, cin, v), generator=g)
        with torch.no_grad():
            y = blk(x)
            o = snake_ref.snake_block(x, sd, "", d)
        _close(o, y, 1e-6, "snake_block " + tag)
        out[tag + "_x"] = x
        out[tag + "_y"] = y
        out[tag + "_checksum"] = ck
        out[tag + "_seed"] = WEIGHT_SEED + d + v
    _save("g3_snake_block", **out)


def gen_snakenet(sh):
    out = {}
    g = torch.Generator().manual_seed(44)
    cfg = ref_loader.dance_cfg()
    for tag, v in (("v128", 128), ("v196", 196), ("v40", 40)):
        net = sh._SnakeNet(cfg, 0).eval()
        sd = net.state_dict()
        ck = synthetic.fill_state_dict_(sd, WEIGHT_SEED + v)
        x = torch.randn((3, 258, v), generator=g)
        x[:, 256:] = torch.rand((3, 2, v), generator=g)
        with torch.no_grad():
            y = net(x)
            o, state = snake_ref.snake_net(x, sd, return_states=True)
        _close(o, y, 1e-5, "snake_net " + tag)
        out[tag + "_x"] = x
        out[tag + "_y"] = y
        out[tag + "_state_l2"] = state.pow(2).sum(dim=2).sqrt()     # (3,1024) per-channel norms
        out[tag + "_checksum"] = ck
        out[tag + "_seed"] = WEIGHT_SEED + v
    _save("g4_snake_net", **out)


# ------------------------------------------------------------------------------- G5
def _dance_case(sh, tag, *, num_sampling, initial, batch, n_per_im, attention=True,
                individual_scale=True, de_loc_type="derange", feat_seed, box_seed, hf=50, wf=76,
                smooth=False):
    cfg = ref_loader.dance_cfg(num_sampling=num_sampling, initial=initial, num_convs=0,
                               attention=attention, individual_scale=individual_scale,
                               de_loc_type=de_loc_type)
    net = sh.RefineNet(cfg).eval()
    sd = net.state_dict()
    ck = synthetic.fill_state_dict_(sd, WEIGHT_SEED)
    H, W = hf * 4, wf * 4
    make = synthetic.make_smooth_features if smooth else synthetic.make_features
    feats = make(batch, 256, hf, wf, feat_seed, attention=attention)
    boxes, insts = [], []
    for i in range(batch):
        b = synthetic.make_boxes(n_per_im, H, W, box_seed + i, r_min=8.0, r_max=60.0)
        if i == 0:
            b[0] = torch.tensor([W - 70.0, H - 50.0, W - 1.0, H - 1.0])   # hugs the corner
        boxes.append(b)
        fields = dict(pred_boxes=ref_loader.FakeBoxes(b.clone()))
        if initial == "octagon":
            g = torch.Generator().manual_seed(box_seed + 100 + i)
            j = torch.rand((n_per_im, 4), generator=g)
            w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
            ex = torch.stack([b[:, 0] + j[:, 0] * w, b[:, 1], b[:, 0], b[:, 1] + j[:, 1] * h,
                              b[:, 0] + j[:, 2] * w, b[:, 3], b[:, 2], b[:, 1] + j[:, 3] * h],
                             dim=1).view(-1, 4, 2)
            ex[1, 0, 0] -= 30.0      # outside its box: fit_to_box must pull it back
            fields["ext_points"] = _Ext(ex.clone())
            fields["_ext_in"] = ex
        insts.append(ref_loader.FakeInstances((H, W), **fields))
    with torch.no_grad():
        if batch == 1:
            # single-image path, snake_head.py:1763-1869
            if initial == "octagon":
                locations, image_idx = net.sample_octagons_fast(insts)
                image_idx = None
            else:
                locations, image_idx = net.single_sample_bboxes_fast(insts)
            image_sizes = insts[0].image_size
            insts[0].pred_boxes.clip(image_sizes)
            bb = insts[0].pred_boxes.tensor
        else:
            # batched path, snake_head.py:1895-1941
            if initial == "octagon":
                locations, image_idx = net.sample_octagons_fast(insts)
            else:
                locations, image_idx = net.sample_bboxes_fast(insts)
            image_sizes = [x.image_size for x in insts]
            bb = torch.cat([x.pred_boxes.tensor for x in insts], dim=0)
        whs = torch.stack([bb[:, 2] - bb[:, 0], bb[:, 3] - bb[:, 1]], dim=1)
        preds, cur = [], locations
        for i in range(3):
            cur, _ = net.evolve(getattr(net, "deformer%d" % i), feats, cur, image_idx,
                                image_sizes, whs, att=(attention and i > 0))
            preds.append(cur)
        o_preds = snake_ref.dance_deform(sd, feats, locations, image_idx, image_sizes, whs,
                                         attention=attention, individual_scale=individual_scale,
                                         de_loc_type=de_loc_type)
        if batch == 1 and initial == "box":
            # whole reference forward (eval): must reproduce the last iteration
            insts2 = [ref_loader.FakeInstances((H, W), pred_boxes=ref_loader.FakeBoxes(boxes[0].clone()))]
            res, _ = net(feats, insts2, None)
            assert torch.equal(torch.isnan(res[0].pred_polys.tensor), torch.isnan(preds[-1]))
            assert torch.equal(torch.nan_to_num(res[0].pred_polys.tensor), torch.nan_to_num(preds[-1]))
    for a, b in zip(o_preds, preds):
        _close(a, b, 1e-5, "dance_deform " + tag)
    out = dict(boxes=torch.stack(boxes), init=locations, whs=whs,
               image_idx=(torch.zeros(0) if image_idx is None else image_idx),
               pred0=preds[0], pred1=preds[1], pred2=preds[2], checksum=ck,
               feat_seed=feat_seed, weight_seed=WEIGHT_SEED, hf=hf, wf=wf, batch=batch,
               num_sampling=num_sampling, attention=int(attention),
               individual_scale=int(individual_scale), smooth=int(smooth))
    if initial == "octagon":
        out["ext_points"] = torch.stack([x._ext_in for x in insts])
    _save("g5_dance_" + tag, **out)
    return out


class _Ext:
    """ExtremePoints duck type with the reference's fit_to_box (pointset.py:132-142),
    restated in oracle.snake_ref.fit_to_box."""

    def __init__(self, t):
        self.tensor = t

    def fit_to_box(self):
        snake_ref.fit_to_box(self.tensor)


def gen_dance(sh):
    _dance_case(sh, "single_box_v196", num_sampling=196, initial="box", batch=1, n_per_im=5,
                feat_seed=501, box_seed=601)
    _dance_case(sh, "batched_box_v128", num_sampling=128, initial="box", batch=2, n_per_im=3,
                feat_seed=502, box_seed=602)
    _dance_case(sh, "single_octagon_v128", num_sampling=128, initial="octagon", batch=1,
                n_per_im=4, feat_seed=503, box_seed=603)
    _dance_case(sh, "plain_demean_v128", num_sampling=128, initial="box", batch=2, n_per_im=2,
                attention=False, individual_scale=False, de_loc_type="demean",
                feat_seed=504, box_seed=604)
    # band-limited features: the 3-iteration chain is well conditioned, whole-chain parity
    _dance_case(sh, "smooth_single_box_v128", num_sampling=128, initial="box", batch=1, n_per_im=6,
                feat_seed=505, box_seed=605, smooth=True)
    _dance_case(sh, "smooth_single_octagon_v128", num_sampling=128, initial="octagon", batch=1,
                n_per_im=5, feat_seed=507, box_seed=607, smooth=True)
    _dance_case(sh, "smooth_batched_box_v196", num_sampling=196, initial="box", batch=3, n_per_im=3,
                feat_seed=506, box_seed=606, smooth=True)


# ------------------------------------------------------------------------------- G6
def gen_dsnake(dh):
    cfg = ref_loader.dsnake_cfg(num_convs=0)
    net = dh.DsnakeHead(cfg).eval()
    sd = net.state_dict()
    ck = synthetic.fill_state_dict_(sd, WEIGHT_SEED)
    hf, wf, batch, n_per_im = 50, 76, 2, 3
    H, W = hf * 4, wf * 4
    feats = synthetic.make_smooth_features(batch, 256, hf, wf, 701, attention=False)
    boxes = [synthetic.make_boxes(n_per_im, H, W, 801 + i, r_min=8.0, r_max=60.0) for i in range(batch)]
    insts = [ref_loader.FakeInstances((H, W), pred_boxes=ref_loader.FakeBoxes(b.clone())) for b in boxes]
    with torch.no_grad():
        res, _ = net(feats, insts, None)
        o = snake_ref.dsnake_deform(sd, feats, boxes)
    final = torch.cat([r.pred_polys.tensor for r in res])
    exts = torch.cat([r.ext_points.tensor for r in res])
    _close(o["preds"][-1], final, 1e-5, "dsnake final")
    _close(o["exts"], exts, 1e-5, "dsnake exts")
    # per-stage values through the reference's own methods (dsnake_head.py:783-830)
    with torch.no_grad():
        init, image_idx = net.sample_quadrangles_fast(insts)
        pe = net.init(net.init_snake, feats, init, image_idx)
        octs = []
        for i in range(batch):
            e = pe[image_idx == i]
            e[..., 0] = torch.clamp(e[..., 0], min=0, max=W - 1)
            e[..., 1] = torch.clamp(e[..., 1], min=0, max=H - 1)
            octs.append(net.get_octagon(e, 128))
        octs = torch.cat(octs)
        preds, cur = [], octs
        for i in range(3):
            cur = net.evolve(getattr(net, "deformer%d" % i), feats, cur, image_idx)
            preds.append(cur)
    assert torch.equal(preds[-1], final)
    assert torch.equal(init, o["init"])
    _close(o["octagons"], octs, 1e-5, "dsnake octagons")
    for a, b in zip(o["preds"], preds):
        _close(a, b, 1e-5, "dsnake preds")
    _save("g6_dsnake", boxes=torch.stack(boxes), init=init, image_idx=image_idx, exts_raw=pe,
          exts=exts, octagons=octs, pred0=preds[0], pred1=preds[1], pred2=preds[2],
          checksum=ck, feat_seed=701, weight_seed=WEIGHT_SEED, hf=hf, wf=wf, smooth=1)


def gen_upsample2x(sh):
    """`RefineNet.contour_upsample_2x` (snake_head.py:1752-1761) on seeded contours."""
    g = torch.Generator().manual_seed(77)
    out = {}
    for i, (n, v) in enumerate([(1, 3), (5, 128), (3, 37)]):
        loc = torch.rand((n, v, 2), generator=g) * 1200.0 - 100.0
        ref = sh.RefineNet.contour_upsample_2x(loc)
        assert torch.equal(snake_ref.contour_upsample_2x(loc), ref)
        out["loc%d" % i], out["out%d" % i] = loc, ref
    _save("g7_upsample2x", **out)


def gen_bottom_out(sh):
    """The reference's own `RefineNet.bottom_out` ModuleList (snake_head.py:562-615) as
    `single_test` runs it (1816-1820) on a small seeded map; weights from the seed."""
    cfg = ref_loader.dance_cfg(num_sampling=128, num_convs=2)
    ref = sh.RefineNet(cfg).eval()
    sd = ref.state_dict()
    ck = synthetic.fill_state_dict_(sd, WEIGHT_SEED)
    ref.load_state_dict(sd)
    feats = synthetic.make_smooth_features(2, 256, 9, 70, 31, attention=False, cell=4)
    with torch.no_grad():
        x = feats
        for i in range(ref.num_convs):
            x = ref.bottom_out[i](x)
    mine = snake_ref.bottom_out({k: v for k, v in sd.items()}, feats)
    _close(mine, x, 1e-5, "bottom_out")
    _save("g8_bottom_out", weight_seed=WEIGHT_SEED, weight_checksum=ck, feat_seed=31, out=x)


def attender_inputs(seed):
    """x = 1 - sigmoid(band-limited logits), the form the attender sees (edge_det.py:199-204):
    a full-width map and a small odd-sized one."""
    a = synthetic.make_smooth_features(2, 1, 50, 76, seed, attention=False, cell=4)
    b = synthetic.make_smooth_features(1, 1, 37, 53, seed + 1, attention=False, cell=2)
    return [1 - torch.sigmoid(3.0 * a), 1 - torch.sigmoid(3.0 * b)]


def gen_attender():
    """The reference's own EdgeSnakeFPNHead.attender (edge_det.py:66-100), built by the
    reference's constructor, with seeded weights."""
    from types import SimpleNamespace as NS
    ed = ref_loader.load_edge_head()
    shapes = {k: NS(stride=s, channels=256) for k, s in (("p2", 4), ("p3", 8), ("p4", 16), ("p5", 32))}
    head = ed.EdgeSnakeFPNHead(ref_loader.edge_head_cfg(num_convs=0), shapes).eval()
    sd = {k: v for k, v in head.attender.state_dict().items()}
    ck = synthetic.fill_state_dict_(sd, WEIGHT_SEED)
    head.attender.load_state_dict(sd)
    out = {"weight_seed": WEIGHT_SEED, "weight_checksum": ck, "input_seed": 41}
    with torch.no_grad():
        for i, x in enumerate(attender_inputs(41)):
            ref = head.attender(x)
            _close(snake_ref.attender(sd, x, prefix=""), ref, 1e-6, "attender")
            out["out%d" % i] = ref
    _save("g9_attender", **out)


EDGE_HEAD_LEVELS = (("p2", 4), ("p3", 8), ("p4", 16), ("p5", 32))


def edge_head_inputs(seed, hf=48, wf=64, n=6):
    """Band-limited FPN levels p2..p5 of one (4*hf) x (4*wf) image and n detector boxes."""
    feats = {}
    for i, (name, stride) in enumerate(EDGE_HEAD_LEVELS):
        k = stride // 4
        feats[name] = synthetic.make_smooth_features(1, 256, hf // k, wf // k, seed + i, attention=False,
                                                     cell=max(2, 8 // k))
    boxes = synthetic.make_boxes(n, hf * 4, wf * 4, seed + 9, r_min=10, r_max=50)
    return feats, boxes


def gen_edge_head():
    """The reference's own EdgeSnakeFPNHead.forward (edge_det.py:187-395) in eval mode on one
    image: scale heads -> predictor -> attender -> RefineNet.single_test; shipped configuration
    (attention, strong features, box initial contours, NUM_CONVS 2), 128 vertices."""
    from types import SimpleNamespace as NS
    ed = ref_loader.load_edge_head()
    shapes = {k: NS(stride=s, channels=256) for k, s in EDGE_HEAD_LEVELS}
    cfg = ref_loader.edge_head_cfg(num_convs=2, num_sampling=128)
    head = ed.EdgeSnakeFPNHead(cfg, shapes).eval()
    sd = head.state_dict()
    ck = synthetic.fill_state_dict_(sd, WEIGHT_SEED)
    head.load_state_dict(sd)
    feats, boxes = edge_head_inputs(51)
    inst = ref_loader.FakeInstances((192, 256), pred_boxes=ref_loader.FakeBoxes(boxes.clone()))
    with torch.no_grad():
        att, _, new = head(feats, [inst], None)
    polys = new[0].pred_polys.tensor
    mine_att, mine = snake_ref.edge_head_forward({k: v for k, v in sd.items()}, feats, boxes, (192, 256), 128)
    _close(mine_att, att, 1e-5, "edge head attention map")
    _close(mine[-1], polys, 1e-4, "edge head contours")
    _save("g10_edge_head", weight_seed=WEIGHT_SEED, weight_checksum=ck, input_seed=51, att_map=att,
          pred_polys=polys)


FCOSE_LEVELS = (((12, 16), 8), ((6, 8), 16), ((3, 4), 32))
FCOSE_CASES = (   # tag, seed, thresh, with_ctr, pre_top_n, nms_thresh, post_top_n, with_ext
    ("plain", 151, 0.05, True, 1000, 0.6, 100, True),
    ("topk", 62, 0.05, False, 20, 0.6, 10, True),
    ("sparse", 63, 0.45, True, 1000, 0.5, 100, True),
)


def fcose_inputs(seed, n=2, c=4):
    g = torch.Generator().manual_seed(seed)
    logits, reg, ext, ctr = [], [], [], []
    for (h, w), _ in FCOSE_LEVELS:
        logits.append(torch.randn(n, c, h, w, generator=g) * 1.5 - 1.0)
        reg.append(torch.rand(n, 4, h, w, generator=g) * 3.5 + 0.5)
        ext.append(torch.randn(n, 4, h, w, generator=g) * 0.3)
        ctr.append(torch.randn(n, 1, h, w, generator=g))
    return logits, reg, ext, ctr


def gen_fcose():
    """The reference's own FCOSEOutputs.predict_proposals (fcose_outputs.py:419-542) on small
    seeded head outputs: three levels, two images, four classes; with and without the per-level
    top-k, the post-NMS cut and the extreme-point branch.  Inputs are stored in the fixture."""
    from types import SimpleNamespace as NS
    from oracle import fcose_ref
    fo = ref_loader.load_fcose_outputs()
    strides = [s for _, s in FCOSE_LEVELS]
    locations = fcose_ref.compute_locations([hw for hw, _ in FCOSE_LEVELS], strides)
    sizes = [(96, 128), (90, 120)]
    out = {}
    for tag, seed, th, with_ctr, pre, nms_th, post, with_ext in FCOSE_CASES:
        logits, reg, ext, ctr = fcose_inputs(seed)
        # no score within 1e-4 of the threshold: the candidate set must not depend on the last
        # bits of exp()
        for l in range(len(logits)):
            s = torch.sigmoid(logits[l])
            v = s * torch.sigmoid(ctr[l]) if with_ctr else s
            assert float((v - th).abs().min()) > 1e-4, tag
        images = type("Images", (), {"__len__": lambda self: len(sizes), "image_sizes": sizes})()
        # torch 2.10 keeps the permuted strides through sigmoid(), and the reference's
        # `candidate_inds.view(N, -1)` (fcose_outputs.py:466) then refuses an NCHW-contiguous
        # input (torch 1.4 returned a contiguous tensor there); the same values in channels-last
        # memory go through unchanged
        cl = [x.contiguous(memory_format=torch.channels_last) for x in logits]
        o = fo.FCOSEOutputs(images, locations, cl, reg, ext if with_ext else None, ctr, 0.25, 2.0, None, None,
                            True, None, strides, 1.5, 4, th, pre, nms_th, post, with_ctr)
        ref = o.predict_proposals()
        mine = fcose_ref.predict_proposals(locations, logits, reg, ext if with_ext else None, ctr, strides, th,
                                           with_ctr, pre, nms_th, post)
        for l in range(len(logits)):
            out["%s_logits%d" % (tag, l)], out["%s_reg%d" % (tag, l)] = logits[l], reg[l]
            out["%s_ext%d" % (tag, l)], out["%s_ctr%d" % (tag, l)] = ext[l], ctr[l]
        for i, (r, m) in enumerate(zip(ref, mine)):
            got = dict(boxes=r.pred_boxes.tensor, scores=r.scores, classes=r.pred_classes, locations=r.locations)
            if with_ext:
                got["ext"] = r.ext_points.tensor
            print("  fcose %-7s image %d: %d detections" % (tag, i, len(r.scores)))
            for k, v in got.items():
                assert v.shape == m[k].shape, (tag, i, k, v.shape, m[k].shape)
                if v.dtype == torch.int64:
                    assert torch.equal(v, m[k]), (tag, i, k)
                else:
                    _close(m[k], v, 1e-6, "fcose %s %s" % (tag, k))
                out["%s_im%d_%s" % (tag, i, k)] = v
    _save("g11_fcose", **out)


# ------------------------------------------------------------------------------- G12
def target_polygons():
    """Clockwise (image coordinates, y down) star-shaped ground-truth polygons, float32."""
    out = []
    for seed, p, cx, cy, r0 in [(1, 5, 300, 200, 80), (2, 17, 310.5, 220.25, 60), (3, 60, 500, 400, 150),
                                (4, 200, 640, 380, 300), (5, 9, 100, 90, 20), (6, 33, 900, 600, 110),
                                (7, 128, 256, 256, 200), (8, 4, 50, 40, 30)]:
        g = np.random.default_rng(seed)
        ang = np.sort(g.random(p) * 2 * np.pi)
        rad = r0 * (0.55 + 0.45 * g.random(p))
        out.append(np.stack([cx + rad * np.cos(ang), cy + 0.8 * rad * np.sin(ang)], 1).astype(np.float32))
    return out


def gen_targets(sh):
    """The NumPy half of the training-side target matching: the reference's own
    uniform_sample / uniform_sample_1d / get_aux_extreme_points / single_uniform_segment_matching
    on seeded polygons; oracle/targets_ref.py must reproduce indices AND float64 points exactly."""
    fu = ref_loader.load_fcose_utils()
    R = sh.RefineNet
    out, v, up = {}, 32, 5
    polys = target_polygons()
    out["n_polys"] = np.int64(len(polys))
    import logging
    me = NS(initial="octagon", num_sampling=v, _logger=logging.getLogger("gen_golden"),
            uniform_sample_1d=R.uniform_sample_1d)
    me.reorder_perloss = lambda a, b: R.reorder_perloss(me, a, b)
    for i, cnt in enumerate(polys):
        p = len(cnt)
        out["poly%d" % i] = cnt
        # uniform_sample: remove branch, small add, the size the training path asks for
        for tag, newp in (("rm", max(3, p - 3)), ("eq", p), ("add", 4 * p + 3), ("big", p * v * up)):
            want = R.uniform_sample(cnt, newp)
            got = targets_ref.uniform_sample(cnt, newp)
            assert want.dtype == got.dtype and np.array_equal(want, got), ("uniform_sample", i, tag)
            if tag != "big":
                out["us_%s_%d" % (tag, i)] = want
                out["us_%s_n%d" % (tag, i)] = np.int64(newp)
        dense_all = R.uniform_sample(cnt, p * v * up)
        # initial contour: octagon of the polygon's extreme points (snake_head.py:1020-1045)
        ex = np.stack([cnt[np.argmin(cnt[:, 1])], cnt[np.argmin(cnt[:, 0])], cnt[np.argmax(cnt[:, 1])],
                       cnt[np.argmax(cnt[:, 0])]])
        octs, eidx = snake_ref.get_octagon(torch.from_numpy(ex)[None], v)
        sampled = octs[0].numpy()
        eidx = eidx[0].numpy()
        tt = int(np.argmin(np.power(dense_all - sampled[0], 2).sum(axis=1)))     # snake_head.py:1139-1144
        dense = np.roll(dense_all, -tt, axis=0)[::p]
        assert dense.shape == (v * up, 2)
        out["dense%d" % i], out["sampled%d" % i], out["edge_idx%d" % i] = dense, sampled, eidx
        # uniform_sample_1d on arcs of the dense contour
        for k, (a, b, m) in enumerate([(0, 40, 7), (10, 17, 6), (3, 90, 30), (0, v * up, v), (5, 7, 1), (20, 60, 64)]):
            want = R.uniform_sample_1d(dense[a:b], m)
            got = targets_ref.uniform_sample_1d(dense[a:b], m)
            assert np.array_equal(want, got), ("uniform_sample_1d", i, k)
            out["s1d_%d_%d" % (i, k)] = want
            out["s1d_arg_%d_%d" % (i, k)] = np.array([a, b, m], np.int64)
        # get_aux_extreme_points
        re_w, ext_w = fu.get_aux_extreme_points(dense)
        re_g, ext_g = targets_ref.get_aux_extreme_points(dense)
        assert np.array_equal(re_w, re_g) and np.array_equal(ext_w, ext_g), ("aux", i)
        out["aux_pts%d" % i], out["aux_idx%d" % i] = re_w, ext_w
        # single_uniform_segment_matching (INITIAL = octagon, NEW_MATCHING)
        want = R.single_uniform_segment_matching(me, dense.copy(), sampled, eidx, up)
        got, status = targets_ref.single_uniform_segment_matching(dense, sampled, eidx, up, v)
        assert status == 0 and np.array_equal(want, got), ("segment_matching", i)
        out["match%d" % i] = want
        # the shipped configuration's form (INITIAL = box): box contour, rolled to the first target
        # of the re-ordered dense contour, then single_uniform_multisegment_matching
        box = torch.tensor([[float(cnt[:, 0].min()), float(cnt[:, 1].min()), float(cnt[:, 0].max()),
                             float(cnt[:, 1].max())]])
        bpts = snake_ref.uniform_upsample(snake_ref.box_polygon(box)[None], v)[0][0].numpy()
        tt = int(np.argmin(np.power(bpts - re_w[0], 2).sum(axis=1)))                 # snake_head.py:1150-1155
        rolled = np.roll(bpts, -tt, axis=0)
        assert np.array_equal(rolled, targets_ref.roll_to_first_target(bpts, re_w))
        me.initial = "box"
        want = R.single_uniform_multisegment_matching(me, re_w.copy(), rolled, ext_w.copy(), up)
        me.initial = "octagon"
        got, status = targets_ref.single_uniform_multisegment_matching(re_w, rolled, ext_w, up, v)
        assert status == 0 and np.array_equal(want, got), ("multisegment_matching", i)
        out["box_pts%d" % i], out["box_rolled%d" % i], out["mmatch%d" % i] = bpts, rolled, want
    # the fall-back of the matching (a collapsed ground truth: every arc has zero length)
    flat = np.tile(np.array([[10.0, 20.0]]), (v * up, 1))
    want = R.single_uniform_segment_matching(me, flat.copy(), sampled, eidx, up)
    got, status = targets_ref.single_uniform_segment_matching(flat, sampled, eidx, up, v)
    assert status == 1 and np.array_equal(want, got)
    out["match_flat"] = want
    out["v"], out["up"] = np.int64(v), np.int64(up)
    _save("g12_targets", **out)


class _RingPolygon:
    """Stand-in for shapely.geometry.Polygon as compute_targets_for_polys uses it (only
    `.exterior.is_ccw`, snake_head.py:1101-1125): Shapely is absent, the orientation predicate is
    oracle/targets_ref.ring_is_ccw (parity of that predicate unpinned, see there)."""

    def __init__(self, pts):
        self.exterior = NS(is_ccw=targets_ref.ring_is_ccw(pts))


def gen_targets_driver(sh):
    """RefineNet.compute_targets_for_polys itself (snake_head.py:982-1254), the shipped INITIAL = "box"
    + NEW_MATCHING configuration, run from the reference's own code on two images of seeded
    polygons plus one image without ground truth: a counter-clockwise polygon (reversed by the
    reference), a fragmented instance (skipped), a zero-area one (skipped)."""
    fu = ref_loader.load_fcose_utils()
    sh.get_aux_extreme_points = fu.get_aux_extreme_points
    sh.get_extreme_points = fu.get_extreme_points
    sh.Polygon = _RingPolygon
    v = 32
    net = sh.RefineNet(ref_loader.dance_cfg(num_sampling=v, num_convs=0, convs_dim=4))
    assert net.initial == "box" and net.new_matching and not net.original
    polys = target_polygons()
    flat = np.array([[10, 10], [20, 20], [30, 30], [40, 40]], np.float32)           # zero area
    inst = [[[polys[0].reshape(-1)], [polys[1][::-1].copy().reshape(-1)], [polys[2].reshape(-1)],
             [polys[3].reshape(-1), polys[4].reshape(-1)]],                           # last: fragmented
            [],
            [[polys[5].reshape(-1)], [flat.reshape(-1)], [polys[6].reshape(-1)], [polys[7].reshape(-1)]]]

    def bounds(parts):
        c = np.concatenate([np.asarray(q).reshape(-1, 2) for q in parts])
        return [float(c[:, 0].min()), float(c[:, 1].min()), float(c[:, 0].max()), float(c[:, 1].max())]

    boxes = [torch.tensor([bounds(q) for q in im], dtype=torch.float32).reshape(-1, 4) for im in inst]
    sizes = [(800, 1216), (800, 1216), (704, 1000)]
    tg = [NS(gt_boxes=NS(tensor=b.clone()), gt_masks=NS(polygons=im)) for b, im in zip(boxes, inst)]
    want = net.compute_targets_for_polys(tg, sizes)
    got = targets_ref.compute_targets_for_polys_box(inst, boxes, sizes, v)
    out = {"v": np.int64(v), "n_images": np.int64(len(inst))}
    assert sorted(want) == sorted(got)
    for k in want:
        a, b = want[k], got[k]
        assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), k
        out["out_" + k] = a.numpy()
    assert want["sample_locs"].shape == (6, v, 2) and want["image_idx"].tolist() == [0, 0, 0, 2, 2, 2]
    for i, im in enumerate(inst):
        out["boxes%d" % i] = boxes[i].numpy()
        out["n_inst%d" % i] = np.int64(len(im))
        for j, parts in enumerate(im):
            out["n_parts%d_%d" % (i, j)] = np.int64(len(parts))
            for k, q in enumerate(parts):
                out["poly%d_%d_%d" % (i, j, k)] = np.asarray(q, np.float32)
    out["sizes"] = np.array(sizes, np.int64)
    _save("g13_targets_driver", **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_flush_denormal(False)
    extreme_utils_ref.build()
    sh, dh = ref_loader.load_reference()
    steps = {"resampler": lambda: gen_resampler(sh), "gather": lambda: gen_gather(sh, dh),
             "blocks": lambda: gen_blocks(sh), "snakenet": lambda: gen_snakenet(sh),
             "dance": lambda: gen_dance(sh), "dsnake": lambda: gen_dsnake(dh),
             "upsample2x": lambda: gen_upsample2x(sh), "bottom_out": lambda: gen_bottom_out(sh),
             "attender": gen_attender, "edge_head": gen_edge_head, "fcose": gen_fcose,
             "targets": lambda: gen_targets(sh), "targets_driver": lambda: gen_targets_driver(sh)}
    # `python -m oracle.gen_golden attender ...` regenerates only the named fixtures (floating-
    # point fixtures move in the last bits with the host CPU's kernels: keep the committed ones)
    for name in (sys.argv[1:] or list(steps)):
        steps[name]()
    print("golden vectors written to", os.path.normpath(OUT))


if __name__ == "__main__":
    sys.exit(main())
