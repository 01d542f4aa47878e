"""GPU: the one-image head as one HIP-graph replay (dance_amd/graph.py) reproduces the eager
forward for the real detections, for changing inputs and detection counts on the same graph --
to the bit: nothing on the path sums in an order that varies (the feature-prep layers' GroupNorm
statistics are per-workgroup partials folded in a fixed order, csrc/common.h k_sum_rows), so two
eager forwards are identical too (SURVEY section 5's run-twice check)."""
import os

import pytest
import torch

from _util import edge_head_cfg
from dance_amd import graph, modules, synthetic
from dance_amd.structures import Boxes, Instances

pytestmark = pytest.mark.gpu

# the opt-out knobs put torch's GroupNorm / convolutions in front of the deformer: those sum
# in an order that varies, so under them eager-twice and graph-vs-eager hold to a tolerance, not to the bit
HIP_FEATPREP = (os.environ.get("DANCE_FEATPREP", "hip") == "hip"
                and os.environ.get("DANCE_PRECISION", "fp16x3") == "fp16x3")


def _same(a, b, tol=0.0):
    if HIP_FEATPREP:
        return torch.equal(a.view(torch.int32), b.view(torch.int32))
    nan_a, nan_b = torch.isnan(a), torch.isnan(b)
    return torch.equal(nan_a, nan_b) and float((a - b)[~nan_a].abs().max()) <= tol


def _spread(a, b):
    """run-to-run distance of two eager forwards under the opt-out knobs (0 on the default path)"""
    ok = ~(torch.isnan(a) | torch.isnan(b))
    return float((a - b)[ok].abs().max()) if bool(ok.any()) else 0.0


def test_graphed_head_equals_eager(device):
    h, w = 192, 256
    cfg, shapes = edge_head_cfg(num_sampling=128, num_convs=2)
    head = modules.EdgeSnakeFPNHead(cfg, shapes)
    sd = {k: v.clone() for k, v in head.state_dict().items()}
    synthetic.fill_state_dict_(sd, 1234)
    head.load_state_dict(sd)
    head = head.to(device).eval()
    g = graph.GraphedEdgeSnakeFPNHead(head, max_instances=24)
    for seed, n in ((1, 24), (2, 7), (3, 0), (4, 19)):
        feats = {k: synthetic.make_smooth_features(1, 256, h // s, w // s, seed + i, attention=False, cell=4).to(device)
                 for i, (k, s) in enumerate((("p2", 4), ("p3", 8), ("p4", 16), ("p5", 32)))}
        boxes = synthetic.make_boxes(max(n, 1), h, w, seed, r_min=6, r_max=60)[:n].to(device)
        att_g, polys_g = g(feats, boxes, (h, w))
        att_g, polys_g = att_g.clone(), polys_g.clone()
        if n == 0:
            assert polys_g.shape[0] == 0
            continue
        with torch.no_grad():
            att_e, _, res = head(feats, [Instances((h, w), pred_boxes=Boxes(boxes.clone()))], None)
            att_e2, _, res2 = head(feats, [Instances((h, w), pred_boxes=Boxes(boxes.clone()))], None)
        want, want2 = res[0].pred_polys.tensor, res2[0].pred_polys.tensor
        # run twice, compare bits: attention map and contours (a box clipped to zero width is a NaN contour
        # in the reference and here -- derange -- at the same places)
        # (opt-out knobs: torch's atomically summed GroupNorm statistics differ run to run and the synthetic
        # features amplify that; the graph must then sit within a few run-to-run spreads of eager)
        tol_att, tol_px = max(1e-5, 4 * _spread(att_e, att_e2)), max(5e-3, 4 * _spread(want, want2))
        assert _same(att_e, att_e2, tol_att) and tol_px < 1.0
        assert _same(want, want2, tol_px)
        # and the graph replays exactly what eager computes
        assert _same(att_g, att_e, tol_att)
        assert _same(polys_g, want, tol_px)
        assert bool(torch.isfinite(polys_g).any())
    assert len(g._graphs) == 1
