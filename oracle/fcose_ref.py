"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by dance_amd).

CPU restatement (plain torch, fp32) of the inference half of the reference's FCOSE decode:
    predict_proposals                  core/modeling/fcose/fcose_outputs.py:419-442
    forward_for_single_feature_map     444-521
    select_over_all_levels             523-542
    ExtremePoints.from_boxes           core/structures/pointset.py:110-130
    FCOSE.compute_locations_per_level  core/modeling/fcose/fcose.py:129-142

Pinned by tests/golden/g11_fcose.npz, which oracle/gen_golden.py produced with the reference's
own FCOSEOutputs class.  The multi-class NMS inside (detectron2.layers.batched_nms =
torchvision's, absent here) is the restatement in oracle/extreme_utils_ref.py on both sides:
that step is PARITY UNPINNED, everything before it is pinned.
"""
import torch

from . import extreme_utils_ref as cref


def compute_locations(shapes, strides):
    """fcose.py:129-142 for every level: (h*w, 2) x,y of the cell centres, row-major."""
    out = []
    for (h, w), s in zip(shapes, strides):
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) * s, torch.arange(w, dtype=torch.float32) * s,
                                indexing="ij")
        out.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], 1) + s // 2)
    return out


def level_candidates(locations, box_cls, reg, ext, ctr, thresh, with_ctr, top_n):
    """One level, every image (fcose_outputs.py:444-521): list of dicts of tensors."""
    n, c, h, w = box_cls.shape
    s = torch.sigmoid(box_cls.permute(0, 2, 3, 1).reshape(n, h * w, c))
    cs = torch.sigmoid(ctr.reshape(n, h * w))
    prod = s * cs[:, :, None]
    cand = (prod if with_ctr else s) > thresh
    reg = reg.permute(0, 2, 3, 1).reshape(n, h * w, 4)
    ext = None if ext is None else ext.permute(0, 2, 3, 1).reshape(n, h * w, 4)
    out = []
    for i in range(n):
        loc_i, cls_i = torch.nonzero(cand[i], as_tuple=True)
        v = prod[i][loc_i, cls_i]
        if v.numel() > top_n:
            v, k = torch.topk(v, top_n, sorted=False)
            loc_i, cls_i = loc_i[k], cls_i[k]
        p, r = locations[loc_i], reg[i][loc_i]
        box = torch.stack([p[:, 0] - r[:, 0], p[:, 1] - r[:, 1], p[:, 0] + r[:, 2], p[:, 1] + r[:, 3]], 1)
        d = dict(boxes=box, scores=torch.sqrt(v), classes=cls_i, locations=p, raw=v)
        if ext is not None:
            e = ext[i][loc_i]
            bw, bh = box[:, 2] - box[:, 0], box[:, 3] - box[:, 1]
            d["ext"] = torch.stack([p[:, 0] + bw * e[:, 0], box[:, 1], box[:, 0], p[:, 1] + bh * e[:, 1],
                                    p[:, 0] + bw * e[:, 2], box[:, 3], box[:, 2], p[:, 1] + bh * e[:, 3]],
                                   1).view(-1, 4, 2)
        out.append(d)
    return out


def select(d, nms_thresh, post_top_n):
    """fcose_outputs.py:523-542 for one image."""
    if nms_thresh > 0:
        keep = cref.batched_nms(d["boxes"], d["scores"], d["classes"], nms_thresh)
        d = {k: v[keep] for k, v in d.items()}
    m = d["scores"].numel()
    if m > post_top_n > 0:
        th = torch.kthvalue(d["scores"], m - post_top_n + 1)[0]
        keep = torch.nonzero(d["scores"] >= th.item()).squeeze(1)
        d = {k: v[keep] for k, v in d.items()}
    return d


def predict_proposals(locations, logits, reg, ext, ctr, strides, thresh, with_ctr, pre_top_n, nms_thresh,
                      post_top_n, before_nms=False):
    """fcose_outputs.py:419-442 -> one dict per image (boxes, scores, classes, locations[, ext])."""
    per_level = [level_candidates(locations[l], logits[l], reg[l] * strides[l], None if ext is None else ext[l],
                                  ctr[l], thresh, with_ctr, pre_top_n) for l in range(len(logits))]
    out = []
    for i in range(logits[0].size(0)):
        d = {k: torch.cat([lv[i][k] for lv in per_level], 0) for k in per_level[0][i]}
        out.append(d if before_nms else select(d, nms_thresh, post_top_n))
    return out
