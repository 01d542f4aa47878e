"""FCOSE box decode: the producer of the boxes and extreme points the contour path starts from
(SURVEY 8f row 3).  Inference half of the reference's `FCOSEOutputs`
(core/modeling/fcose/fcose_outputs.py:125-173, 419-542), same constructor and method names:

    FCOSEOutputs(images, locations, logits_pred, reg_pred, ext_pred, ctrness_pred, ...,
                 thresh_with_ctr, gt_instances=None).predict_proposals() -> list[Instances]

Fields of each result as the reference's: pred_boxes, scores, pred_classes, locations,
ext_points (when ext_pred is given).  The threshold / `nonzero` / gather / decode of all levels
and images runs in libdance_hip.so (csrc/fcose.hip: two launches over all levels' logits and one
prefix sum, one device -> host copy of the segment boundaries instead of the reference's two syncs per
image and level); the per-level top-k is torch.topk on the level's slice, the multi-class NMS is
dance_amd.ml_nms.  `losses()` (training) is out of scope and raises.
"""
import torch

from . import ml_nms as _ml_nms
from ._lib import FcoseLevelStruct, check, current_stream, lib, ptr
from .structures import Boxes, ExtremePoints, Instances


def compute_locations(features, strides):
    """FCOSE.compute_locations (core/modeling/fcose/fcose.py:118-142): per level the (h*w, 2)
    image coordinates (x, y) of the feature map's cells, row-major."""
    out = []
    for f, s in zip(features, strides):
        h, w = f.size()[-2:]
        xs = torch.arange(0, w * s, step=s, dtype=torch.float32, device=f.device)
        ys = torch.arange(0, h * s, step=s, dtype=torch.float32, device=f.device)
        out.append(torch.stack((xs.repeat(h), ys.repeat_interleave(w)), dim=1) + s // 2)
    return out


def decode_candidates(locations, logits_pred, reg_pred, ext_pred, ctrness_pred, strides, pre_nms_thresh,
                      thresh_with_ctr):
    """All levels and images at once -> (fields, bounds): fields = dict of flat tensors over the
    candidates in the reference's order (image, level, location, class): raw (score before the
    square root), scores, classes, locations, boxes, ext (or None); bounds[i][l] .. bounds[i][l+1]
    = slice of image i, level l (Python ints; bounds[i] has one entry per level plus the end)."""
    dev = logits_pred[0].device
    if not dev.type == "cuda":
        raise RuntimeError("dance_amd.fcose_outputs needs GPU tensors (got a %s tensor)" % dev)
    n, c = logits_pred[0].size(0), logits_pred[0].size(1)
    hws = [int(x.size(2) * x.size(3)) for x in logits_pred]
    lvl_off = [0]
    for hw in hws:
        lvl_off.append(lvl_off[-1] + hw)
    total = lvl_off[-1]
    cls = [x.float().contiguous() for x in logits_pred]
    ctr = [x.float().contiguous() for x in ctrness_pred]
    reg = [x.float().contiguous() for x in reg_pred]
    ext = [x.float().contiguous() for x in ext_pred] if ext_pred is not None else None
    locs = [x.float().contiguous() for x in locations]
    for l, hw in enumerate(hws):
        if cls[l].shape[:2] != (n, c) or ctr[l].numel() != n * hw or reg[l].numel() != n * 4 * hw \
                or locs[l].shape != (hw, 2) or (ext is not None and ext[l].numel() != n * 4 * hw):
            raise RuntimeError("fcose decode: level %d: logits (n,c,h,w), ctrness (n,1,h,w), reg / ext "
                               "(n,4,h,w), locations (h*w,2) expected" % l)
    if len(hws) > 8:
        raise RuntimeError("fcose decode: at most 8 levels (DANCE_FCOSE_MAX_LEVELS)")
    counts = torch.empty((n, total), dtype=torch.int32, device=dev)
    flag = 1 if thresh_with_ctr else 0
    table = (FcoseLevelStruct * len(hws))()
    for l, hw in enumerate(hws):
        table[l] = FcoseLevelStruct(ptr(cls[l]), ptr(ctr[l]), ptr(reg[l]), ptr(ext[l]) if ext is not None else None,
                                    ptr(locs[l]), hw, float(strides[l]))
    with torch.cuda.device(dev):
        st = current_stream()
        check(lib().dance_fcose_count(table, len(hws), ptr(counts), n, c, float(pre_nms_thresh), flag, st), "fcose_count")
        cum = torch.cumsum(counts.view(-1), 0, dtype=torch.int64)
        # segment starts of every (image, level) and the grand total: the one host copy
        # (a segment starts where the inclusive sum stood one slot earlier; the last entry is the total)
        if n * total:
            ends = [i * total + o - 1 for i in range(n) for o in lvl_off[:-1]][1:] + [n * total - 1]
            starts = [0] + cum[torch.tensor(ends, device=dev)].tolist()
        else:
            starts = [0] * (n * len(hws) + 1)
        m = starts[-1]
        raw = torch.empty((m,), dtype=torch.float32, device=dev)
        scores = torch.empty((m,), dtype=torch.float32, device=dev)
        classes = torch.empty((m,), dtype=torch.int64, device=dev)
        out_loc = torch.empty((m, 2), dtype=torch.float32, device=dev)
        boxes = torch.empty((m, 4), dtype=torch.float32, device=dev)
        extpts = torch.empty((m, 4, 2), dtype=torch.float32, device=dev) if ext is not None else None
        if m:
            check(lib().dance_fcose_fill(table, len(hws), ptr(counts), ptr(cum), n, c, float(pre_nms_thresh), flag,
                                         ptr(raw), ptr(scores), ptr(classes), ptr(out_loc), ptr(boxes),
                                         ptr(extpts) if extpts is not None else None, st), "fcose_fill")
    nl = len(hws)
    bounds = [starts[i * nl:(i + 1) * nl] + [starts[(i + 1) * nl]] for i in range(n)]
    return dict(raw=raw, scores=scores, classes=classes, locations=out_loc, boxes=boxes, ext=extpts), bounds


class FCOSEOutputs(object):
    def __init__(self, images, locations, logits_pred, reg_pred, ext_pred, ctrness_pred, focal_loss_alpha,
                 focal_loss_gamma, iou_loss, ext_loss, center_sample, sizes_of_interest, strides, radius,
                 num_classes, pre_nms_thresh, pre_nms_top_n, nms_thresh, fpn_post_nms_top_n, thresh_with_ctr,
                 gt_instances=None, instances_cls=Instances, boxes_cls=Boxes, ext_cls=ExtremePoints):
        """fcose_outputs.py:126-173.  `images` needs `len()` and `.image_sizes`.  A Detectron2
        deployment passes its own Instances / Boxes / ExtremePoints classes through the last
        three arguments."""
        self.logits_pred, self.reg_pred, self.ext_pred, self.ctrness_pred = logits_pred, reg_pred, ext_pred, ctrness_pred
        self.locations = locations
        self.gt_instances = gt_instances
        self.num_feature_maps = len(logits_pred)
        self.num_images = len(images)
        self.image_sizes = images.image_sizes
        self.focal_loss_alpha, self.focal_loss_gamma = focal_loss_alpha, focal_loss_gamma
        self.iou_loss, self.ext_loss = iou_loss, ext_loss
        self.center_sample, self.sizes_of_interest = center_sample, sizes_of_interest
        self.strides, self.radius, self.num_classes = strides, radius, num_classes
        self.pre_nms_thresh, self.pre_nms_top_n = pre_nms_thresh, pre_nms_top_n
        self.nms_thresh, self.fpn_post_nms_top_n = nms_thresh, fpn_post_nms_top_n
        self.thresh_with_ctr = thresh_with_ctr
        self._instances, self._boxes, self._ext = instances_cls, boxes_cls, ext_cls

    def losses(self):
        raise NotImplementedError("dance_amd.FCOSEOutputs covers inference (predict_proposals); "
                                  "training losses are outside the contour path's scope")

    def _boxlist(self, fields, b, image_size):
        """One image: the reference's forward_for_single_feature_map results of the levels in
        `b` (fcose_outputs.py:472-521), already concatenated as predict_proposals does (438-439).
        The candidates of an image are contiguous in level order, so this is a slice -- or an
        index when a level has more candidates than pre_nms_top_n and the reference's top-k
        applies (492-499; `sorted=False`: the order among the kept is unspecified there)."""
        sel = slice(b[0], b[-1])
        if any(b[l + 1] - b[l] > self.pre_nms_top_n for l in range(len(b) - 1)):
            parts = []
            for l in range(len(b) - 1):
                lo, hi = b[l], b[l + 1]
                if hi - lo > self.pre_nms_top_n:
                    parts.append(torch.topk(fields["raw"][lo:hi], self.pre_nms_top_n, sorted=False)[1] + lo)
                else:
                    parts.append(torch.arange(lo, hi, device=fields["raw"].device))
            sel = torch.cat(parts)
        inst = self._instances(image_size)
        inst.pred_boxes = self._boxes(fields["boxes"][sel])
        inst.scores = fields["scores"][sel]
        inst.pred_classes = fields["classes"][sel]
        inst.locations = fields["locations"][sel]
        if fields["ext"] is not None:
            inst.ext_points = self._ext(fields["ext"][sel])
        return inst

    def forward_for_single_feature_map(self, locations, box_cls, reg_pred, ext_pred, ctrness, image_sizes):
        """fcose_outputs.py:444-521 for one level; `reg_pred` already multiplied by the level's
        stride as predict_proposals does (431)."""
        fields, bounds = decode_candidates([locations], [box_cls], [reg_pred], None if ext_pred is None else [ext_pred],
                                           [ctrness], [1.0], self.pre_nms_thresh, self.thresh_with_ctr)
        return [self._boxlist(fields, b, image_sizes[i]) for i, b in enumerate(bounds)]

    def predict_proposals(self):
        """fcose_outputs.py:419-442."""
        fields, bounds = decode_candidates(self.locations, self.logits_pred, self.reg_pred, self.ext_pred,
                                           self.ctrness_pred, self.strides, self.pre_nms_thresh,
                                           self.thresh_with_ctr)
        boxlists = [self._boxlist(fields, b, self.image_sizes[i]) for i, b in enumerate(bounds)]
        return self.select_over_all_levels(boxlists)

    def select_over_all_levels(self, boxlists):
        """fcose_outputs.py:523-542: multi-class NMS, then the fpn_post_nms_top_n best by score
        (ties at the threshold all stay, as with the reference's `>=`).  The NMS of all images
        runs as one segmented call (ml_nms.batched_nms_many); the survivors come back in
        decreasing score, so the reference's kthvalue / `>=` selection is a prefix of them."""
        if self.nms_thresh <= 0 or not boxlists:
            return [self._cut_unsorted(b) for b in boxlists]
        if len(boxlists) == 1:
            # one image (the reference's inference mode): the per-image call has fewer launches
            result = _ml_nms.ml_nms(boxlists[0], self.nms_thresh)
            top = self.fpn_post_nms_top_n
            if len(result) > top > 0:
                result = result[:int((result.scores >= result.scores[top - 1]).sum())]
            return [result]
        keeps = _ml_nms.batched_nms_many([b.pred_boxes.tensor for b in boxlists], [b.scores for b in boxlists],
                                         [b.pred_classes for b in boxlists], self.nms_thresh)
        # one gather per field for all images, then per-image views (instead of one indexing
        # kernel per field and image)
        dev = keeps[0].device
        sizes, kept = [len(b) for b in boxlists], [int(k.numel()) for k in keeps]
        starts = [0]
        for n in sizes:
            starts.append(starts[-1] + n)
        index = torch.cat([k + st for k, st in zip(keeps, starts)])
        top, keep_n = self.fpn_post_nms_top_n, list(kept)
        over = [i for i, m in enumerate(kept) if m > top > 0]
        offs = [0]
        for m in kept:
            offs.append(offs[-1] + m)
        if over:
            # how many scores reach the top-th best (>= : ties stay): one host copy for all images
            sc = torch.cat([b.scores for b in boxlists])[index]
            seg = torch.repeat_interleave(torch.arange(len(kept), device=dev), torch.tensor(kept, device=dev),
                                          output_size=offs[-1])
            kth = sc[torch.tensor([offs[i] + top - 1 if i in over else offs[i] for i in range(len(kept))],
                                  device=dev).clamp_(max=max(offs[-1] - 1, 0))]
            tied = torch.zeros(len(kept), dtype=torch.int64, device=dev).scatter_add_(
                0, seg, (sc >= kth[seg]).to(torch.int64)).tolist()
            for i in over:
                keep_n[i] = tied[i]
        fields = {}
        for name, v0 in boxlists[0].get_fields().items():
            vals = [b.get_fields()[name] for b in boxlists]
            if isinstance(v0, torch.Tensor):
                fields[name] = (None, torch.cat(vals)[index])
            else:
                fields[name] = (type(v0), torch.cat([v.tensor for v in vals])[index])
        results = []
        for i, b in enumerate(boxlists):
            r = type(b)(b.image_size)
            for name, (cls, t) in fields.items():
                piece = t[offs[i]:offs[i] + keep_n[i]]
                setattr(r, name, piece if cls is None else cls(piece))
            results.append(r)
        return results

    def _cut_unsorted(self, result):
        """The reference's cut on a list in arbitrary order (nms_thresh <= 0: ml_nms returns its
        input, ml_nms.py:20-21)."""
        number_of_detections = len(result)
        if number_of_detections > self.fpn_post_nms_top_n > 0:
            cls_scores = result.scores
            image_thresh, _ = torch.kthvalue(cls_scores.cpu(), number_of_detections - self.fpn_post_nms_top_n + 1)
            result = result[torch.nonzero(cls_scores >= image_thresh.item()).squeeze(1)]
        return result
