"""Multi-class NMS of the detector's boxes, the step that hands the boxes to the contour path.

    ml_nms(boxlist, nms_thresh, max_proposals=-1, score_field="scores", label_field="labels")
        <- core/layers/ml_nms.py:8-31 (called from fcose_outputs.py:528 / fcos_outputs.py)

The reference delegates to detectron2.layers.batched_nms = torchvision's: boxes of different
classes are shifted apart by `class * (max coordinate + 1)`, then plain greedy NMS (IoU without
"+1", a box is dropped when IoU > threshold) and the kept indices come back in decreasing score
order.  Here the IoU bit mask and the sweep run in libdance_hip.so (`dance_nms_sorted_plain`,
csrc/exact_ops.hip); torch only sorts the scores.  GPU tensors required.
"""
import torch

from ._lib import check, current_stream, lib, ptr


def batched_nms(boxes, scores, idxs, iou_threshold):
    """detectron2.layers.batched_nms / torchvision.ops.batched_nms: (n,4) xyxy, (n,), (n,) ->
    int64 indices of the kept boxes, decreasing score."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    if not boxes.is_cuda:
        raise RuntimeError("dance_amd.ml_nms needs GPU tensors (got a %s tensor)" % boxes.device)
    if boxes.dim() != 2 or boxes.size(1) != 4 or scores.shape != (boxes.size(0),) or idxs.shape != scores.shape:
        raise RuntimeError("batched_nms: boxes (n,4), scores (n,), idxs (n,)")
    n = boxes.size(0)
    offsets = idxs.to(boxes) * (boxes.max() + 1)
    order = torch.sort(scores.float(), descending=True, stable=True)[1]
    b = torch.cat([boxes.float() + offsets[:, None].float(), scores.float()[:, None]], 1)[order].contiguous()
    cb = (n + 63) // 64
    ws = torch.empty(((n + 1) * cb,), dtype=torch.int64, device=boxes.device)
    keep = torch.empty((n,), dtype=torch.int64, device=boxes.device)
    count = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
    with torch.cuda.device(boxes.device):
        check(lib().dance_nms_sorted_plain(ptr(b), n, float(iou_threshold), ptr(ws), ptr(keep), ptr(count),
                                           current_stream()), "batched_nms")
    return order[keep[:int(count.item())]]


def ml_nms(boxlist, nms_thresh, max_proposals=-1, score_field="scores", label_field="labels"):
    """core/layers/ml_nms.py:8-31 (like the reference, the fields read are `pred_boxes`, `scores`
    and `pred_classes` whatever the two *_field arguments say)."""
    if nms_thresh <= 0:
        return boxlist
    keep = batched_nms(boxlist.pred_boxes.tensor, boxlist.scores, boxlist.pred_classes, nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[keep]
