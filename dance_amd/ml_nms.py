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

# Largest list one NMS call takes.  The bit mask is (n + 1) * ceil(n / 64) words -- 537 MB at this
# bound, 137 GB at the 2^20 the C side would accept -- and the greedy sweep is one wavefront
# walking all column blocks serially.  The detector hands over at most PRE_NMS_TOPK (1000) x 5
# levels per image (fcose_outputs.py:492-499).
MAX_BOXES = 65536


def _check_size(n, what):
    if n > MAX_BOXES:
        raise RuntimeError("%s: %d boxes in one list, at most %d are supported (the IoU bit mask grows with "
                           "n^2 / 8 bytes and the sweep is serial); cut the list with a score top-k first"
                           % (what, n, MAX_BOXES))


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
    _check_size(n, "batched_nms")
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


def batched_nms_many(boxes, scores, idxs, iou_threshold):
    """`batched_nms` of several independent lists (one per image) at once: lists of (n_i,4),
    (n_i,), (n_i,) -> list of int64 index tensors, each exactly what batched_nms(boxes[i],
    scores[i], idxs[i]) returns (the class offset uses the list's own largest coordinate, the
    order is decreasing score, ties in input order).  Two stable sorts and two launches for all
    lists (`dance_nms_segments_plain`), one host sync for the kept counts."""
    sizes = [int(b.size(0)) for b in boxes]
    total, n_seg = sum(sizes), len(sizes)
    _check_size(max(sizes, default=0), "batched_nms_many")
    if n_seg == 0:
        return []
    dev = boxes[0].device
    if total == 0:
        return [torch.empty((0,), dtype=torch.int64, device=dev) for _ in sizes]
    if not dev.type == "cuda":
        raise RuntimeError("dance_amd.ml_nms needs GPU tensors (got a %s tensor)" % dev)
    starts, ws_off, seg_rows = [0], [0], []
    for n in sizes:
        seg_rows.append((starts[-1], n, ws_off[-1]))
        starts.append(starts[-1] + n)
        ws_off.append(ws_off[-1] + (n + 1) * ((n + 63) // 64))
    seg = torch.tensor(seg_rows, dtype=torch.int64).to(dev, non_blocking=True)
    b_all = torch.cat([b.float().reshape(-1, 4) for b in boxes])
    s_all = torch.cat([x.float() for x in scores])
    c_all = torch.cat(list(idxs))
    seg_id = torch.repeat_interleave(torch.arange(n_seg, device=dev), torch.tensor(sizes, device=dev),
                                     output_size=total)
    # each list's largest coordinate (an atomic scatter-amax over 16 addresses took 1.3 ms here)
    top = torch.stack([b.max() if b.numel() else b.new_zeros(()) for b in boxes]).float()
    offsets = c_all.to(b_all) * (top + 1)[seg_id]
    by_score = torch.sort(s_all, descending=True, stable=True)[1]
    order = by_score[torch.sort(seg_id[by_score], stable=True)[1]]           # image-major, score-descending
    rows = torch.cat([b_all + offsets[:, None], s_all[:, None]], 1)[order].contiguous()
    ws = torch.empty((ws_off[-1],), dtype=torch.int64, device=dev)
    keep = torch.empty((total,), dtype=torch.int64, device=dev)
    count = torch.empty((n_seg,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib().dance_nms_segments_plain(ptr(rows), ptr(seg), n_seg, max(sizes), float(iou_threshold), ptr(ws),
                                             ptr(keep), ptr(count), current_stream()), "batched_nms_many")
    counts = count.tolist()
    return [order[starts[i] + keep[starts[i]:starts[i] + counts[i]]] - starts[i] for i in range(n_seg)]
