"""Drop-in for the reference's `core.layers.extreme_utils` (`_ext`) module.

Same five function names (utils.cpp:4-10), argument order, dtypes, in-place behaviour and
error type (RuntimeError for non-GPU tensors and shape mismatches, with the reference's
messages: core/layers/extreme_utils/src/utils.cu:162-184, 263-280, 338-348, 408-417), but
backed by the hand-written gfx950 kernels of libdance_hip.so through its C ABI.  `nms`
(utils.h:36-43) is not on the contour path and has no caller in the reference; it is
provided so that the whole `_ext` surface is covered.

A maintainer swaps it in with one line in core/layers/__init__.py:4
    from dance_amd import extreme_utils
"""
import torch

from ._lib import check, current_stream, lib, ptr


def _assert(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def collect_extreme_point(ext_hm, bbox, radius, vote, ct):
    """utils.cu:155-212 -> (b,n,4,2) int64."""
    _assert(ext_hm.is_cuda, "ext_hm must be a CUDA tensor")
    _assert(bbox.is_cuda, "bbox must be a CUDA tensor")
    _assert(radius.is_cuda, "radius must be a CUDA tensor")
    _assert(vote.is_cuda, "vote must be a CUDA tensor")
    _assert(ct.is_cuda, "ct must be a CUDA tensor")
    b, h, w = ext_hm.size(0), ext_hm.size(2), ext_hm.size(3)
    _assert(b == bbox.size(0), "bbox must have the same batch size with ext_hm")
    n = bbox.size(1)
    _assert(b == vote.size(0), "vote must have the same batch size with ext_hm")
    _assert(h == vote.size(1), "vote must have the same batch size with ext_hm")
    _assert(w == vote.size(2), "vote must have the same batch size with ext_hm")
    _assert(b == ct.size(0), "ct must have the same batch size with bbox")
    _assert(n == ct.size(1), "ct must have the same batch size with bbox")
    _assert(ext_hm.dtype == torch.float32 and vote.dtype == torch.float32 and ct.dtype == torch.float32,
            "expected scalar type Float")
    _assert(bbox.dtype == torch.int64 and radius.dtype == torch.int64, "expected scalar type Long")
    out = torch.zeros((b, n, 4, 2), dtype=bbox.dtype, device=bbox.device)
    ext_hm, bbox, radius = ext_hm.contiguous(), bbox.contiguous(), radius.contiguous()
    vote, ct = vote.contiguous(), ct.contiguous()
    with torch.cuda.device(bbox.device):
        check(lib().dance_collect_extreme_point(ptr(ext_hm), ptr(bbox), ptr(radius), ptr(vote),
                                                ptr(ct), ptr(out), b, n, h, w, current_stream()),
              "collect_extreme_point")
    return out


def calculate_edge_num(edge_num, edge_num_sum, edge_idx_sort, p_num):
    """utils.cu:257-300; mutates `edge_num` in place, returns None."""
    _assert(edge_num.is_cuda, "edge_num must be a CUDA tensor")
    _assert(edge_num_sum.is_cuda, "edge_num_sum must be a CUDA tensor")
    _assert(edge_idx_sort.is_cuda, "edge_idx_sort must be a CUDA tensor")
    b, n, e = edge_num.size(0), edge_num.size(1), edge_num.size(2)
    _assert(edge_num.is_contiguous(), "edge_num must be contiguous")
    _assert(b == edge_num_sum.size(0), "edge_num_sum must have the same batch size with edge_num")
    _assert(n == edge_num_sum.size(1), "edge_num_sum must have the same batch size with edge_num")
    _assert(b == edge_idx_sort.size(0), "edge_idx_sort must have the same batch size with edge_num")
    _assert(n == edge_idx_sort.size(1), "edge_idx_sort must have the same batch size with edge_num")
    _assert(e == edge_idx_sort.size(2), "edge_idx_sort must have the same batch size with edge_num")
    _assert(edge_num.dtype == torch.int64 and edge_num_sum.dtype == torch.int64
            and edge_idx_sort.dtype == torch.int64, "expected scalar type Long")
    s, o = edge_num_sum.contiguous(), edge_idx_sort.contiguous()
    with torch.cuda.device(edge_num.device):
        check(lib().dance_calculate_edge_num(ptr(edge_num), ptr(s), ptr(o), b, n, e, int(p_num),
                                             current_stream()), "calculate_edge_num")


def calculate_wnp(edge_num, edge_start_idx, p_num):
    """utils.cu:333-374 -> (weight (b,n,p,1) f32, ind (b,n,p,2) i64)."""
    _assert(edge_num.is_cuda, "edge_num must be a CUDA tensor")
    b, n, e = edge_num.size(0), edge_num.size(1), edge_num.size(2)
    _assert(b == edge_start_idx.size(0), "edge_start_idx must have the same batch size with edge_num")
    _assert(n == edge_start_idx.size(1), "edge_start_idx must have the same batch size with edge_num")
    _assert(e == edge_start_idx.size(2), "edge_start_idx must have the same batch size with edge_num")
    _assert(edge_num.dtype == torch.int64 and edge_start_idx.dtype == torch.int64,
            "expected scalar type Long")
    _assert(edge_start_idx.is_cuda, "edge_start_idx must be a CUDA tensor")
    weight = torch.zeros((b, n, p_num, 1), dtype=torch.float32, device=edge_num.device)
    ind = torch.zeros((b, n, p_num, 2), dtype=torch.int64, device=edge_num.device)
    en, es = edge_num.contiguous(), edge_start_idx.contiguous()
    with torch.cuda.device(edge_num.device):
        check(lib().dance_calculate_wnp(ptr(en), ptr(es), ptr(weight), ptr(ind), b, n, e, int(p_num),
                                        current_stream()), "calculate_wnp")
    return weight, ind


def roll_array(array, step):
    """utils.cu:404-440 -> rolled copy of `array` (b,n,d) f32."""
    _assert(array.is_cuda, "array must be a CUDA tensor")
    _assert(step.is_cuda, "step must be a CUDA tensor")
    b, n, d = array.size(0), array.size(1), array.size(2)
    _assert(b == step.size(0), "step must have the same batch size with array")
    _assert(array.dtype == torch.float32, "expected scalar type Float")
    _assert(step.dtype == torch.int64, "expected scalar type Long")
    out = torch.zeros_like(array, memory_format=torch.contiguous_format)
    a, s = array.contiguous(), step.contiguous()
    with torch.cuda.device(array.device):
        check(lib().dance_roll_array(ptr(a), ptr(s), ptr(out), b, n, d, current_stream()),
              "roll_array")
    return out


def nms(dets, scores, threshold):
    """utils.h:36-43 -> ascending int64 indices of the boxes that survive greedy NMS
    ("+1" IoU convention, strict > threshold).  dets (n,4) f32 xyxy, scores (n,) f32, on the
    GPU; like the reference, an empty input returns an empty CPU tensor, otherwise the
    result lives on the device.  Sorting is torch's (as the reference uses at::sort); the
    mask and the sweep run in libdance_hip.so, without the reference's host round trip
    (one 4-byte count is read back to size the result)."""
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    _assert(dets.is_cuda, "boxes must be a CUDA tensor")
    _assert(dets.dim() == 2 and dets.size(1) == 4 and scores.dim() == 1 and scores.size(0) == dets.size(0),
            "dets must be (n, 4) and scores (n,)")
    n = dets.size(0)
    from .ml_nms import _check_size
    _check_size(n, "nms")
    order = torch.sort(scores.float(), descending=True, stable=True)[1]
    boxes = torch.cat([dets.float(), scores.float()[:, None]], 1)[order].contiguous()
    cb = (n + 63) // 64
    ws = torch.empty(((n + 1) * cb,), dtype=torch.int64, device=dets.device)
    keep = torch.empty((n,), dtype=torch.int64, device=dets.device)
    count = torch.zeros((1,), dtype=torch.int32, device=dets.device)
    with torch.cuda.device(dets.device):
        check(lib().dance_nms_sorted(ptr(boxes), n, float(threshold), ptr(ws), ptr(keep), ptr(count),
                                     current_stream()), "nms")
    return torch.sort(order[keep[:int(count.item())]])[0]
