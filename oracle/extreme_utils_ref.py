"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/extreme_utils_ref.c for the rules).

ctypes front-end of the C restatement of the reference's `extreme_utils` extension, with
the call signatures of the reference's pybind module `_ext`
(reference: core/layers/extreme_utils/utils.cpp:4-10, utils.h:6-33) on CPU tensors.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libextreme_utils_ref.so")
_lib = None


def build(force=False):
    """Compile the C oracle with the committed recipe (oracle/Makefile)."""
    src = os.path.join(_HERE, "extreme_utils_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def collect_extreme_point(ext_hm, bbox, radius, vote, ct):
    """utils.cu:155-212; returns a zero-initialised (b,n,4,2) int64 tensor, then filled."""
    b, _, h, w = ext_hm.shape
    n = bbox.size(1)
    assert bbox.size(0) == b and vote.shape[:3] == (b, h, w) and ct.shape[:2] == (b, n)
    out = torch.zeros((b, n, 4, 2), dtype=torch.int64)
    ext_hm, bbox, radius = ext_hm.contiguous(), bbox.contiguous(), radius.contiguous()
    vote, ct = vote.contiguous(), ct.contiguous()
    lib().ref_collect_extreme_point(_p(ext_hm), _p(bbox), _p(radius), _p(vote), _p(ct),
                                    _p(out), b, n, h, w)
    return out


def calculate_edge_num(edge_num, edge_num_sum, edge_idx_sort, p_num):
    """utils.cu:257-300; mutates `edge_num` in place (must be contiguous, utils.cu:271)."""
    assert edge_num.is_contiguous() and edge_num.dtype == torch.int64
    b, n, E = edge_num.shape
    s, o = edge_num_sum.contiguous(), edge_idx_sort.contiguous()
    lib().ref_calculate_edge_num(_p(edge_num), _p(s), _p(o), b, n, E,
                                 ctypes.c_int64(int(p_num)))


def calculate_wnp(edge_num, edge_start_idx, p_num):
    """utils.cu:333-374; returns (weight (b,n,p,1) f32, ind (b,n,p,2) i64), zero-initialised."""
    b, n, E = edge_num.shape
    weight = torch.zeros((b, n, p_num, 1), dtype=torch.float32)
    ind = torch.zeros((b, n, p_num, 2), dtype=torch.int64)
    en, es = edge_num.contiguous(), edge_start_idx.contiguous()
    lib().ref_calculate_wnp(_p(en), _p(es), _p(weight), _p(ind), b, n, E, int(p_num))
    return weight, ind


def roll_array(array, step):
    """utils.cu:404-440."""
    b, n, d = array.shape
    out = torch.zeros_like(array, memory_format=torch.contiguous_format)
    a, s = array.contiguous(), step.contiguous()
    lib().ref_roll_array(_p(a), _p(s), _p(out), b, n, d)
    return out


def nms(dets, scores, threshold):
    """utils.h:36-43 + nms.cu:70-128: dets (n,4), scores (n,) -> ascending int64 indices of
    the kept boxes.  Ties in the scores are broken by a stable sort here (the reference's
    unstable device sort leaves them unspecified)."""
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    order = torch.sort(scores.float(), descending=True, stable=True)[1]
    b = torch.cat([dets.float(), scores.float()[:, None]], 1)[order].contiguous()
    keep = torch.zeros((b.size(0),), dtype=torch.int64)
    lib().ref_nms_sorted.restype = ctypes.c_int
    num = lib().ref_nms_sorted(_p(b), b.size(0), ctypes.c_float(float(threshold)), _p(keep))
    return torch.sort(order[keep[:num]])[0]


def batched_nms(boxes, scores, idxs, iou_threshold):
    """torchvision.ops.batched_nms as detectron2.layers.batched_nms calls it (the reference's
    ml_nms, core/layers/ml_nms.py:25): boxes of different classes are moved apart by
    idxs * (max coordinate + 1), then plain NMS (IoU without "+1", strict >); returns the kept
    indices in decreasing score order.  PARITY UNPINNED (torchvision absent)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    off = idxs.to(boxes) * (boxes.max() + 1)
    shifted = boxes.float() + off[:, None]
    order = torch.sort(scores.float(), descending=True, stable=True)[1]
    b = torch.cat([shifted, scores.float()[:, None]], 1)[order].contiguous()
    keep = torch.zeros((b.size(0),), dtype=torch.int64)
    lib().ref_nms_sorted_plain.restype = ctypes.c_int
    num = lib().ref_nms_sorted_plain(_p(b), b.size(0), ctypes.c_float(float(iou_threshold)), _p(keep))
    return order[keep[:num]]


def uniform_upsample_fused(poly, p_num):
    """C restatement of RefineNet.uniform_upsample (snake_head.py:764-794) as one routine.

    poly (N,E,2) f32 -> (points (N,p,2) f32, edge_start_idx (N,E) i64, status (N,) i32).
    """
    N, E, _ = poly.shape
    poly = poly.contiguous().float()
    out = torch.zeros((N, p_num, 2), dtype=torch.float32)
    start = torch.zeros((N, E), dtype=torch.int64)
    status = torch.zeros((N,), dtype=torch.int32)
    if N:
        lib().ref_uniform_upsample(_p(poly), N, E, int(p_num), _p(out), _p(start),
                                   _p(status))
    return out, start, status
