"""Minimal tensor wrappers the contour path hands back to its caller.

The reference carries contours on detectron2 `Instances` as `PolygonPoints` /
`ExtremePoints` (core/structures/pointset.py:8-22, 221-234).  detectron2 is not a
dependency here, so only the tensor semantics the path itself touches are restated; a
Detectron2 deployment passes its own classes through `RefineNet(..., polygon_cls=...)`.
"""
import torch


class _PointSet:
    _shape_tail = (128, 2)

    def __init__(self, tensor):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device("cpu")
        tensor = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = torch.zeros((0,) + self._shape_tail, dtype=torch.float32, device=device)
        assert tensor.dim() == 3 and tensor.size(-1) == 2, tensor.size()
        self.tensor = tensor

    def clone(self):
        return type(self)(self.tensor.clone())

    def to(self, device):
        return type(self)(self.tensor.to(device))

    def scale(self, scale_x, scale_y):
        self.tensor[:, :, 0] *= scale_x
        self.tensor[:, :, 1] *= scale_y

    def __getitem__(self, item):
        if isinstance(item, int):
            return type(self)(self.tensor[item][None])
        return type(self)(self.tensor[item])

    def __len__(self):
        return self.tensor.shape[0]

    @property
    def device(self):
        return self.tensor.device


class PolygonPoints(_PointSet):
    """(N, V, 2) contour vertices (pointset.py:221-234)."""

    def clip(self, box_size):
        h, w = box_size
        self.tensor[:, :, 0].clamp_(min=0, max=w)
        self.tensor[:, :, 1].clamp_(min=0, max=h)

    def flatten(self):
        """(N, 2V) view, what the evaluator hands to get_polygon_rles (pointset.py:253-257);
        an empty set stays (0, V, 2) as in the reference."""
        n = self.tensor.size(0)
        return self.tensor if n == 0 else self.tensor.reshape(n, -1)

    def get_box(self):
        """(N,4) xyxy hull of the vertices (pointset.py:259-260)."""
        return torch.cat([self.tensor.min(1)[0], self.tensor.max(1)[0]], dim=1)

    def __repr__(self):
        return "PolyPts(" + str(self.tensor) + ")"


class ExtremePoints(_PointSet):
    """(N, 4, 2) extreme points in order top, left, bottom, right (pointset.py:8-22)."""

    _shape_tail = (4, 2)

    def fit_to_box(self):
        """pointset.py:132-142: pull every point back inside the box the four span."""
        t = self.tensor
        box = torch.stack([t[:, 1, 0], t[:, 0, 1], t[:, 3, 0], t[:, 2, 1]], dim=1)
        n = box.size(0)
        lower = box.view(-1, 2, 2)[:, :1, :]
        upper = box.view(-1, 2, 2)[:, 1:, :]
        lo, hi = t < lower, t > upper
        if lo.any():
            t[lo] = lower.expand(n, 4, 2)[lo]
        if hi.any():
            t[hi] = upper.expand(n, 4, 2)[hi]

    def __repr__(self):
        return "ExtPts(" + str(self.tensor) + ")"
