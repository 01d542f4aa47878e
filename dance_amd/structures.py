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
        """pointset.py:247-251, the finite check included: a collapsed contour (`derange` divides by
        max - min with no epsilon, snake_head.py:1588-1593) is NaN there and here, and the reference
        raises at this point instead of rasterising garbage (one host sync, as in the reference;
        DANCE_SYNC_CHECKS=lazy raises it from a later call instead)."""
        if self.tensor.is_cuda:
            from . import ops                    # the assert in the mode DANCE_SYNC_CHECKS names (ops.sync_checks)
            ops.assert_not(~torch.isfinite(self.tensor).all(), "Polygon tensor contains infinite or NaN!")
        else:
            assert torch.isfinite(self.tensor).all(), "Polygon tensor contains infinite or NaN!"
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


class Boxes:
    """(N, 4) xyxy boxes: the part of detectron2.structures.Boxes the detector's decode and the
    postprocessing touch (`tensor`, indexing, `clip`, `scale`, `nonempty`, `area`)."""

    def __init__(self, tensor):
        self.tensor = tensor.reshape(-1, 4) if tensor.numel() == 0 else tensor

    def clip(self, box_size):
        h, w = box_size
        self.tensor[:, 0].clamp_(min=0, max=w)
        self.tensor[:, 1].clamp_(min=0, max=h)
        self.tensor[:, 2].clamp_(min=0, max=w)
        self.tensor[:, 3].clamp_(min=0, max=h)

    def scale(self, scale_x, scale_y):
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def nonempty(self, threshold=0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])

    def __len__(self):
        return self.tensor.shape[0]

    @property
    def device(self):
        return self.tensor.device

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"


class Instances:
    """Per-image bag of equally long fields: the part of detectron2.structures.Instances the
    detector's decode, `ml_nms` and the contour heads use (attribute fields, `has`,
    `get_fields`, indexing, `len`, `cat`)."""

    def __init__(self, image_size, **fields):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in fields.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return self._fields[name]

    def set(self, name, value):
        for v in self._fields.values():
            assert len(v) == len(value), "Adding a field of length {} to a Instances of length {}".format(
                len(value), len(v))
            break
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def __getitem__(self, item):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        raise NotImplementedError("Empty Instances does not support __len__!")

    @staticmethod
    def cat(instance_lists):
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        ret = Instances(instance_lists[0].image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                values = type(v0)(torch.cat([v.tensor for v in values], dim=0))
            ret.set(k, values)
        return ret
