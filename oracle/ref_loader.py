"""ORACLE -- TEST INFRASTRUCTURE ONLY; runs in the BUILD CONTAINER only.

Loads the reference's Python snake heads straight from /root/reference (read-only, never
copied) so that oracle/gen_golden.py can produce golden vectors with the reference's own
code.  Third-party packages the reference imports but this image lacks (detectron2,
pycocotools, shapely, fvcore) are satisfied with minimal import-time placeholders; the
reference's `core.layers.extreme_utils` (an unbuildable CUDA extension) is satisfied by
the C restatement in oracle/extreme_utils_ref.c -- which is exactly how that restatement
gets pinned: the reference's `uniform_upsample` drives it.

Nothing here is importable on the GPU box (/root/reference does not exist there).
"""
import importlib.util
import sys
import types
from types import SimpleNamespace as NS

import torch
import torch.nn as nn

from . import extreme_utils_ref

REF = "/root/reference"


class _Conv2d(nn.Conv2d):
    """Signature-compatible stand-in for detectron2.layers.Conv2d (norm=, activation=)."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = super().forward(x)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def _cat(tensors, dim=0):
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)


class _Dummy(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


class _TensorBox:
    """Placeholder for core.structures.{ExtremePoints,PolygonPoints}: holds `.tensor`."""

    def __init__(self, tensor):
        self.tensor = tensor


class _SmoothL1:
    def __init__(self, beta=0.11):
        self.beta = beta


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_loaded = {}


def load_reference():
    """Returns (snake_head_module, dsnake_head_module) loaded from /root/reference."""
    if _loaded:
        return _loaded["snake"], _loaded["dsnake"]
    _module("detectron2")
    _module("detectron2.layers", Conv2d=_Conv2d, DeformConv=_Dummy, ModulatedDeformConv=_Dummy,
            cat=_cat, ShapeSpec=NS)
    _module("pycocotools")
    _module("pycocotools.mask")
    _module("shapely")
    _module("shapely.geometry", Polygon=object)
    _module("fvcore")
    _module("fvcore.nn")
    _module("fvcore.nn.weight_init")
    timer = _load("_dance_ref_timer", REF + "/core/utils/timer.py")
    _module("core")
    _module("core.layers", DFConv2d=_Dummy, SmoothL1Loss=_SmoothL1,
            extreme_utils=extreme_utils_ref)
    _module("core.modeling")
    _module("core.modeling.fcose")
    _module("core.modeling.fcose.utils", get_aux_extreme_points=None, get_extreme_points=None)
    _module("core.structures", ExtremePoints=_TensorBox, PolygonPoints=_TensorBox)
    _module("core.utils", timer=timer)
    _loaded["snake"] = _load("_dance_ref_snake_head",
                             REF + "/core/modeling/edge_snake/snake_head.py")
    _loaded["dsnake"] = _load("_dance_ref_dsnake_head",
                              REF + "/core/modeling/dsnake_baseline/dsnake_head.py")
    return _loaded["snake"], _loaded["dsnake"]


def load_fcose_utils():
    """The reference's core/modeling/fcose/utils.py (get_aux_extreme_points, get_extreme_points:
    pure NumPy), loaded by path; its detectron2 / pycocotools imports get placeholders."""
    if "fcose_utils" in _loaded:
        return _loaded["fcose_utils"]
    load_reference()
    _module("detectron2.structures")
    _module("detectron2.structures.boxes", Boxes=_TensorBox)
    _module("core.structures.points_set", ExtremePoints=_TensorBox)
    _loaded["fcose_utils"] = _load("_dance_ref_fcose_utils", REF + "/core/modeling/fcose/utils.py")
    return _loaded["fcose_utils"]


def dance_cfg(num_sampling=196, initial="box", num_convs=2, attention=True,
              individual_scale=True, de_loc_type="derange", convs_dim=256):
    """configs/Dance_R_50_3x.yaml on top of core/config/defaults.py -- only the keys
    RefineNet.__init__ (snake_head.py:434-542) and _SnakeNet (220-258) read."""
    snake = NS(
        DETACH=True, ORIGINAL=False, STRUCTURE="sequential", CONV_TYPE="ccn", FEAT_DIM=128,
        NUM_ITER=(0, 0, 1), NUM_CONVS=num_convs, STRONGER=False, MULTI_OFFSET=1, SKIP=False,
        NUM_LAYER=(8, 8, 8), CIR_DILATIONS=((1, 1, 1, 2, 2, 4, 4),) * 3,
        MSCORE_SNAKE_ON=False, PRE_OFFSET=False, USE_ASPP=False, USE_PSP=False,
        LAST_UP_SAMPLE=False, UP_SAMPLE_RATE=2, LAST_CHAMFER=False, LAST_NEIGHBOR=False,
        TRACK_PATH=False, NEW_MATCHING=True, INITIAL=initial, DE_LOC_TYPE=de_loc_type,
        INDIVIDUAL_SCALE=individual_scale, LOSS_TYPE="smoothl1", LOSS_ADAPTIVE=False,
        LOSS_SEPARATE_REFINE=False, LOSS_WEIGH=False, LOSS_DISTRIBUTION=(1 / 3, 1 / 3, 2 / 3),
        LOSS_L1_BETA=0.11, EDGE_IN=False, PRED_EDGE=False, EDGE_IN_SEPARATE=(False, False),
        DILATIONS=(1, 1), COORD_CONV=(False, False), FILTER_WIDTH=4,
        USE_DEFORMABLE=(False, False), NUM_SAMPLING=num_sampling, MARK_INDEX=False,
        REORDER_METHOD="dsnake", JITTERING=0.0, POINT_WEIGH=False, ATTENTION=attention,
        SELECTIVE_REFINE=False, DOUBLE_SELECTIVE_REFINE=False, VIS_PATH=False,
    )
    return NS(MODEL=NS(EDGE_HEAD=NS(CONVS_DIM=convs_dim, NORM="GN"),
                       DEFORM_HEAD=NS(NUM_CONVS=256), SNAKE_HEAD=snake))


def dsnake_cfg(num_sampling=128, num_convs=2, convs_dim=256):
    """configs/Dsnake_R_50_1x.yaml on defaults.py -- keys DsnakeHead.__init__ reads
    (dsnake_head.py:198-259)."""
    snake = NS(CONV_TYPE="ccn", INITIAL="octagon", FILTER_WIDTH=4, NUM_ITER=(0, 0, 1),
               NUM_CONVS=num_convs, NUM_SAMPLING=num_sampling, DE_LOC_TYPE="derange",
               DILATIONS=(1, 1), REORDER_METHOD="dsnake", VIS_PATH=False,
               LOSS_TYPE="smoothl1", LOSS_L1_BETA=0.11)
    return NS(MODEL=NS(EDGE_HEAD=NS(CONVS_DIM=convs_dim, NORM="GN"), SNAKE_HEAD=snake))


class FakeBoxes:
    """Duck-typed detectron2 Boxes: `.tensor`, `.clip(image_size)` (in place)."""

    def __init__(self, tensor):
        self.tensor = tensor

    def clip(self, box_size):
        h, w = box_size
        self.tensor[:, 0].clamp_(min=0, max=w)
        self.tensor[:, 1].clamp_(min=0, max=h)
        self.tensor[:, 2].clamp_(min=0, max=w)
        self.tensor[:, 3].clamp_(min=0, max=h)


class FakeInstances:
    """Duck-typed detectron2 Instances carrying only what the snake heads touch."""

    def __init__(self, image_size, **fields):
        self.image_size = image_size
        self._fields = set()
        for k, v in fields.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if not k.startswith("_") and k != "image_size":
            self._fields.add(k)
        object.__setattr__(self, k, v)

    def has(self, name):
        return name in self._fields


class _Registry:
    """Placeholder for detectron2.utils.registry.Registry (register() decorator + get())."""

    def __init__(self, name):
        self._name, self._map = name, {}

    def register(self, obj=None):
        def deco(o):
            self._map[o.__name__] = o
            return o
        return deco if obj is None else deco(obj)

    def get(self, name):
        return self._map[name]


def load_edge_head():
    """The reference's core/modeling/edge_snake/edge_det.py (EdgeSnakeFPNHead: scale heads,
    edge predictor, attender, refine head), loaded from /root/reference as a member of a
    placeholder package so that its `from .snake_head import RefineNet` resolves to the
    reference's own snake_head.py."""
    if "edge" in _loaded:
        return _loaded["edge"]
    snake, _ = load_reference()
    sys.modules["detectron2.layers"].ShapeSpec = NS
    _module("detectron2.utils")
    _module("detectron2.utils.registry", Registry=_Registry)
    _module("detectron2.structures", Instances=FakeInstances, Boxes=FakeBoxes)
    sys.modules["fvcore.nn.weight_init"].c2_msra_fill = lambda m: None     # weights are seeded afterwards
    sys.modules["fvcore.nn"].weight_init = sys.modules["fvcore.nn.weight_init"]
    sys.modules["core.layers"].DiceLoss = lambda *a, **k: None             # training only
    pkg = _module("_dance_ref_edge_pkg")
    pkg.__path__ = []
    sys.modules["_dance_ref_edge_pkg.snake_head"] = snake
    _loaded["edge"] = _load("_dance_ref_edge_pkg.edge_det", REF + "/core/modeling/edge_snake/edge_det.py")
    return _loaded["edge"]


def edge_head_cfg(**snake_over):
    """dance_cfg() plus the keys EdgeSnakeFPNHead.__init__ reads (edge_det.py:40-64, 102-104;
    values: configs/Base-dance.yaml + core/config/defaults.py)."""
    cfg = dance_cfg(**snake_over)
    eh = cfg.MODEL.EDGE_HEAD
    eh.NAME, eh.IN_FEATURES, eh.IGNORE_VALUE, eh.NUM_CLASSES = "EdgeSnakeFPNHead", ["p2", "p3", "p4", "p5"], 255, 1
    eh.COMMON_STRIDE, eh.LOSS_WEIGHT, eh.BCE_WEIGHT, eh.TRAIN, eh.STRONG_FEAT = 4, 1.0, 0.0, True, True
    cfg.MODEL.SNAKE_HEAD.EDGE_IN_TH = -1.0
    cfg.TEST = NS(GT_IN=NS(ON=False, WHAT=("instance",)))
    return cfg


class _Instances:
    """Placeholder for detectron2.structures.Instances as the detector's decode uses it:
    attribute fields, indexing, len, cat (fcose_outputs.py:439, 508-519; ml_nms.py:22-28)."""

    def __init__(self, image_size, **fields):
        object.__setattr__(self, "image_size", image_size)
        object.__setattr__(self, "_f", {})
        for k, v in fields.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        self._f[k] = v

    def __getattr__(self, k):
        f = object.__getattribute__(self, "_f")
        if k not in f:
            raise AttributeError(k)
        return f[k]

    def has(self, name):
        return name in self._f

    def __len__(self):
        return len(next(iter(self._f.values())))

    def __getitem__(self, item):
        return _Instances(self.image_size, **{k: v[item] for k, v in self._f.items()})

    @staticmethod
    def cat(lst):
        out = _Instances(lst[0].image_size)
        for k, v0 in lst[0]._f.items():
            vals = [x._f[k] for x in lst]
            setattr(out, k, torch.cat(vals, 0) if isinstance(v0, torch.Tensor)
                    else type(v0)(torch.cat([v.tensor for v in vals], 0)))
        return out


class _Boxes(FakeBoxes):
    def __getitem__(self, item):
        return _Boxes(self.tensor[item])

    def __len__(self):
        return self.tensor.shape[0]


def load_fcose_outputs():
    """The reference's core/modeling/fcose/fcose_outputs.py (FCOSEOutputs) with its own
    core/structures/pointset.py and core/layers/ml_nms.py.  detectron2.layers.batched_nms
    (torchvision's, absent) is the restatement in oracle/extreme_utils_ref.py: the NMS step is
    unpinned, the decode in front of it is the reference's own code."""
    if "fcose" in _loaded:
        return _loaded["fcose"]
    load_reference()
    _module("detectron2.structures", Instances=_Instances, Boxes=_Boxes)
    _module("detectron2.utils")
    _module("detectron2.utils.comm", get_world_size=lambda: 1)
    sys.modules["fvcore.nn"].sigmoid_focal_loss_jit = None
    _module("core.utils.comm", reduce_sum=lambda t: t)
    sys.modules["detectron2.layers"].batched_nms = extreme_utils_ref.batched_nms
    mlnms = _load("_dance_ref_ml_nms", REF + "/core/layers/ml_nms.py")
    sys.modules["core.layers"].ml_nms = mlnms.ml_nms
    pointset = _load("_dance_ref_pointset", REF + "/core/structures/pointset.py")
    saved = sys.modules["core.structures"]
    _module("core.structures", ExtremePoints=pointset.ExtremePoints, PolygonPoints=pointset.PolygonPoints)
    pkg = _module("_dance_ref_fcose_pkg")
    pkg.__path__ = []
    _module("_dance_ref_fcose_pkg.utils", get_extreme_points=None)
    try:
        _loaded["fcose"] = _load("_dance_ref_fcose_pkg.fcose_outputs", REF + "/core/modeling/fcose/fcose_outputs.py")
    finally:
        sys.modules["core.structures"] = saved
    return _loaded["fcose"]
