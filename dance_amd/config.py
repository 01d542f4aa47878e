"""The configuration keys the contour path reads, as a plain dataclass.

Defaults are the reference's (core/config/defaults.py:38-134, 140-162); `dance_r50()` /
`dsnake_r50()` apply what configs/Dance_R_50_3x.yaml and configs/Dsnake_R_50_1x.yaml
override.  `from_cfg` accepts the reference's yacs CfgNode (or any object with the same
attribute tree), so `RefineNet(cfg)` keeps the reference constructor signature.
"""
from dataclasses import dataclass, field
from typing import Tuple


@dataclass
class SnakeHeadConfig:
    convs_dim: int = 128                     # MODEL.EDGE_HEAD.CONVS_DIM (256 in shipped yamls)
    norm: str = "GN"                         # MODEL.EDGE_HEAD.NORM
    feat_dim: int = 128                      # SNAKE_HEAD.FEAT_DIM
    num_iter: Tuple[int, ...] = (0, 0, 1)    # len() = number of deformers (snake_head.py:621)
    num_convs: int = 2                       # bottom_out depth
    num_layer: Tuple[int, ...] = (8, 8, 8)
    cir_dilations: Tuple[Tuple[int, ...], ...] = ((1, 1, 1, 2, 2, 4, 4),) * 3
    dilations: Tuple[int, ...] = (1, 1)      # bottom_out conv dilations
    num_sampling: int = 128
    initial: str = "octagon"                 # {"octagon", "box"}
    de_loc_type: str = "derange"             # {"derange", "demean", "demin"}
    individual_scale: bool = False
    attention: bool = False
    detach: bool = False
    vis_path: bool = False                   # batched inference branch (snake_head.py:1892-1941)
    last_up_sample: bool = False
    conv_type: str = "ccn"
    unsupported: dict = field(default_factory=dict)

    @classmethod
    def from_cfg(cls, cfg):
        if isinstance(cfg, cls):
            return cfg
        sh = cfg.MODEL.SNAKE_HEAD
        eh = cfg.MODEL.EDGE_HEAD

        def get(node, key, default):
            return getattr(node, key, default)

        c = cls(
            convs_dim=eh.CONVS_DIM, norm=get(eh, "NORM", "GN"), feat_dim=get(sh, "FEAT_DIM", 128),
            num_iter=tuple(get(sh, "NUM_ITER", (0, 0, 1))), num_convs=get(sh, "NUM_CONVS", 2),
            num_layer=tuple(get(sh, "NUM_LAYER", (8, 8, 8))),
            cir_dilations=tuple(tuple(d) for d in get(sh, "CIR_DILATIONS", ((1, 1, 1, 2, 2, 4, 4),) * 3)),
            dilations=tuple(get(sh, "DILATIONS", (1, 1))), num_sampling=get(sh, "NUM_SAMPLING", 128),
            initial=get(sh, "INITIAL", "octagon"), de_loc_type=get(sh, "DE_LOC_TYPE", "derange"),
            individual_scale=bool(get(sh, "INDIVIDUAL_SCALE", False)),
            attention=bool(get(sh, "ATTENTION", False)), detach=bool(get(sh, "DETACH", False)),
            vis_path=bool(get(sh, "VIS_PATH", False)), last_up_sample=bool(get(sh, "LAST_UP_SAMPLE", False)),
            conv_type=get(sh, "CONV_TYPE", "ccn"),
        )
        # switches that change the inference graph and are not built (all default-off in the
        # reference and in every shipped yaml): refuse loudly rather than compute something else
        for key, off in (("ORIGINAL", False), ("PRE_OFFSET", False), ("USE_ASPP", False),
                         ("USE_PSP", False), ("MSCORE_SNAKE_ON", False), ("MARK_INDEX", False),
                         ("TRACK_PATH", False), ("SELECTIVE_REFINE", False), ("STRONGER", False),
                         ("EDGE_IN", False), ("PRED_EDGE", False), ("SKIP", False)):
            val = get(sh, key, off)
            if val != off:
                c.unsupported[key] = val
        if get(sh, "MULTI_OFFSET", 1) != 1:
            c.unsupported["MULTI_OFFSET"] = sh.MULTI_OFFSET
        if any(get(sh, "COORD_CONV", (False, False))) or any(get(sh, "USE_DEFORMABLE", (False, False))) \
                or any(get(sh, "EDGE_IN_SEPARATE", (False, False))):
            c.unsupported["COORD_CONV/USE_DEFORMABLE/EDGE_IN_SEPARATE"] = True
        return c

    def validate(self):
        if self.unsupported:
            raise NotImplementedError("SNAKE_HEAD options outside the built path: %s" % self.unsupported)
        if self.conv_type != "ccn":
            raise NotImplementedError("only CONV_TYPE 'ccn' (circular convolution) is built")
        if any(n != 8 for n in self.num_layer[:len(self.num_iter)]) or self.feat_dim != 128:
            raise NotImplementedError("the kernels implement NUM_LAYER 8 / FEAT_DIM 128 snakes")
        if self.convs_dim % 64 != 0:
            raise NotImplementedError("EDGE_HEAD.CONVS_DIM must be a multiple of 64")


def dance_r50(**over):
    """configs/Dance_R_50_3x.yaml (and Dance_R_101_3x.yaml: same head keys)."""
    kw = dict(convs_dim=256, num_sampling=196, detach=True, initial="box", attention=True,
              individual_scale=True)
    kw.update(over)
    return SnakeHeadConfig(**kw)


def dsnake_r50(**over):
    """configs/Dsnake_R_50_1x.yaml."""
    kw = dict(convs_dim=256, num_sampling=128)
    kw.update(over)
    return SnakeHeadConfig(**kw)
