"""Drop-in snake-head modules with the reference's constructor / forward signature and
state-dict keys, running the contour path on the gfx950 kernels.

    RefineNet(cfg)   <- core/modeling/edge_snake/snake_head.py:433  (DANCE)
    DsnakeHead(cfg)  <- core/modeling/dsnake_baseline/dsnake_head.py:197  (deep-snake variant)

    forward(features, pred_instances=None, targets=None) -> (list[Instances], {})

`features` is what the reference's callers pass (edge_det.py:387 / dsnake_head.py:193): the
stride-4 map (B, [1+]C, Hf, Wf), channel 0 being the attention map when ATTENTION is on.
`pred_instances` only need `.pred_boxes.tensor`, `.image_size`, `.has(name)`,
`.ext_points.tensor` and attribute assignment (detectron2 `Instances` do; no detectron2
import happens here).  Parameters live in ordinary nn.Modules so that
`load_state_dict` of a reference checkpoint works key for key; `bottom_out` (2 x conv3x3
+ GroupNorm + ReLU, snake_head.py:562-615) stays torch.nn and is run channels_last so its
output already is the NHWC layout the gather wants.

Inference only: the training branch (target matching with Shapely/NumPy on the CPU,
snake_head.py:982-1384) is outside the built path and raises.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, timer
from ._lib import InitFuse16Struct
from .config import SnakeHeadConfig
from .snake import ContourDeformer, SnakeNetWeights, sub_state_dict
from .structures import ExtremePoints, PolygonPoints


# ----------------------------------------------------------------- parameter containers
class DilatedCircularConv(nn.Module):
    """Parameter holder for snake_head.py:92-116 (`fc` = Conv1d k=9)."""

    def __init__(self, state_dim, out_state_dim=None, n_adj=4, dilation=1):
        super().__init__()
        self.n_adj, self.dilation = n_adj, dilation
        out_state_dim = state_dim if out_state_dim is None else out_state_dim
        self.fc = nn.Conv1d(state_dim, out_state_dim, kernel_size=n_adj * 2 + 1, dilation=dilation)


class SnakeBlock(nn.Module):
    """Parameter holder for snake_head.py:140-153 (conv -> ReLU -> BatchNorm1d)."""

    def __init__(self, state_dim, out_state_dim, n_adj=4, dilation=1):
        super().__init__()
        self.conv = DilatedCircularConv(state_dim, out_state_dim, n_adj, dilation)
        self.relu = nn.ReLU(inplace=True)
        self.norm = nn.BatchNorm1d(out_state_dim)


class _SnakeNet(nn.Module):
    """Parameter holder for snake_head.py:220-258 / dsnake_head.py:95-117."""

    def __init__(self, state_dim, feature_dim, dilations):
        super().__init__()
        self.head = SnakeBlock(feature_dim, state_dim)
        self.res_layer_num = len(dilations)
        self.dilations = tuple(dilations)
        for i, d in enumerate(dilations):
            setattr(self, "res" + str(i), SnakeBlock(state_dim, state_dim, n_adj=4, dilation=d))
        fusion_state_dim = 256
        self.fusion = nn.Conv1d(state_dim * (self.res_layer_num + 1), fusion_state_dim, 1)
        self.prediction = nn.Sequential(
            nn.Conv1d(state_dim * (self.res_layer_num + 1) + fusion_state_dim, 256, 1),
            nn.ReLU(inplace=True),
            nn.Conv1d(256, 64, 1),
            nn.ReLU(inplace=True),
            nn.Conv1d(64, 2, 1),
        )


class _ConvNormAct(nn.Conv2d):
    """conv3x3 (+ GroupNorm) + ReLU with detectron2.layers.Conv2d's state-dict keys
    (`weight`, `norm.weight`, `norm.bias`).  In eval mode on the GPU the shipped shape
    (C -> C, C % 128 == 0, GroupNorm(32), no bias, dilation 1) runs on the feature-prep kernels
    (conv3x3 as split-half f16 MFMA implicit GEMM + GroupNorm + ReLU; output in
    channels_last memory); anything else, `DANCE_FEATPREP=torch`, or `use_hip = False` runs
    the stock torch.nn ops."""

    use_hip = True

    def __init__(self, cin, cout, dilation, norm):
        super().__init__(cin, cout, kernel_size=3, stride=1, padding=1, dilation=dilation,
                         bias=not norm)
        self.norm = nn.GroupNorm(32, cout) if norm == "GN" else None

    def hip_layer(self, dev):
        """(w16, scale, gamma, beta) for ops.bottom_out16, or None when the HIP kernels do not
        cover this layer."""
        if (self.norm is None or self.bias is not None or self.dilation != (1, 1)
                or self.in_channels != self.out_channels or self.in_channels % 128 != 0
                or self.norm.num_groups != 32):
            return None
        sig = (dev,) + tuple(int(p._version) for p in self.parameters())
        if getattr(self, "_hip_sig", None) != sig:
            w16, s = ops.pack_conv3x3_weight16(self.weight.detach().float())
            self._hip_packed = (w16, s, self.norm.weight.detach().float().contiguous(),
                                self.norm.bias.detach().float().contiguous())
            self._hip_sig = sig
        return self._hip_packed

    def forward(self, x):
        if (self.use_hip and x.is_cuda and not self.training and not torch.is_grad_enabled()
                and os.environ.get("DANCE_FEATPREP", "hip") == "hip"
                and os.environ.get("DANCE_PRECISION", "fp16x3") == "fp16x3"):
            layer = self.hip_layer(x.device)
            if layer is not None:
                return ops.bottom_out16(x, 0, [layer], groups=32, eps=self.norm.eps, measure=False).permute(0, 3, 1, 2)
        x = super().forward(x)
        if self.norm is not None:
            x = self.norm(x)
        return F.relu(x)


class _AttConv(nn.Conv2d):
    """Parameter holder with detectron2.layers.Conv2d's keys (`weight`, [`bias`], `norm.*`)."""

    def __init__(self, cin, cout, norm):
        super().__init__(cin, cout, kernel_size=3, stride=1, padding=1, bias=not norm)
        self.norm = nn.GroupNorm(4, cout) if norm else None


class Attender(nn.Sequential):
    """Drop-in for `EdgeSnakeFPNHead.attender` (edge_det.py:66-100): same state-dict keys
    (`0.weight`, `0.norm.weight`, ..., `2.bias`), `forward(1 - pred_edge) -> att_map`
    (edge_det.py:202-204), run as three HIP launches (csrc/attender.hip).  GPU only."""

    def __init__(self):
        super().__init__(_AttConv(1, 32, True), _AttConv(32, 32, True), _AttConv(32, 1, False))
        # edge_det.py:96-99 (c2_msra_fill = kaiming_normal_(fan_out, relu); the last conv N(0, 0.01), bias 0)
        for m in (self[0], self[1]):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        nn.init.normal_(self[2].weight, 0, 0.01)
        nn.init.constant_(self[2].bias, 0)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("dance_amd.modules.Attender runs on a GPU only; there is no CPU fallback")
        sig = (x.device,) + tuple(int(p._version) for p in self.parameters())
        if getattr(self, "_att_sig", None) != sig:
            f = lambda t: t.detach().float().contiguous()
            self._att_w = (f(self[0].weight), f(self[0].norm.weight), f(self[0].norm.bias),
                           f(self[1].weight.detach().permute(1, 2, 3, 0)), f(self[1].norm.weight),
                           f(self[1].norm.bias), f(self[2].weight), f(self[2].bias))
            self._att_sig = sig
        return ops.attender(x, *self._att_w, eps=self[0].norm.eps)


def _reference_init(module):
    # snake_head.py:632-641 / dsnake_head.py:255-259
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv1d)):
            m.weight.data.normal_(0.0, 0.01)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


def _image_index(tensors):
    """(sum of len(t),) image number of every row, in the tensors' dtype: one arange and one cat instead of
    a fill launch per image (the reference: `b.new_empty(len(b)).fill_(i)` per image)."""
    ar = torch.arange(len(tensors), dtype=tensors[0].dtype, device=tensors[0].device)
    return torch.cat([ar[i:i + 1].expand(len(t)) for i, t in enumerate(tensors)])


def _image_slices(counts):
    """contiguous per-image ranges of a batch laid out image by image"""
    out, start = [], 0
    for c in counts:
        out.append(slice(start, start + int(c)))
        start += int(c)
    return out


class _SnakeHeadBase(nn.Module):
    def _pack(self, names):
        """(Re)pack the deformers for the kernels when parameters changed or moved."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("dance_amd modules run on a GPU only (module is on %s); "
                               "there is no CPU fallback" % dev)
        sig = (dev,) + tuple(int(p._version) for p in self.parameters()) + \
              tuple(int(b._version) for b in self.buffers())
        if getattr(self, "_packed_sig", None) != sig:
            sd = {k: v for k, v in self.state_dict().items()}
            self._packed = {n: SnakeNetWeights(sub_state_dict(sd, n + "."), dev,
                                               getattr(self, n).dilations) for n in names}
            self._packed_sig = sig
        return self._packed

    def _bottom_hip_layers(self, dev):
        """bottom_out packed for the HIP feature-prep kernels, or None when this configuration
        is outside what they cover (then the stock torch.nn modules run)."""
        if os.environ.get("DANCE_FEATPREP", "hip") != "hip" or os.environ.get("DANCE_PRECISION", "fp16x3") != "fp16x3":
            return None
        layers = [m.hip_layer(dev) for m in self.bottom_out]
        return None if any(l is None for l in layers) else layers

    def _bottom(self, feats_all, c_off):
        """bottom_out (snake_head.py:1816-1820) on channels [c_off:] of the NCHW input; returns
        NHWC (B,Hf,Wf,C) contiguous memory.  HIP path (conv3x3 + GroupNorm + ReLU in the
        fast-mode arithmetic, writes NHWC directly); the stock modules in channels_last
        otherwise."""
        if len(self.bottom_out) == 0:
            return None
        layers = self._bottom_hip_layers(feats_all.device)
        if layers is not None:
            return ops.bottom_out16(feats_all, c_off, layers, groups=32, eps=self.bottom_out[0].norm.eps)
        x = feats_all[:, c_off:].contiguous(memory_format=torch.channels_last)
        for conv in self.bottom_out:
            x = conv(x)
        x = x.permute(0, 2, 3, 1)
        return x if x.is_contiguous() else x.contiguous()


# ------------------------------------------------------------------------------ DANCE
class RefineNet(_SnakeHeadBase):
    def __init__(self, cfg, polygon_cls=PolygonPoints):
        super().__init__()
        c = SnakeHeadConfig.from_cfg(cfg)
        c.validate()
        self.cfg = c
        self.polygon_cls = polygon_cls
        self.initial = c.initial
        self.num_iter = c.num_iter
        self.num_convs = c.num_convs
        self.num_sampling = c.num_sampling
        self.de_location_type = c.de_loc_type
        self.individual_scale = c.individual_scale
        self.attention = c.attention
        self.visualize_path = c.vis_path
        self.bottom_out = nn.ModuleList(
            [_ConvNormAct(c.convs_dim, c.convs_dim, c.dilations[i], c.norm) for i in range(c.num_convs)])
        for i in range(len(c.num_iter)):
            setattr(self, "deformer" + str(i), _SnakeNet(c.feat_dim, c.convs_dim + 2, c.cir_dilations[i]))
        _reference_init(self)
        self._deformer = None

    # -- initial contours ----------------------------------------------------------
    def uniform_upsample(self, poly, p_num):
        """snake_head.py:764-794 on (1,N,E,2) -> ((N,p,2), edge_start_idx (N,E))."""
        if poly.size(1) == 0:
            return torch.zeros([0, p_num, 2], device=poly.device), None
        return ops.uniform_upsample(poly, p_num)

    def get_octagon(self, ex, p_num):
        """snake_head.py:796-832."""
        if len(ex) == 0:
            return torch.zeros([0, p_num, 2], device=ex.device), None
        return ops.uniform_upsample(ex, p_num, kind=ops.POLY_OCTAGON)

    def _sample_boxes(self, pred_instances):
        """sample_bboxes_fast / single_sample_bboxes_fast (snake_head.py:865-886, 911-924):
        every image's boxes resampled in one launch."""
        boxes = [x.pred_boxes.tensor for x in pred_instances]
        allb = torch.cat(boxes, dim=0) if len(boxes) > 1 else boxes[0]
        locs, _ = ops.uniform_upsample(allb, self.num_sampling, kind=ops.POLY_BOX) if len(allb) else \
            (torch.zeros([0, self.num_sampling, 2], device=allb.device), None)
        return locs, _image_index(boxes)

    def _sample_octagons(self, pred_instances):
        """sample_octagons_fast (snake_head.py:844-863), incl. the in-place fit_to_box /
        clamp of the instances' extreme points."""
        exts, idx = [], []
        for i, inst in enumerate(pred_instances):
            if inst.has("ext_points"):
                h, w = inst.image_size
                inst.ext_points.fit_to_box()
                ext = inst.ext_points.tensor
                ext[..., 0].clamp_(min=0, max=w - 1)
                ext[..., 1].clamp_(min=0, max=h - 1)
                exts.append(ext)
                idx.append(ext.new_empty(len(ext)).fill_(i))
        if not exts:
            return exts, idx
        # every image's octagons in ONE resampler launch (and one status assert) instead of the
        # reference's launch and synchronising assert per image; contours are independent
        return self.get_octagon(torch.cat(exts, dim=0), self.num_sampling)[0], torch.cat(idx)

    # -- the path ------------------------------------------------------------------
    def _get_deformer(self):
        names = ["deformer%d" % i for i in range(len(self.num_iter))]
        packed = self._pack(names)
        if self._deformer is None or self._deformer.nets[0] is not packed[names[0]]:
            self._deformer = ContourDeformer([packed[n] for n in names], padding="reflection",
                                             de_loc_type=self.de_location_type,
                                             individual_scale=self.individual_scale,
                                             attention=self.attention)
        return self._deformer

    def _features(self, features):
        """Attention split (snake_head.py:1812-1814) + bottom_out (1816-1820) ->
        (feat NHWC, att plane); the re-concatenation of :1824 is never materialised."""
        d = self._get_deformer()
        if isinstance(features, (tuple, list)):
            # (att_map (B,1,Hf,Wf), x (B,C,Hf,Wf)) from EdgeSnakeFPNHead below: the reference's
            # torch.cat([att_map, x]) (edge_det.py:313-315) and this split cancel out
            att, feats, c_off = features[0][:, 0].contiguous(), features[1], 0
            features = feats
        elif self.attention:
            att, feats, c_off = features[:, 0].contiguous(), features[:, 1:], 1
        else:
            att, feats, c_off = None, features, 0
        nhwc = self._bottom(features, c_off)
        if nhwc is None:
            nhwc, _ = ops.nchw_to_nhwc(feats.contiguous(), 0)
        return d, nhwc.float(), att

    def deform(self, features, locations, image_idx, image_sizes, whs):
        """The `snake_deformation` scope (snake_head.py:1822-1862): per-iteration contours."""
        with timer.env("snake_feat_prep"):
            d, nhwc, att = self._features(features)
        # a batch that is not whole rounds of workgroups: deform() itself peels the remainder off to a
        # side stream (279.0k vs 276.0k contours/s with an even split at 1600, same box).  Whole rounds,
        # from 1024 contours on: halves on two HIP streams (one half's kernel tails and launch gaps overlap
        # the other half's kernels, +0.3..0.5 %).  Bit-identical either way, never under stream capture;
        # DANCE_STREAMS=<k> fixes the count
        env = os.environ.get("DANCE_STREAMS")
        if env:
            streams = int(env)
        elif torch.cuda.is_current_stream_capturing() or len(locations) < 1024 or \
                (image_idx is not None and d.peel_count(len(locations), locations.device)):
            streams = 1
        else:
            streams = 2
        wait_for = getattr(self, "_wait_for", None)
        if wait_for is not None:                 # the attention map is being produced on another stream (EdgeSnakeFPNHead)
            torch.cuda.current_stream(nhwc.device).wait_stream(wait_for)
        with timer.env("snake_deformation"):
            return d.deform((nhwc, att), locations, image_idx, image_sizes, whs, layout="nhwc", streams=streams)

    def single_test(self, features, pred_instances):
        """snake_head.py:1763-1869 (one image)."""
        with timer.env("contour_sampling"):                    # snake_head.py:1786
            if self.initial == "octagon":
                locations, _ = self._sample_octagons(pred_instances)
            elif self.initial == "box":
                locations, _ = self._sample_boxes(pred_instances[:1])
            else:
                raise ValueError("Invalid initial!")
        if len(locations) == 0:
            return pred_instances, {}
        image_sizes = pred_instances[0].image_size
        pred_instances[0].pred_boxes.clip(image_sizes)
        bboxes = pred_instances[0].pred_boxes.tensor
        whs = torch.stack([bboxes[:, 2] - bboxes[:, 0], bboxes[:, 3] - bboxes[:, 1]], dim=1)
        preds = self.deform(features, locations, None, image_sizes, whs)
        with timer.env("snake_postprocess"):                   # snake_head.py:1864
            inst = pred_instances[0]
            inst.pred_polys = self.polygon_cls(preds[-1])
        return [inst], {}

    def forward(self, features, pred_instances=None, targets=None):
        """snake_head.py:1871-2385, inference branches."""
        if self.training:
            raise NotImplementedError("dance_amd.RefineNet is inference-only (call .eval())")
        if not self.visualize_path:
            return self.single_test(features, pred_instances)
        assert pred_instances is not None
        if self.initial == "octagon":
            locations, image_idx = self._sample_octagons(pred_instances)
        elif self.initial == "box":
            locations, image_idx = self._sample_boxes(pred_instances)
        else:
            raise ValueError("Invalid initial!")
        if len(locations) == 0:
            return pred_instances, {}
        image_sizes = [x.image_size for x in pred_instances]
        bboxes = torch.cat([x.pred_boxes.tensor for x in pred_instances], dim=0)
        whs = torch.stack([bboxes[:, 2] - bboxes[:, 0], bboxes[:, 3] - bboxes[:, 1]], dim=1)
        preds = self.deform(features, locations, image_idx, image_sizes, whs)
        # predict_postprocess, snake_head.py:2399-2445
        results = []
        # contours are laid out image by image (the samplers concatenate per image): plain
        # slices instead of the reference's `image_idx == i` masks, which cost a device
        # synchronisation per image
        if self.initial == "octagon":
            counts = [len(x.ext_points.tensor) if x.has("ext_points") else 0 for x in pred_instances]
        else:
            counts = [len(x.pred_boxes.tensor) for x in pred_instances]
        for inst, sel in zip(pred_instances, _image_slices(counts)):
            inst.pred_polys = self.polygon_cls(preds[-1][sel])
            path = [locations[sel]] + [p[sel] for p in preds if p.size(1) == self.num_sampling]
            inst.pred_path = torch.stack(path, dim=1)
            results.append(inst)
        return results, {}


# ------------------------------------------------------------------------------ Dsnake
class DsnakeHead(_SnakeHeadBase):
    def __init__(self, cfg, polygon_cls=PolygonPoints, extreme_cls=ExtremePoints):
        super().__init__()
        c = SnakeHeadConfig.from_cfg(cfg)
        c.validate()
        self.cfg = c
        self.polygon_cls, self.extreme_cls = polygon_cls, extreme_cls
        self.num_iter = c.num_iter
        self.num_convs = c.num_convs
        self.num_sampling = c.num_sampling
        self.visualize_path = c.vis_path
        dil = (1, 1, 1, 2, 2, 4, 4)                                   # dsnake_head.py:102
        self.bottom_out = nn.ModuleList(
            [_ConvNormAct(c.convs_dim, c.convs_dim, c.dilations[i], c.norm) for i in range(c.num_convs)])
        self.fuse = nn.Conv1d(2 * c.convs_dim, c.convs_dim, 1)
        self.init_snake = _SnakeNet(128, c.convs_dim + 2, dil)
        for i in range(len(c.num_iter)):
            setattr(self, "deformer" + str(i), _SnakeNet(128, c.convs_dim + 2, dil))
        _reference_init(self)
        self._deformer = None
        self._fuse_sig = None

    def _get_deformer(self):
        names = ["init_snake"] + ["deformer%d" % i for i in range(len(self.num_iter))]
        packed = self._pack(names)
        if self._deformer is None or self._deformer.nets[0] is not packed[names[0]]:
            self._deformer = ContourDeformer([packed[n] for n in names], padding="zeros",
                                             de_loc_type="demin", individual_scale=False,
                                             attention=False)
            w = self.fuse.weight.detach().float()                     # (C, 2C, 1)
            c = w.shape[0]
            self._fuse_w_sampled = ops.pack_conv_weight(w[:, :c, :])
            self._fuse_wg_center = w[:, c:, 0].t().contiguous()       # [g][o]
            self._fuse_b = self.fuse.bias.detach().float().contiguous()
            # the same Conv1d for the one-launch init stage of the fast precision mode (init16.hip)
            self._fuse_w16, sc = ops.pack_conv_weight16(w[:, :c, :])
            f = InitFuse16Struct()
            f.w16, f.wg_center, f.bias = (self._fuse_w16.data_ptr(), self._fuse_wg_center.data_ptr(),
                                          self._fuse_b.data_ptr())
            f.w_inv_scale = 1.0 / sc
            self._fuse16 = f
        return self._deformer

    def init_stage(self, d, feat, init_locations, idx):
        """DsnakeHead.init (dsnake_head.py:718-746): 40-point quadrangle + centre feature
        -> fuse Conv1d -> init snake -> 4 extreme points."""
        return self.init_contours(d, feat, init_locations, idx)[:, ::10]

    def init_contours(self, d, feat, init_locations, idx, one_launch=None):
        """The init snake's deformed ring (dsnake_head.py:718-744).  one_launch (default: whenever
        the deformer supports it; `self.one_launch_init = False` switches it off): gather + centre feature + fuse +
        de_location + head block as one kernel of the fast precision mode (csrc/init16.hip);
        otherwise separate launches with `fuse` as an exact fp32 convolution."""
        c = feat.shape[-1]
        if one_launch is None:
            one_launch = getattr(self, "one_launch_init", True)
        if one_launch and d.supports_init_evolve(c, init_locations.shape[1]):
            return d.init_evolve(0, feat, init_locations, idx, self._fuse16)
        sampled, _ = ops.gather_features(feat, None, init_locations, idx, ops.PAD_ZEROS, "demin")
        center = (torch.min(init_locations, dim=1)[0] + torch.max(init_locations, dim=1)[0]) * 0.5
        cfeat, _ = ops.gather_features(feat, None, center[:, None].contiguous(), idx, ops.PAD_ZEROS, None)
        # centre half of `fuse` is vertex-invariant: fold it into a per-contour row constant
        rowinit = ops.row_constant(cfeat[:, :, 0], self._fuse_wg_center, self._fuse_b)
        n, _, v = sampled.shape
        x = torch.empty((n, c + 2, v), dtype=torch.float32, device=feat.device)
        ops.circconv(sampled, self._fuse_w_sampled, None, cin=c, rowinit=rowinit,
                     out=x[:, :c])
        x[:, c:] = sampled[:, c:]
        return d.evolve(0, feat, None, init_locations, idx, None, use_att=False, clip_first=False,
                        clip_max=(0.0, 0.0), x_in=x)

    @staticmethod
    def _side_deformer(d):
        """A deformer of its own workspace sharing d's packed nets (one per HIP stream)."""
        side = ContourDeformer(d.nets, padding="zeros", de_loc_type="demin", individual_scale=False,
                               attention=False, precision=d.precision)
        side.nets = d.nets
        side.policy = d.policy
        return side

    def _chain(self, d, nhwc, init_locations, idx, gt_exts, gt_mask, deferred):
        """init -> clamped extreme points (GT ones where given) -> octagon -> the evolve iterations
        (dsnake_head.py:791-830) for a set of contours; no host synchronisation inside (the resampler's
        status goes to `deferred`)."""
        pred_exts = self.init_stage(d, nhwc, init_locations, idx)
        h, w = nhwc.shape[1] * 4, nhwc.shape[2] * 4
        # clamp to the image (:800-805), GT points where given (:811-817) and get_octagon (:808-810)
        # in one library launch: no torch elementwise kernel runs inside a forked stream
        locations, exts, _ = ops.octagon_upsample(pred_exts, self.num_sampling, clamp_max=(w - 1.0, h - 1.0),
                                                  override=gt_exts if gt_mask is not None else None,
                                                  override_mask=gt_mask, deferred=deferred)
        preds, cur = [], locations
        for i in range(len(self.num_iter)):
            cur = d.evolve(1 + i, nhwc, None, cur, idx, None, use_att=False, clip_first=False,
                           clip_max=(0.0, 0.0))
            preds.append(cur)
        return pred_exts, exts, locations, preds

    def forward(self, features, pred_instances=None, targets=None):
        """dsnake_head.py:767-855, inference branch."""
        if self.training:
            raise NotImplementedError("dance_amd.DsnakeHead is inference-only (call .eval())")
        assert pred_instances is not None
        boxes = [x.pred_boxes.tensor for x in pred_instances]
        allb = torch.cat(boxes, dim=0)
        if len(allb) == 0:
            return pred_instances, {}
        deferred = []                                # resampler status tensors: one assert (one sync) at the end
        init_locations, _ = ops.uniform_upsample(allb, 40, kind=ops.POLY_QUAD, deferred=deferred)   # :438-451
        image_idx = _image_index(boxes)
        idx = image_idx.to(torch.int32)
        d = self._get_deformer()
        nhwc = self._bottom(features, 0)
        if nhwc is None:
            nhwc, _ = ops.nchw_to_nhwc(features.contiguous(), 0)
        nhwc = nhwc.float()
        slices = _image_slices(len(b) for b in boxes)
        gt_exts = gt_mask = None                     # GT extreme points given (:811-817)
        if any(inst.has("ext_points") for inst in pred_instances):
            gt_exts = torch.zeros((len(allb), 4, 2), dtype=torch.float32, device=allb.device)
            gt_mask = torch.zeros(len(allb), dtype=torch.bool, device=allb.device)
            for inst, sel in zip(pred_instances, slices):
                if inst.has("ext_points"):
                    gt_exts[sel] = inst.ext_points.tensor
                    gt_mask[sel] = True
        # every contour is independent: DANCE_STREAMS=<k> runs the whole chain for k parts of the batch on
        # k HIP streams (bit-identical).  Not the default here: measured 14.97 vs 15.08 ms per forward
        # at 32 x 100 (+0.7 %), and slower with NUM_CONVS = 2 (the feature prep stays on one stream)
        k = int(os.environ.get("DANCE_STREAMS", "1"))
        if k <= 1:
            pred_exts, exts, locations, preds = self._chain(d, nhwc, init_locations, idx, gt_exts, gt_mask, deferred)
        else:
            cur = torch.cuda.current_stream(allb.device)
            if getattr(self, "_side", None) is None or len(self._side) != k or self._side[0][1].nets is not d.nets:
                self._side = [(torch.cuda.Stream(allb.device), self._side_deformer(d)) for _ in range(k)]
            n = len(allb)
            cuts = [n * i // k for i in range(k + 1)]
            parts = []
            for i, (st, di) in enumerate(self._side):
                sl = slice(cuts[i], cuts[i + 1])
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    parts.append(self._chain(di, nhwc, init_locations[sl], idx[sl],
                                             None if gt_exts is None else gt_exts[sl],
                                             None if gt_mask is None else gt_mask[sl], deferred))
            for st, _ in self._side:
                cur.wait_stream(st)
            for prt in parts:
                for t in list(prt[:3]) + list(prt[3]):
                    t.record_stream(cur)
            for t in deferred:
                t.record_stream(cur)
            pred_exts, exts, locations = (torch.cat([prt[i] for prt in parts]) for i in range(3))
            preds = [torch.cat([prt[3][i] for prt in parts]) for i in range(len(self.num_iter))]
        ops.check_upsample_status(deferred)
        for inst, sel in zip(pred_instances, slices):
            if not inst.has("ext_points"):
                inst.ext_points = self.extreme_cls(exts[sel])
        results = []
        for inst, sel in zip(pred_instances, slices):                             # :857-877
            inst.pred_polys = self.polygon_cls(preds[-1][sel])
            if self.visualize_path:
                path = [locations[sel]] + [p[sel] for p in preds[:-1]]
                inst.pred_path = torch.stack(path, dim=1)
            results.append(inst)
        self.last_stages = {"init": init_locations, "exts_raw": pred_exts, "exts": exts,
                            "octagons": locations, "preds": preds}
        return results, {}


# ------------------------------------------------------------- the heads' callers (L5)
class EdgeSnakeFPNHead(nn.Module):
    """Drop-in for the reference's `EdgeSnakeFPNHead` (core/modeling/edge_snake/edge_det.py:35-395),
    inference path of the shipped configurations (EDGE_HEAD.TRAIN, STRONG_FEAT, SNAKE_HEAD.ATTENTION):

        x = sum_f scale_heads[f](features[f])                 edge_det.py:188-194   (stock torch.nn)
        pred_edge = sigmoid(predictor(x))                     :197-199              (stock torch.nn)
        att_map = attender(1 - pred_edge)                     :200-204              (HIP, Attender)
        refine_head(cat([att_map, x]), pred_instances, None)  :313-315, 387         (HIP, RefineNet)
        return att_map, {}, new_instances                     :392-395

    Same constructor `(cfg, input_shape)` and state-dict keys (`attender.*`, `p2.0.*` ... `p5.4.*`,
    `predictor.*`, `refine_head.*`).  The semantic-FPN head stays torch.nn modules (SURVEY.md
    section 2: boundary / caller) -- its 256 -> 256 conv3x3 + GroupNorm + ReLU layers route to the
    feature-prep kernels in eval mode (_ConvNormAct), the first layer of each level, the upsampling,
    the sum and the 1x1 predictor are stock ops; the concatenation is not materialised."""

    def __init__(self, cfg, input_shape, polygon_cls=PolygonPoints):
        super().__init__()
        eh, sh = cfg.MODEL.EDGE_HEAD, cfg.MODEL.SNAKE_HEAD
        self.in_features = list(eh.IN_FEATURES)
        self.common_stride = eh.COMMON_STRIDE
        self.attention = bool(getattr(sh, "ATTENTION", False))
        self.strong_feat = bool(getattr(eh, "STRONG_FEAT", False))
        self.edge_on = bool(getattr(eh, "TRAIN", True))
        conv_dims, norm = eh.CONVS_DIM, getattr(eh, "NORM", "GN")
        gt_in = getattr(getattr(cfg, "TEST", None), "GT_IN", None)
        refused = {k: v for k, v in (("SNAKE_HEAD.EDGE_IN", getattr(sh, "EDGE_IN", False)),
                                     ("SNAKE_HEAD.SELECTIVE_REFINE", getattr(sh, "SELECTIVE_REFINE", False)),
                                     ("TEST.GT_IN.ON", bool(gt_in is not None and getattr(gt_in, "ON", False))),
                                     ("EDGE_HEAD.TRAIN off", not self.edge_on)) if v}
        if refused:
            raise NotImplementedError("EdgeSnakeFPNHead options outside the built path: %s" % refused)
        import math
        if self.attention:                                                  # edge_det.py:66-100
            self.attender = Attender()
        self.scale_heads = []
        for f in self.in_features:                                          # edge_det.py:106-138
            stride, cin = input_shape[f].stride, input_shape[f].channels
            ops_ = []
            for k in range(max(1, int(math.log2(stride) - math.log2(self.common_stride)))):
                ops_.append(_ConvNormAct(cin if k == 0 else conv_dims, conv_dims, 1, norm if norm == "GN" else ""))
                if stride != self.common_stride:
                    ops_.append(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False))
            self.scale_heads.append(nn.Sequential(*ops_))
            self.add_module(f, self.scale_heads[-1])
        num_classes = eh.NUM_CLASSES
        if not self.strong_feat:                                            # edge_det.py:140-163
            self.predictor = nn.Conv2d(conv_dims, num_classes, kernel_size=1)
        else:
            self.predictor = nn.Sequential(_ConvNormAct(conv_dims, conv_dims, 1, norm if norm == "GN" else ""),
                                           nn.Conv2d(conv_dims, num_classes, kernel_size=1))
        self.refine_head = RefineNet(cfg, polygon_cls=polygon_cls)

    @staticmethod
    def _hip_upsample_ok():
        return (_ConvNormAct.use_hip and not torch.is_grad_enabled()
                and os.environ.get("DANCE_FEATPREP", "hip") == "hip")

    @classmethod
    def _scale_head_front(cls, head, feat):
        """A scale head up to, not including, an x2 upsampling it ends in (edge_det.py:106-138) ->
        (y, the pending nn.Upsample or None)."""
        mods = list(head)
        if isinstance(mods[-1], nn.Upsample) and feat.is_cuda and cls._hip_upsample_ok():
            y = feat
            for m in mods[:-1]:
                y = m(y)
            return y, mods[-1]
        return head(feat), None

    @staticmethod
    def _scale_head_join(x, y, up):
        """x + up(y) (edge_det.py:188-194).  When both sides are channels_last GPU tensors (the HIP conv
        layers' output) the pending upsampling and the sum are one launch, written into x."""
        if up is None:
            return y if x is None else x + y
        if (x is not None and ops.is_nhwc(y) and ops.is_nhwc(x) and y.size(1) % 4 == 0
                and tuple(x.shape) == (y.size(0), y.size(1), 2 * y.size(2), 2 * y.size(3))):
            return ops.upsample2x_add(y, x, inplace=True)
        return up(y) if x is None else x + up(y)

    def _predict(self, x):
        """predictor(x) (edge_det.py:140-163, 197).  Its last layer, a 1x1 convolution to NUM_CLASSES
        channels, runs as a fixed-order HIP kernel when its input is a channels_last GPU tensor (the HIP
        conv layers' output): the stock kernel's split-K atomics made two forwards differ in the last bit."""
        last = self.predictor[-1] if self.strong_feat else self.predictor
        y = self.predictor[0](x) if self.strong_feat else x
        if (_ConvNormAct.use_hip and not torch.is_grad_enabled() and ops.is_nhwc(y) and y.size(1) % 4 == 0
                and os.environ.get("DANCE_FEATPREP", "hip") == "hip"):
            return ops.conv1x1_nhwc(y, last.weight, last.bias)
        return last(y)

    # Few images (the reference's inference mode is ONE image per GPU, dance.py:52): the kernels of the
    # low-resolution pyramid levels (250, 78 and 28 workgroups on 256 CUs) and of the edge / attention
    # branch leave most of the chip idle, and nothing orders them against the full-resolution convolutions
    # they run beside in the data-flow graph -- so they go to HIP streams of their own:
    #   side 0..k   scale heads of p3.. up to their last upsampling    ||  main: scale head of p2
    #   main        the x2 upsample + sum of every level (in the reference's order: same bits)
    #   side edge   predictor -> sigmoid -> attender                   ||  main: refine_head's bottom_out
    #   main        waits for the attention map right before the first gather.
    # Under stream capture (dance_amd/graph.py) the forks become parallel branches of the HIP graph.
    OVERLAP_MAX_IMAGES = 2

    def _side_streams(self, dev, k):
        cache = getattr(self, "_streams", None)
        if cache is None or cache[0] != dev or len(cache[1]) < k:
            self._streams = cache = (dev, [torch.cuda.Stream(dev) for _ in range(k)])
        return cache[1][:k]

    def forward(self, features, pred_instances=None, targets=None):
        if self.training:
            raise NotImplementedError("dance_amd.EdgeSnakeFPNHead is inference-only (call .eval())")
        first = features[self.in_features[0]]
        overlap = (first.is_cuda and first.size(0) <= self.OVERLAP_MAX_IMAGES and self._hip_upsample_ok()
                   and os.environ.get("DANCE_STREAMS", "") != "1")
        if not overlap:
            with timer.env("pfpn_back"):
                x = None
                for f, head in zip(self.in_features, self.scale_heads):
                    x = self._scale_head_join(x, *self._scale_head_front(head, features[f]))
            with timer.env("edge"):
                pred_edge = self._predict(x).sigmoid()
                att_map = self.attender(1 - pred_edge) if self.attention else None
            feats = x if self.strong_feat else features["p2"]
            snake_input = (att_map, feats) if self.attention else feats
            new_instances, _ = self.refine_head(snake_input, pred_instances, None)
            return (att_map if self.attention else pred_edge), {}, new_instances
        cur = torch.cuda.current_stream(first.device)
        sides = self._side_streams(first.device, len(self.in_features))
        with timer.env("pfpn_back"):
            fronts = []
            for st, f, head in zip(sides[1:], self.in_features[1:], self.scale_heads[1:]):
                st.wait_stream(cur)                                 # (the inputs exist; the previous forward is done with this stream's buffers)
                with torch.cuda.stream(st):
                    fronts.append(self._scale_head_front(head, features[f]))
            x = self._scale_head_join(None, *self._scale_head_front(self.scale_heads[0], first))
            for st, (y, up) in zip(sides[1:], fronts):
                cur.wait_stream(st)
                x = self._scale_head_join(x, y, up)
        edge = sides[0]
        edge.wait_stream(cur)
        with timer.env("edge"), torch.cuda.stream(edge):
            pred_edge = self._predict(x).sigmoid()
            att_map = self.attender(1 - pred_edge) if self.attention else None
        feats = x if self.strong_feat else features["p2"]
        snake_input = (att_map, feats) if self.attention else feats
        self.refine_head._wait_for = edge                        # joined right before the deformation reads the map
        try:
            new_instances, _ = self.refine_head(snake_input, pred_instances, None)
        finally:
            self.refine_head._wait_for = None
            cur.wait_stream(edge)                                # (also when refine_head returned early: no instances)
        return (att_map if self.attention else pred_edge), {}, new_instances


class SnakeFPNHead(nn.Module):
    """Drop-in for the deep-snake baseline's caller (dsnake_head.py:182-194)."""

    def __init__(self, cfg, input_shape=None, polygon_cls=PolygonPoints, extreme_cls=ExtremePoints):
        super().__init__()
        self.refine_head = DsnakeHead(cfg, polygon_cls=polygon_cls, extreme_cls=extreme_cls)

    def forward(self, features, pred_instances=None, targets=None):
        if self.training:
            raise NotImplementedError("dance_amd.SnakeFPNHead is inference-only (call .eval())")
        new_instances, _ = self.refine_head(features["p2"], pred_instances, None)
        return {}, new_instances
