"""Host side of the contour-deformation path: weight packing and the iteration driver.

`ContourDeformer.deform` is the tensor-level API the benchmark and the tests use
(features + initial contours in, one deformed contour tensor per iteration out); the
drop-in modules in dance_amd/modules.py wrap it behind the reference's
`forward(features, pred_instances, targets)` interface.

Every arithmetic step runs in libdance_hip.so (see include/dance_hip.h); this file only
re-lays-out parameters once and enqueues `dance_snake_evolve` per iteration.
"""
import ctypes
import math
import os

import torch

from . import ops
from ._lib import EVOLVE_EVENTS, PolicyStruct, SnakeNet16Struct, SnakeNetStruct, check, current_stream, lib, ptr

BN_EPS = 1e-5                          # nn.BatchNorm1d default, snake_head.py:146
DEFAULT_DILATIONS = (1, 1, 1, 2, 2, 4, 4)   # defaults.py:60-64 / dsnake_head.py:102


ACT_TARGET_EXP = 4     # a stored activation's ESTIMATED largest magnitude lands in (2^3, 2^4]: twelve octaves
                       # of headroom to fp16's 65504 (a worse under-estimate turns into inf -> NaN, loud) and
                       # seven down to 2^-3, below which a hi/lo pair starts to lose bits (csrc/p16.h) -- with
                       # an absolute resolution of 2^-25 the pair is then still finer than 2^-28 of the
                       # block's largest value.  Measured on the synthetic nets (tools/act_range_probe.py):
                       # DANCE activations sit within 2^2.4 of the estimate; Dsnake's, whose demin
                       # de_location is in pixels and whose synthetic BatchNorm statistics do not fit it,
                       # up to 2^7.7 ABOVE -- hence the asymmetric placement.


def act_scale_exp(estimate):
    """a with estimate * 2^a in (2^(ACT_TARGET_EXP - 1), 2^ACT_TARGET_EXP]; 0 for a useless estimate."""
    if not (estimate > 0 and math.isfinite(estimate)):
        return 0
    return max(-40, min(40, ACT_TARGET_EXP - math.ceil(math.log2(estimate))))


def _fold_bn(sd, prefix):
    """Eval BatchNorm1d as y*scale + shift -- the same folded form torch's CPU kernel
    evaluates (alpha = w * invstd, beta = b - mean * alpha)."""
    w, b = sd[prefix + "weight"].float(), sd[prefix + "bias"].float()
    mean, var = sd[prefix + "running_mean"].float(), sd[prefix + "running_var"].float()
    invstd = 1.0 / torch.sqrt(var + BN_EPS)
    scale = w * invstd
    return scale.contiguous(), (b - mean * scale).contiguous()


class SnakeNetWeights:
    """One `_SnakeNet` (snake_head.py:220-284 / dsnake_head.py:95-136) packed for the
    kernels: conv weights as (tap, cin_pad16, cout), BatchNorm folded to scale/shift,
    prediction[0] split into its global-state columns (256, hoisted per contour) and its
    per-vertex columns (1024)."""

    def __init__(self, sd, device, dilations=DEFAULT_DILATIONS):
        if len(dilations) != 7:
            raise ValueError("the kernels implement the 8-layer snake (head + 7 residual blocks)")
        dev = torch.device(device)
        g = lambda k: sd[k].detach().to(dev, torch.float32)   # noqa: E731
        sdd = {k: g(k) for k in sd if not k.endswith("num_batches_tracked")}
        self.dilations = tuple(int(d) for d in dilations)
        self.state_dim = sdd["head.conv.fc.weight"].shape[0]
        self.feature_dim = sdd["head.conv.fc.weight"].shape[1]      # C + 2
        if self.state_dim != 128 or sdd["fusion.weight"].shape[:2] != (256, 1024):
            raise ValueError("unsupported snake geometry (need state_dim 128, 8 states, fusion 256)")
        t = {}
        t["head_w"] = ops.pack_conv_weight(sdd["head.conv.fc.weight"])
        t["head_b"] = sdd["head.conv.fc.bias"].contiguous()
        t["head_scale"], t["head_shift"] = _fold_bn(sdd, "head.norm.")
        for i in range(7):
            t["res%d_w" % i] = ops.pack_conv_weight(sdd["res%d.conv.fc.weight" % i])
            t["res%d_b" % i] = sdd["res%d.conv.fc.bias" % i].contiguous()
            t["res%d_scale" % i], t["res%d_shift" % i] = _fold_bn(sdd, "res%d.norm." % i)
        t["fusion_w"] = ops.pack_conv_weight(sdd["fusion.weight"])
        t["fusion_b"] = sdd["fusion.bias"].contiguous()
        p0 = sdd["prediction.0.weight"]                              # (256, 1280, 1)
        t["pred0_wg"] = p0[:, :256, 0].t().contiguous()              # [g][o]
        t["pred0_ws"] = ops.pack_conv_weight(p0[:, 256:, :])
        t["pred0_b"] = sdd["prediction.0.bias"].contiguous()
        t["pred1_w"] = ops.pack_conv_weight(sdd["prediction.2.weight"])
        t["pred1_b"] = sdd["prediction.2.bias"].contiguous()
        w3 = sdd["prediction.4.weight"]                              # (2, 64, 1)
        if w3.shape[0] != 2:
            raise ValueError("MULTI_OFFSET != 1 is not supported (reference default is 1)")
        t["pred2_w"] = w3[:, :, 0].t().contiguous()                  # (64, 2)
        t["pred2_b"] = sdd["prediction.4.bias"].contiguous()
        self.tensors = t
        self.device = dev
        s = SnakeNetStruct()
        for name in ("head_w", "head_b", "head_scale", "head_shift", "fusion_w", "fusion_b",
                     "pred0_wg", "pred0_ws", "pred0_b", "pred1_w", "pred1_b", "pred2_w", "pred2_b"):
            setattr(s, name, t[name].data_ptr())
        for i in range(7):
            s.res_w[i] = t["res%d_w" % i].data_ptr()
            s.res_b[i] = t["res%d_b" % i].data_ptr()
            s.res_scale[i] = t["res%d_scale" % i].data_ptr()
            s.res_shift[i] = t["res%d_shift" % i].data_ptr()
            s.dilation[i] = self.dilations[i]
        s.feat_channels = self.feature_dim - 2
        self.struct = s
        self._pack16(sdd)

    def _pack16(self, sdd):
        """Split-half (P16) packing for the fast precision mode (dance_snake_evolve16), with the
        power-of-two activation scales of csrc/p16.h folded into the constants.

        State slot l (l = 0: the head block's output, l + 1: residual block l's) is stored as
        x * 2^(a_l), prediction[0]'s output as x * 2^(a_p).  The exponents come from estimates
        of the largest magnitudes: a block's BatchNorm output is bounded by |beta| + 4 |gamma|
        per channel as long as its running statistics fit the data, the residual stream by the
        running sum of those bounds; prediction[0]'s output by weight-row norms times the state
        bounds.  An estimate may be 2^12 too low or 2^7 too high before anything is lost (ACT_TARGET_EXP)."""
        t = self.tensors
        s = SnakeNet16Struct()
        scales = {}

        def pack(name, weight):
            w16, sc = ops.pack_conv_weight16(weight)
            t[name + "_w16"] = w16
            scales[name] = sc
            return w16.data_ptr(), sc

        def bn_bound(prefix):
            return float((sdd[prefix + "bias"].abs() + 4.0 * sdd[prefix + "weight"].abs()).max())

        # estimated bound M_l of slot l and its exponent a_l
        bound, m = [], 0.0
        for prefix in ["head.norm."] + ["res%d.norm." % i for i in range(7)]:
            m += bn_bound(prefix)
            bound.append(m)
        a = [act_scale_exp(x) for x in bound]
        p2 = lambda e: 2.0 ** e        # noqa: E731

        # the head block: the feature columns (head_w; the two de_location columns zeroed, they are
        # not read from this tensor) and the de_location columns (head_wd) at separate scales --
        # features and de_location values differ in magnitude by whatever the checkpoint says
        hw = sdd["head.conv.fc.weight"]                                           # (128, C + 2, 9)
        hw_feat = hw.clone()
        hw_feat[:, -2:, :] = 0
        s.head_w, sc = pack("head", hw_feat)
        s.head_winv = 1.0 / sc
        t["head_bs"] = (sdd["head.conv.fc.bias"] * sc).contiguous()
        s.head_bs = t["head_bs"].data_ptr()
        # the two de_location input channels as one 32-deep K step (head16.hip): k = 0..8 the 9 ring
        # taps of dx, 9..17 of dy, the rest zero
        wd = torch.zeros((hw.shape[0], 32, 1), dtype=torch.float32, device=hw.device)
        wd[:, 0:9, 0] = hw[:, -2, :]
        wd[:, 9:18, 0] = hw[:, -1, :]
        t["head_wd16"], scd = ops.pack_conv_weight16(wd)
        scales["head_wd"] = scd
        s.head_wd, s.head_wd_winv = t["head_wd16"].data_ptr(), 1.0 / scd
        t["head_scale16"] = (t["head_scale"] * p2(a[0])).contiguous()
        t["head_shift16"] = (t["head_shift"] * p2(a[0])).contiguous()
        s.head_scale, s.head_shift = t["head_scale16"].data_ptr(), t["head_shift16"].data_ptr()
        for i in range(7):
            n = "res%d" % i
            s.res_w[i], sc = pack(n, sdd[n + ".conv.fc.weight"])
            s.res_winv[i] = p2(-a[i]) / sc
            s.res_rmul[i] = p2(a[i + 1] - a[i])
            t[n + "_bs"] = (sdd[n + ".conv.fc.bias"] * (sc * p2(a[i]))).contiguous()
            t[n + "_scale16"] = (t[n + "_scale"] * p2(a[i + 1])).contiguous()
            t[n + "_shift16"] = (t[n + "_shift"] * p2(a[i + 1])).contiguous()
            s.res_bs[i] = t[n + "_bs"].data_ptr()
            s.res_scale[i], s.res_shift[i] = t[n + "_scale16"].data_ptr(), t[n + "_shift16"].data_ptr()
            s.dilation[i] = self.dilations[i]
        # fusion and the state columns of prediction[0] as ONE 512-output GEMM over the 8 slots: the
        # columns of slot j divided by 2^(a_j), each half of the rows packed with its own scale
        col = torch.tensor([p2(-a[j]) for j in range(8)], dtype=torch.float32,
                           device=hw.device).repeat_interleave(128)[None, :, None]
        p0 = sdd["prediction.0.weight"]
        wf, wp = sdd["fusion.weight"], p0[:, 256:, :]
        f16, sf = ops.pack_conv_weight16(wf * col)
        p16, sp = ops.pack_conv_weight16(wp * col)
        t["fuse_w16"] = torch.cat([f16, p16], dim=3).contiguous()                 # (2,1,128,512,8)
        scales["fuse"], scales["pred0"] = sf, sp
        s.fuse_w = t["fuse_w16"].data_ptr()
        t["fuse_bs"] = torch.cat([sdd["fusion.bias"] * sf, torch.zeros_like(sdd["fusion.bias"])]).contiguous()
        s.fuse_bs = t["fuse_bs"].data_ptr()
        # prediction[0]'s output scale: |state part| + |row constant|, from row norms and the slot bounds
        mb = torch.tensor(bound, dtype=torch.float32, device=hw.device)
        row = lambda w: float(((w[:, :, 0].reshape(w.shape[0], 8, 128) ** 2).sum(2) * mb[None] ** 2).sum(1).sqrt().max())  # noqa: E731
        g_bound = float(sdd["fusion.bias"].abs().max()) + row(wf)
        h_bound = float(sdd["prediction.0.bias"].abs().max()) + float((t["pred0_wg"] ** 2).sum(0).sqrt().max()) * g_bound
        ap = act_scale_exp(row(wp) + h_bound)
        s.fuse_winv, s.pred0_winv = 1.0 / sf, p2(ap) / sp
        t["pred0_wg16"] = (t["pred0_wg"] * p2(ap)).contiguous()
        t["pred0_b16"] = (t["pred0_b"] * p2(ap)).contiguous()
        s.pred0_wg, s.pred0_b = t["pred0_wg16"].data_ptr(), t["pred0_b16"].data_ptr()
        s.pred1_w, sc = pack("pred1", sdd["prediction.2.weight"])
        s.pred1_winv = p2(-ap) / sc
        t["pred1_bs"] = (sdd["prediction.2.bias"] * (sc * p2(ap))).contiguous()
        s.pred1_bs = t["pred1_bs"].data_ptr()
        s.pred2_w, s.pred2_b = t["pred2_w"].data_ptr(), t["pred2_b"].data_ptr()
        s.feat_channels = self.feature_dim - 2
        self.struct16 = s
        self.scales16 = scales
        self.act_exps = {"slots": a, "pred0": ap, "bounds": bound}


def sub_state_dict(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


_CUS = {}


def _cu_count(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _CUS:
        _CUS[key] = torch.cuda.get_device_properties(key).multi_processor_count
    return _CUS[key]


class ContourDeformer:
    """Iterative contour deformation on one GPU.

    nets: one SnakeNetWeights per iteration (deformer0..2, snake_head.py:621-630).
    padding: "reflection" (DANCE, snake_head.py:1570) or "zeros" (Dsnake, dsnake_head.py:690).
    de_loc_type: "derange" | "demean" | "demin" (snake_head.py:1587-1610).
    individual_scale: tanh(offset) * (w,h) (snake_head.py:1702-1703).
    attention: features carry the attention map; offsets of iterations >= 1 are multiplied
               by the attention sampled at the iteration's input vertices (1643-1645, 1705-1707).
    """

    MAX_CHUNK = 4096      # contours per kernel sequence: bounds the workspace

    def __init__(self, nets, padding="reflection", de_loc_type="derange", individual_scale=True,
                 attention=True, fused_head=True, precision=None):
        """precision: "fp16" -- APPROXIMATE: plain fp16 operands (one MFMA per product, fp32
        accumulate), ~1e-2 coordinate error, outside the 1e-4 parity bar; never the default.
        "fp16x3" -- split-half operands, three f16 MFMAs per product, fp32
        accumulation (fp32-level accuracy, the default); "fp32" -- exact fp32 MFMA."""
        if precision is None:
            precision = os.environ.get("DANCE_PRECISION", "fp16x3")
        if precision not in ("fp16x3", "fp32", "fp16"):
            raise ValueError("precision must be 'fp16x3', 'fp32' or 'fp16'")
        self.precision = precision
        self.nets = list(nets)
        self._padding_name, self._deloc_name, self._side = padding, de_loc_type, None
        self.padding = ops.PAD_REFLECTION if padding == "reflection" else ops.PAD_ZEROS
        self.deloc = ops.DELOC[de_loc_type]
        self.individual_scale = bool(individual_scale)
        self.attention = bool(attention)
        self.fused_head = bool(fused_head)
        self.policy = PolicyStruct()         # kernel-form choices of the fast mode (include/dance_hip.h dance_policy)
        self._ws = None

    def _workspace(self, n, v, device):
        need = lib().dance_snake_workspace_bytes(n, v)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=device)
        return self._ws, need

    def prepare_features(self, features):
        """(B, [1+]C, Hf, Wf) NCHW (channel 0 = attention map when `attention`) ->
        (feat NHWC (B,Hf,Wf,C), att (B,Hf,Wf) or None).  The split replaces
        snake_head.py:1812-1814; the re-concatenation of :1824 never happens."""
        return ops.nchw_to_nhwc(features, 1 if self.attention else 0)

    def _deform_streams(self, features, init_contours, image_idx, image_sizes, whs, layout, k):
        feat_att = self.prepare_features(features) if layout == "nchw" else features
        n = init_contours.size(0)
        cuts = [n * i // k for i in range(k + 1)]
        cur = torch.cuda.current_stream(init_contours.device)
        if getattr(self, "_side", None) is None or len(self._side) != k:
            # every part needs its own workspace: one helper deformer (sharing the packed nets) per stream
            self._side = [(torch.cuda.Stream(init_contours.device),
                           ContourDeformer(self.nets, self._padding_name, self._deloc_name, self.individual_scale,
                                           self.attention, self.fused_head, self.precision)) for _ in range(k)]
        parts = []
        for i, (st, d) in enumerate(self._side):
            d.MAX_CHUNK = self.MAX_CHUNK
            d.policy = self.policy
            d._no_peel = True                    # a part of a split batch is not split again
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                sl = slice(cuts[i], cuts[i + 1])
                parts.append(d.deform(feat_att, init_contours[sl], image_idx[sl], image_sizes,
                                      None if whs is None else whs[sl], layout="nhwc"))
        for st, _ in self._side:
            cur.wait_stream(st)
        for p in parts:
            for t in p:
                t.record_stream(cur)
        return [torch.cat([p[i] for p in parts]) for i in range(len(self.nets))]

    def peel_count(self, n, device):
        """Contours deform() hands to a side stream so that the main stream's kernels run whole rounds
        of workgroups: n mod CUs.  Every kernel of the sequence is one workgroup per contour with 1-3
        resident per CU, so a batch that is not a multiple of the CU count pays nearly a full round for
        its remainder in every kernel (BASELINE config 2's 1600 contours on 256 CUs: 6.25 rounds,
        ~5 % of the step).  The remainder's own kernel chain is small and overlaps the main
        one.  0 = no peel: batches below two rounds, chunked batches, under stream capture,
        DANCE_PEEL=0.  Results are bit-identical either way (contours are independent)."""
        if os.environ.get("DANCE_PEEL", "1") == "0" or device.type != "cuda" or getattr(self, "_no_peel", False):
            return 0
        cus = _cu_count(device)
        if n < 2 * cus or n > self.MAX_CHUNK or n % cus == 0 or torch.cuda.is_current_stream_capturing():
            return 0
        return n % cus

    def _deform_peeled(self, features, init_contours, image_idx, image_sizes, whs, layout, stage_events, side_n):
        """deform() with the last `side_n` contours as their own kernel chain on a side stream (forked
        after the layout pass, joined before returning); both chains write into the same output
        tensors, `stage_events` time the main stream's launches."""
        feat_att = self.prepare_features(features) if layout == "nchw" else features
        n, v = init_contours.shape[:2]
        m = n - side_n
        dev = init_contours.device
        cur = torch.cuda.current_stream(dev)
        if getattr(self, "_peel_side", None) is None or self._peel_side[0].device != dev:
            self._peel_side = (torch.cuda.Stream(dev),
                               ContourDeformer(self.nets, self._padding_name, self._deloc_name, self.individual_scale,
                                               self.attention, self.fused_head, self.precision))
        st, side = self._peel_side
        side._no_peel = True
        side.policy = self.policy
        locations = init_contours.contiguous().float()
        whs = None if whs is None else whs.contiguous().float()
        idx = image_idx.to(torch.int32).contiguous()
        preds = [torch.empty((n, v, 2), dtype=torch.float32, device=dev) for _ in self.nets]
        feat, att = feat_att
        absmax = self._absmax(feat)              # measured once (here, if the tensor does not carry it) for both chains
        st.wait_stream(cur)                      # features, inputs and the output buffers exist
        for d, lo, hi, stream, ev in ((side, m, n, st, None), (self, 0, m, cur, stage_events)):
            with torch.cuda.stream(stream):
                prev = locations[lo:hi]
                for i in range(len(self.nets)):
                    prev = d.evolve(i, feat, att, prev, idx[lo:hi], None if whs is None else whs[lo:hi],
                                    use_att=(self.attention and i > 0), clip_first=False, clip_max=(0.0, 0.0),
                                    out=preds[i][lo:hi], stage_events=None if ev is None else ev[i], absmax=absmax)
        cur.wait_stream(st)                      # everything the side stream touched is ordered before what follows
        return preds

    def _fast(self, c, x_in=None):
        """Does an iteration on c feature channels run the split-half kernels?"""
        return self.precision in ("fp16x3", "fp16") and x_in is None and c % 64 == 0

    def _absmax(self, feat):
        """The features' per-image max words for the fast mode's head block (ops.absmax_of), or None."""
        return ops.absmax_of(feat) if self._fast(feat.shape[-1]) else None

    def evolve(self, it, feat, att, locations, image_idx, whs, *, use_att, clip_first, clip_max,
               x_in=None, out=None, stage_events=None, absmax=None):
        """One iteration with deformer `it` (RefineNet.evolve, snake_head.py:1624-1713).
        stage_events: optional list of EVOLVE_EVENTS recorded torch.cuda.Event(enable_timing),
        entries may be None (boundary not timed)
        (single-chunk calls only) -- bench instrumentation.
        absmax: `ops.absmax_of(feat)` when the caller already holds it (fast mode only)."""
        net = self.nets[it]
        n, v, _ = locations.shape
        dev = locations.device
        locations = locations.contiguous()
        if out is None:
            out = torch.empty((n, v, 2), dtype=torch.float32, device=dev)
        if n == 0:
            return out
        b, hf, wf, c = feat.shape
        if c != net.feature_dim - 2:
            raise RuntimeError("feature channels %d do not match the snake's %d" % (c, net.feature_dim - 2))
        if self.individual_scale and whs is None:
            raise RuntimeError("individual_scale needs whs")
        if image_idx is not None and image_idx.dtype != torch.int32:
            image_idx = image_idx.to(torch.int32)
        chunk = min(n, self.MAX_CHUNK)
        ws, need = self._workspace(chunk, v, dev)
        ev = None
        if stage_events is not None:
            if n > chunk or len(stage_events) != EVOLVE_EVENTS:
                raise RuntimeError("stage events need a single-chunk call and %d events" % EVOLVE_EVENTS)
            ev = (ctypes.c_void_p * EVOLVE_EVENTS)(*[None if e is None else e.cuda_event for e in stage_events])
        fast = self._fast(c, x_in)               # else exact fp32: precision="fp32", a precomputed input, odd widths
        if self.precision == "fp16" and x_in is None and not fast:
            raise RuntimeError("precision 'fp16' needs feature channels % 64 == 0")
        terms = 1 if self.precision == "fp16" else 3
        if fast and absmax is None:
            absmax = ops.absmax_of(feat)
        with torch.cuda.device(dev):
            stream = current_stream()
            for n0 in range(0, n, chunk):
                nb = min(chunk, n - n0)
                idx_p = ptr(image_idx[n0:]) if image_idx is not None else None
                whs_p = ptr(whs[n0:]) if whs is not None else None
                if fast:
                    check(lib().dance_snake_evolve16(
                        ctypes.byref(net.struct16), ptr(feat), ptr(att), ptr(absmax), ptr(locations[n0:]), idx_p,
                        whs_p, ptr(out[n0:]), ptr(ws), need, nb, v, b, hf, wf, self.padding, self.deloc,
                        int(self.individual_scale), int(use_att), int(clip_first and n0 == 0),
                        float(clip_max[0]), float(clip_max[1]), terms, ctypes.byref(self.policy), ev, stream),
                        "snake_evolve16")
                else:
                    # exact fp32 (precision="fp32", a precomputed input x_in, odd channel counts)
                    check(lib().dance_snake_evolve(
                        ctypes.byref(net.struct), ptr(feat), ptr(att), ptr(locations[n0:]), idx_p, whs_p,
                        ptr(x_in[n0:]) if x_in is not None else None, ptr(out[n0:]), ptr(ws), need,
                        nb, v, b, hf, wf, self.padding, self.deloc, int(self.individual_scale),
                        int(use_att), int(clip_first and n0 == 0), float(clip_max[0]), float(clip_max[1]),
                        int(self.fused_head), ev, stream), "snake_evolve")
        return out

    def init_evolve(self, it, feat, locations, image_idx, fuse, out=None, stage_events=None, absmax=None):
        """Dsnake's init stage (DsnakeHead.init, dsnake_head.py:718-744) with deformer `it` in one
        kernel sequence of the fast precision mode: gather + centre feature + `fuse` +
        de_location + head block in one launch, then stack / fusion / tail.  fuse: InitFuse16Struct
        (the packed `fuse` Conv1d).  Needs 256 feature channels and rings of at most 64 vertices
        (`supports_init_evolve`)."""
        net = self.nets[it]
        n, v, _ = locations.shape
        dev = locations.device
        locations = locations.contiguous()
        if out is None:
            out = torch.empty((n, v, 2), dtype=torch.float32, device=dev)
        if n == 0:
            return out
        b, hf, wf, c = feat.shape
        if not self.supports_init_evolve(c, v):
            raise RuntimeError("init_evolve needs the fast precision mode, 256 channels, 16 <= v <= 64")
        if image_idx is not None and image_idx.dtype != torch.int32:
            image_idx = image_idx.to(torch.int32)
        chunk = min(n, self.MAX_CHUNK)
        ws, need = self._workspace(chunk, v, dev)
        ev = None
        if stage_events is not None:
            if n > chunk or len(stage_events) != EVOLVE_EVENTS:
                raise RuntimeError("stage events need a single-chunk call and %d events" % EVOLVE_EVENTS)
            ev = (ctypes.c_void_p * EVOLVE_EVENTS)(*[None if e is None else e.cuda_event for e in stage_events])
        if absmax is None:
            absmax = ops.absmax_of(feat)
        with torch.cuda.device(dev):
            stream = current_stream()
            for n0 in range(0, n, chunk):
                nb = min(chunk, n - n0)
                check(lib().dance_snake_init16(
                    ctypes.byref(net.struct16), ctypes.byref(fuse), ptr(feat), ptr(locations[n0:]),
                    ptr(image_idx[n0:]) if image_idx is not None else None, ptr(out[n0:]), ptr(ws), need,
                    nb, v, b, hf, wf, self.padding, self.deloc, 1 if self.precision == "fp16" else 3,
                    ptr(absmax), ctypes.byref(self.policy), ev, stream), "snake_init16")
        return out

    def supports_init_evolve(self, c, v):
        return self.precision in ("fp16x3", "fp16") and c == 256 and 16 <= v <= 64 and v % 4 == 0

    def deform(self, features, init_contours, image_idx, image_sizes, whs, layout="nchw",
               stage_events=None, streams=1):
        """features: NCHW (B,[1+]C,Hf,Wf) as the reference's evolve loop sees it after
        `bottom_out` (+ attention map in channel 0), or with layout="nhwc" the pair
        (feat NHWC, att) from `prepare_features`.
        init_contours (N,V,2) f32 image px; image_idx (N,) float/int tensor or None;
        image_sizes: (h,w) when image_idx is None, else list of (h,w);
        whs (N,2) box sizes.  Returns [contours after iteration 0, 1, 2] (N,V,2).

        Reference driver: snake_head.py:1822-1862 (single image) / 1969-1995 (batched).
        `clip_locations` (1715-1727) is reproduced as it behaves: with image_idx None only
        contour 0 is clamped to the image; with a tensor image_idx nothing is clamped.

        streams > 1: the contours are split into that many contiguous parts, each deformed
        (all iterations) on its own HIP stream -- contours are independent, so one part's kernel
        tails and launch gaps overlap the other's kernels (measured: 2 streams -4 % wall on
        config 2; results bit-identical)."""
        if layout == "nhwc" and self._fast(features[0].shape[-1]) and getattr(features[0], ops.ABSMAX_ATTR, None) is None:
            # an NHWC tensor from elsewhere: measure its per-image maxima once for this call and hand them
            # down on a private view (nothing is cached on the caller's tensor, which may change in place)
            feat = features[0].view(features[0].shape)
            setattr(feat, ops.ABSMAX_ATTR, ops.feature_absmax(features[0]))
            features = (feat, features[1])
        if streams > 1 and init_contours.size(0) >= 2 * streams and stage_events is None and image_idx is not None:
            return self._deform_streams(features, init_contours, image_idx, image_sizes, whs, layout, streams)
        side_n = self.peel_count(init_contours.size(0), init_contours.device) if image_idx is not None else 0
        if side_n:
            return self._deform_peeled(features, init_contours, image_idx, image_sizes, whs, layout, stage_events,
                                       side_n)
        if layout == "nchw":
            feat, att = self.prepare_features(features)
        else:
            feat, att = features
        locations = init_contours.contiguous().float()
        if whs is not None:
            whs = whs.contiguous().float()
        idx = None
        if image_idx is not None:
            idx = image_idx.to(torch.int32).contiguous()
        clip_first = image_idx is None
        clip_max = (image_sizes[1] - 1, image_sizes[0] - 1) if clip_first else (0.0, 0.0)
        absmax = self._absmax(feat)
        preds, cur = [], locations
        for i in range(len(self.nets)):
            cur = self.evolve(i, feat, att, cur, idx, whs, use_att=(self.attention and i > 0),
                              clip_first=clip_first, clip_max=clip_max,
                              stage_events=None if stage_events is None else stage_events[i], absmax=absmax)
            preds.append(cur)
        return preds
