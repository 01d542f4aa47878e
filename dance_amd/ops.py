"""Tensor-level wrappers over the C ABI (include/dance_hip.h): validation, output
allocation and stream plumbing only -- all arithmetic happens in libdance_hip.so.

Every function requires CUDA(HIP) tensors and raises RuntimeError otherwise; there is no
CPU or torch fallback.
"""
import math
import os

import torch

from ._lib import (ABSMAX_OF_FEATURES, ABSMAX_OF_GN_RELU, ABSMAX_OF_LAYOUT, ABSMAX_WORDS, check, current_stream,
                   lib, ptr)

POLY_RAW, POLY_BOX, POLY_QUAD, POLY_OCTAGON = 0, 1, 2, 3
PAD_ZEROS, PAD_REFLECTION = 0, 1
DELOC = {"derange": 0, "demean": 1, "demin": 2, None: 3, "none": 3}
EP_RELU, EP_AFFINE, EP_RESIDUAL, EP_ROWINIT, EP_MAXV = 1, 2, 4, 8, 16


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dance_amd ops need GPU tensors (got a %s tensor)" % t.device)


def _f32c(t):
    if t.dtype != torch.float32:
        raise RuntimeError("expected a float32 tensor, got %s" % t.dtype)
    return t.contiguous()


# --------------------------------------------------------------------------- resampler
def sync_checks():
    """How the reference's device-synchronising asserts (uniform_upsample's edge-count assert,
    snake_head.py:782) are kept -- DANCE_SYNC_CHECKS:
      "1" (default): as the reference, the host waits for the status right away;
      "lazy": the status goes to pinned host memory asynchronously and is asserted by a later call
              (or by flush_checks()) once its copy has completed -- same assert, a forward or two
              late, and the host keeps queueing work meanwhile (Dsnake module at 32 x 100: 224k ->
              242k contours/s);
      "0": never looked at."""
    return os.environ.get("DANCE_SYNC_CHECKS", "1")


_PENDING = []       # (pinned slot index, event, message) of lazy checks not yet looked at
_PINNED, _PINNED_SLOTS = None, 64
_FREE_SLOTS = list(range(_PINNED_SLOTS))      # a slot returns here when its entry is flushed


def assert_not(bad, message):
    """`assert not bad` for a boolean device scalar, in the mode sync_checks() names."""
    mode = sync_checks()
    # under HIP-graph capture no status can reach the host: the check is skipped there and the
    # graphed caller looks at its outputs after the replay instead (dance_amd/graph.py)
    if mode == "0" or torch.cuda.is_current_stream_capturing():
        return
    if mode != "lazy":
        assert not bool(bad), message
        return
    global _PINNED
    if _PINNED is None:
        _PINNED = torch.zeros((_PINNED_SLOTS,), dtype=torch.bool, pin_memory=True)   # one allocation, reused
    if not _FREE_SLOTS:
        flush_checks(wait=True)                  # every slot taken: look at them all, which frees them
    slot = _FREE_SLOTS.pop()
    # copy and event on the stream `bad` was produced on: its device's current stream
    with torch.cuda.device(bad.device):
        _PINNED[slot].copy_(bad, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    _PENDING.append((slot, ev, message))
    flush_checks(wait=False)


def _assert_status(status_list):
    """The resampler's edge-count assert over a list of status tensors."""
    status_list = [s for s in status_list if s.numel()]
    if sync_checks() == "0" or not status_list:
        return
    assert_not(torch.stack([s.any() for s in status_list]).any(), "uniform_upsample: edge counts do not sum to p_num")


def flush_checks(wait=True):
    """Look at the lazy checks whose status has arrived (wait=True: wait for all of them); a check
    that fails is reported once."""
    keep, failed = [], None
    for slot, ev, message in _PENDING:
        if wait:
            ev.synchronize()
        if wait or ev.query():
            if bool(_PINNED[slot]) and failed is None:
                failed = message
            _FREE_SLOTS.append(slot)
        else:
            keep.append((slot, ev, message))
    _PENDING[:] = keep
    assert failed is None, failed


def uniform_upsample(poly, p_num, kind=POLY_RAW, check_status=True, deferred=None):
    """Fused RefineNet.uniform_upsample (snake_head.py:764-794).

    kind RAW:     poly (N,E,2) [or the reference's (1,N,E,2)] polygon vertices
    kind BOX/QUAD: poly (N,4) xyxy boxes;  kind OCTAGON: poly (N,4,2) extreme points
    Returns (points (N,p_num,2) f32, edge_start_idx (N,E) i64).  With `check_status` the
    reference's `assert torch.all(edge_num_sum == p_num)` (snake_head.py:782) is kept,
    which -- like the reference -- synchronises with the device.  `deferred`: a list; the status
    tensor is appended to it instead and the caller runs `check_upsample_status` on the list later
    (one synchronisation for a whole chain of asynchronous launches).
    """
    _need_gpu(poly)
    if kind == POLY_RAW and poly.dim() == 4:
        if poly.size(0) != 1:
            raise RuntimeError("uniform_upsample: leading batch dim must be 1")
        poly = poly[0]
    poly = _f32c(poly)
    n = poly.size(0)
    if kind == POLY_RAW:
        e = poly.size(1)
    elif kind in (POLY_BOX, POLY_QUAD):
        if poly.dim() != 2 or poly.size(1) != 4:
            raise RuntimeError("uniform_upsample: boxes must be (N,4)")
        e = 4
    else:
        if poly.dim() != 3 or tuple(poly.shape[1:]) != (4, 2):
            raise RuntimeError("uniform_upsample: extreme points must be (N,4,2)")
        e = 12
    out = torch.empty((n, p_num, 2), dtype=torch.float32, device=poly.device)
    start = torch.empty((n, e), dtype=torch.int64, device=poly.device)
    status = torch.empty((n,), dtype=torch.int32, device=poly.device)
    with torch.cuda.device(poly.device):
        check(lib().dance_uniform_upsample(kind, ptr(poly), n, e, int(p_num), ptr(out), ptr(start),
                                           ptr(status), current_stream()), "uniform_upsample")
    # (under HIP-graph capture the check, a host synchronisation, is skipped: dance_amd/graph.py)
    if deferred is not None:
        deferred.append(status)
    elif check_status and n:
        _assert_status([status])
    return out, start


def octagon_upsample(ext_points, p_num, clamp_max=None, override=None, override_mask=None, deferred=None,
                     check_status=True):
    """DsnakeHead.forward's extreme-point clamp (dsnake_head.py:800-805), ground-truth override
    (:811-817) and get_octagon (:808-810 -> 373-411) in ONE launch.

    ext_points (N,4,2) f32; clamp_max = (x_max, y_max) or None; override (N,4,2) with
    override_mask (N,) bool: rows taken from `override` unclamped.  Returns (points (N,p_num,2),
    ext_used (N,4,2): the points the octagons were built from, edge_start_idx (N,12) i64)."""
    _need_gpu(ext_points, override, override_mask)
    if ext_points.dim() != 3 or tuple(ext_points.shape[1:]) != (4, 2):
        raise RuntimeError("octagon_upsample: extreme points must be (N,4,2)")
    # a strided view (every 10th vertex of the init ring) is read in place: no copy kernel
    if ext_points.dtype != torch.float32 or ext_points.stride(2) != 1 or ext_points.stride(1) < 2 or \
            (ext_points.size(0) > 1 and ext_points.stride(0) < 2):
        ext_points = _f32c(ext_points)
    n = ext_points.size(0)
    if (override is None) != (override_mask is None):
        raise RuntimeError("octagon_upsample: override and override_mask go together")
    if override is not None:
        override = _f32c(override)
        override_mask = override_mask.to(torch.uint8).contiguous()
        if tuple(override.shape) != (n, 4, 2) or override_mask.numel() != n:
            raise RuntimeError("octagon_upsample: override (N,4,2) / override_mask (N,) expected")
    dev = ext_points.device
    out = torch.empty((n, p_num, 2), dtype=torch.float32, device=dev)
    used = torch.empty((n, 4, 2), dtype=torch.float32, device=dev)
    start = torch.empty((n, 12), dtype=torch.int64, device=dev)
    status = torch.empty((n,), dtype=torch.int32, device=dev)
    cx, cy = (float(clamp_max[0]), float(clamp_max[1])) if clamp_max is not None else (0.0, 0.0)
    with torch.cuda.device(dev):
        check(lib().dance_octagon_upsample(ptr(ext_points), max(ext_points.stride(0), 2), ext_points.stride(1),
                                           ptr(override), ptr(override_mask),
                                           int(clamp_max is not None), cx, cy, n, int(p_num), ptr(out), ptr(used),
                                           ptr(start), ptr(status), current_stream()), "octagon_upsample")
    if deferred is not None:
        deferred.append(status)
    elif check_status and n:
        _assert_status([status])
    return out, used, start


def check_upsample_status(deferred):
    """The assert of `uniform_upsample` for status tensors collected with `deferred=`."""
    if deferred:
        _assert_status(deferred)


# ------------------------------------------------------------------------------ layout
def contour_upsample_2x(locations):
    """`RefineNet.contour_upsample_2x` (snake_head.py:1752-1761): (N,V,2) -> (N,2V,2)."""
    _need_gpu(locations)
    loc = locations.contiguous().float()
    n, v, _ = loc.shape
    out = torch.empty((n, 2 * v, 2), dtype=torch.float32, device=loc.device)
    with torch.cuda.device(loc.device):
        check(lib().dance_contour_upsample_2x(ptr(loc), ptr(out), n, v, current_stream()), "contour_upsample_2x")
    return out


ABSMAX_ATTR = "_dance_absmax"


def absmax_of(feat_nhwc):
    """The per-image max |feature| words (B, ABSMAX_WORDS) int32 the fast mode's head block scales its
    input by (include/dance_hip.h "Range").  Tensors this package produces (nchw_to_nhwc,
    bottom_out16) carry them, measured on the way out; any other NHWC tensor is measured here --
    one more read of it, on every call (never cached: the caller may change it in place)."""
    words = getattr(feat_nhwc, ABSMAX_ATTR, None)
    if words is not None:
        return words
    return feature_absmax(feat_nhwc)


def feature_absmax(feat_nhwc):
    """dance_feature_absmax: (B, ...) fp32 dense -> (B, ABSMAX_WORDS) int32 float bit patterns."""
    _need_gpu(feat_nhwc)
    feat_nhwc = _f32c(feat_nhwc)
    b = feat_nhwc.size(0)
    words = torch.empty((b, ABSMAX_WORDS), dtype=torch.int32, device=feat_nhwc.device)
    if b:
        per_image = feat_nhwc[0].numel()
        partial = _absmax_partial(ABSMAX_OF_FEATURES, b, 1, per_image, feat_nhwc.device)
        with torch.cuda.device(feat_nhwc.device):
            check(lib().dance_feature_absmax(ptr(feat_nhwc), b, per_image, ptr(words), ptr(partial), current_stream()),
                  "feature_absmax")
    return words


def _absmax_partial(producer, b, c, hw, device):
    """scratch for a producer's per-workgroup maxima (include/dance_hip.h dance_absmax_partials)"""
    return torch.empty((b, max(1, int(lib().dance_absmax_partials(producer, c, hw)))), dtype=torch.int32, device=device)


def nchw_to_nhwc(features, att_channels):
    """(B, att+C, Hf, Wf) NCHW -> ((B,Hf,Wf,C) NHWC, (B,Hf,Wf) attention plane or None).  The NHWC
    tensor carries its per-image max |feature| words (`absmax_of`), measured in the same pass."""
    _need_gpu(features)
    features = _f32c(features)
    b, ct, hf, wf = features.shape
    c = ct - att_channels
    feat = torch.empty((b, hf, wf, c), dtype=torch.float32, device=features.device)
    att = torch.empty((b, hf, wf), dtype=torch.float32, device=features.device) if att_channels else None
    words = torch.empty((b, ABSMAX_WORDS), dtype=torch.int32, device=features.device)
    partial = _absmax_partial(ABSMAX_OF_LAYOUT, b, c, hf * wf, features.device)
    with torch.cuda.device(features.device):
        check(lib().dance_nchw_to_nhwc(ptr(features), ptr(feat), ptr(att), ptr(words), ptr(partial), b, ct,
                                       att_channels, hf * wf, current_stream()), "nchw_to_nhwc")
    setattr(feat, ABSMAX_ATTR, words)
    return feat, att


# ------------------------------------------------------------------------------ gather
def gather_features(feat_nhwc, att, locations, image_idx, padding_mode, deloc):
    """get_locations_feature + de_location + concat (snake_head.py:1542-1652).

    Returns (x (N, C+2 or C, V) channel-major, att_scores (N,V) or None)."""
    _need_gpu(feat_nhwc, att, locations, image_idx)
    feat_nhwc, locations = _f32c(feat_nhwc), _f32c(locations)
    b, hf, wf, c = feat_nhwc.shape
    n, v, _ = locations.shape
    deloc_t = DELOC[deloc]
    if image_idx is not None:
        image_idx = image_idx.to(torch.int32).contiguous()
    x = torch.empty((n, c + (2 if deloc_t != 3 else 0), v), dtype=torch.float32, device=locations.device)
    att_out = torch.empty((n, v), dtype=torch.float32, device=locations.device) if att is not None else None
    with torch.cuda.device(locations.device):
        check(lib().dance_gather_features(ptr(feat_nhwc), ptr(att), ptr(locations), ptr(image_idx),
                                          ptr(x), ptr(att_out), n, v, b, hf, wf, c, padding_mode,
                                          deloc_t, current_stream()), "gather_features")
    return x, att_out


# ------------------------------------------------------------------------ snake blocks
def pack_conv_weight(weight):
    """Conv1d weight (Cout, Cin, K) -> packed (K, Cin_pad16, Cout) f32 with zero pad rows."""
    cout, cin, k = weight.shape
    cin_pad = (cin + 15) // 16 * 16
    w = torch.zeros((k, cin_pad, cout), dtype=torch.float32, device=weight.device)
    w[:, :cin, :] = weight.float().permute(2, 1, 0)
    return w.contiguous()


def pack_conv_weight16(weight, scale=None):
    """Conv1d weight (Cout, Cin, K) -> split-half packing for the fast precision mode:
    (2, K, Cin_pad32/8, Cout, 8) fp16 = hi/lo planes of w * 2^s, plus 2^s.  The power-of-two
    scale keeps the lo halves normal fp16 numbers and is undone exactly in the epilogue.
    Input channels are zero-padded to a multiple of 32 (the kernels' smallest chunk)."""
    cout, cin, k = weight.shape
    cin_pad = (cin + 31) // 32 * 32
    w = torch.zeros((cout, cin_pad, k), dtype=torch.float32, device=weight.device)
    w[:, :cin, :] = weight.float()
    amax = float(w.abs().max())
    s = 1.0
    if amax > 0 and math.isfinite(amax):        # exponent kept within +-60: 2^s and 2^-s stay plain floats
        s = 2.0 ** max(-60, min(60, math.floor(math.log2(1024.0 / amax))))
    if scale is not None:               # a slice of a tensor packed elsewhere: that tensor's scale
        s = float(scale)
    ws = w * s
    hi = ws.half()
    lo = (ws - hi.float()).half()

    def arrange(t):                    # (Cout, Cin_pad, K) -> (K, Cin_pad/8, Cout, 8)
        return t.permute(2, 1, 0).reshape(k, cin_pad // 8, 8, cout).permute(0, 1, 3, 2)

    return torch.stack([arrange(hi), arrange(lo)]).contiguous(), s


def to_p16(x):
    """(N, C, V) fp32 channel-major -> P16 block (N, C/128 slots, 2, 16, V, 8) fp16
    (C a multiple of 128).  Test / glue helper; the kernels produce P16 themselves."""
    n, c, v = x.shape
    hi = x.half()
    lo = (x - hi.float()).half()
    t = torch.stack([hi, lo], dim=1).reshape(n, 2, c // 128, 16, 8, v)      # n, plane, slot, cb, j, v
    return t.permute(0, 2, 1, 3, 5, 4).contiguous()


def from_p16(p):
    """Inverse of to_p16: (N, slots, 2, 16, V, 8) fp16 -> (N, C, V) fp32 (hi + lo)."""
    n, slots, _, _, v, _ = p.shape
    f = p.float()
    x = f[:, :, 0] + f[:, :, 1]                                              # n, slot, cb, v, j
    return x.permute(0, 1, 2, 4, 3).reshape(n, slots * 128, v).contiguous()


def conv16(x16, w16, scale_pow2, bias, *, cin, cout, v, dilation=1, bn_scale=None, bn_shift=None,
           residual=False, relu=False, rowinit=None, maxv=False, out_f32=False, mfma_terms=3, residual_mul=1.0):
    """dance_conv16 on a P16 block (N, slots, 2, 16, V, 8); returns P16 (N, cout/128 slots, ...)
    or fp32 (N, cout, V) / (N, cout) with out_f32 / maxv."""
    _need_gpu(x16, w16)
    n = x16.size(0)
    taps = w16.size(1)
    slot = 256 * v
    bias_s = None if bias is None else (bias.float() * scale_pow2).contiguous()
    if maxv:
        y16, yf, yfs = None, torch.empty((n, cout), dtype=torch.float32, device=x16.device), 0
    elif out_f32:
        y16, yf, yfs = None, torch.empty((n, cout, v), dtype=torch.float32, device=x16.device), cout * v
    else:
        slots = max(1, cout // 128)
        y16 = torch.zeros((n, slots, 2, 16, v, 8), dtype=torch.float16, device=x16.device)
        yf, yfs = None, 0
    with torch.cuda.device(x16.device):
        check(lib().dance_conv16(ptr(x16), x16.stride(0), slot, ptr(w16), ptr(bias_s), 1.0 / scale_pow2,
                                 ptr(bn_scale), ptr(bn_shift), ptr(rowinit), int(residual), int(relu),
                                 int(maxv), ptr(y16), 0 if y16 is None else y16.stride(0), slot, ptr(yf),
                                 yfs, float(residual_mul), n, v, cin, cout, taps, dilation, mfma_terms,
                                 current_stream()), "conv16")
    return yf if y16 is None else y16


def circconv(x, w_packed, bias, *, cin, dilation=1, scale=None, shift=None, residual=None,
             rowinit=None, relu=False, maxv=False, out=None):
    """One circular dilated Conv1d block on channel-major activations (N,Cin,V).

    w_packed from `pack_conv_weight`; scale/shift = folded eval-BatchNorm; returns
    (N,Cout,V), or (N,Cout) with `maxv` (max over the ring)."""
    _need_gpu(x, w_packed)
    if x.dtype != torch.float32 or x.stride(2) != 1 or x.stride(1) != x.size(2):
        x = _f32c(x)
    n, cx, v = x.shape
    taps, cin_pad, cout = w_packed.shape
    if cx < cin or cin_pad != (cin + 15) // 16 * 16:
        raise RuntimeError("circconv: weight/input channel mismatch")
    flags = (EP_RELU if relu else 0) | (EP_AFFINE if scale is not None else 0) | \
            (EP_RESIDUAL if residual is not None else 0) | (EP_ROWINIT if rowinit is not None else 0) | \
            (EP_MAXV if maxv else 0)
    if out is None:
        out = torch.empty((n, cout) if maxv else (n, cout, v), dtype=torch.float32, device=x.device)
    elif not maxv and (out.stride(2) != 1 or out.stride(1) != v or out.dtype != torch.float32):
        raise RuntimeError("circconv: `out` must be channel-major (N,Cout,V) with dense rows")
    y_stride = 0 if maxv else out.stride(0)
    if residual is not None:
        residual = _f32c(residual)
        if y_stride != cout * v:
            raise RuntimeError("circconv: residual needs a dense output")
    with torch.cuda.device(x.device):
        check(lib().dance_circconv(ptr(x), x.stride(0), ptr(w_packed), ptr(bias), ptr(scale),
                                   ptr(shift), ptr(residual), ptr(rowinit), ptr(out), y_stride,
                                   n, v, cin, cout, taps, dilation, flags, current_stream()),
              "circconv")
    return out


def row_constant(g, wg, bias):
    """h[n][o] = bias[o] + sum_k wg[k][o] * g[n][k]  (vertex-invariant columns of a k=1 conv)."""
    _need_gpu(g, wg, bias)
    g, wg, bias = _f32c(g), _f32c(wg), _f32c(bias)
    n, gd = g.shape
    cout = wg.size(1)
    h = torch.empty((n, cout), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        check(lib().dance_row_constant(ptr(g), ptr(wg), ptr(bias), ptr(h), n, gd, cout,
                                       current_stream()), "row_constant")
    return h


# ------------------------------------------------------------------------ feature prep
def pack_conv3x3_weight16(weight):
    """Conv2d weight (Cout, Cin, 3, 3) -> split-half packing (2, 9, Cin/8, Cout, 8), tap = ky*3+kx."""
    cout, cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3):
        raise ValueError("3x3 kernels only")
    return pack_conv_weight16(weight.reshape(cout, cin, 9))


def bottom_out16(features, c_off, layers, groups=32, eps=1e-5, measure=True):
    """`RefineNet.bottom_out` (snake_head.py:562-615, 1816-1820): for every layer
    Conv2d 3x3 (pad 1, no bias) -> GroupNorm(groups) -> ReLU, in the fast-mode arithmetic.

    features (B, c_off + C, H, W) fp32 NCHW (channels before c_off, e.g. the attention map,
    are skipped); layers = [(w16, scale_pow2, gamma, beta), ...] with w16 from
    `pack_conv3x3_weight16`.  Returns the NHWC fp32 map (B, H, W, C) the gather kernels read;
    measure: it carries its per-image max |value| words (`absmax_of`), measured on the way out --
    switch off for a layer whose output does not go to the deformer."""
    _need_gpu(features)
    # a channels_last tensor (what this function's own output viewed as NCHW, an upsampling or a
    # sum of such tensors is) is read in place: no NCHW copy between the layers of a head
    nhwc_in = (features.dim() == 4 and features.dtype == torch.float32 and not features.is_contiguous()
               and features.permute(0, 2, 3, 1).is_contiguous())
    if not nhwc_in:
        features = _f32c(features)
    b, ct, h, w = features.shape
    c = ct - c_off
    dev = features.device
    if not layers:
        raise ValueError("no layers")
    x16 = torch.empty((b, 2, c // 8, h, w, 8), dtype=torch.float16, device=dev)
    yb = torch.empty((b, c // 16, h, w, 16), dtype=torch.float32, device=dev)
    stats = torch.empty((b, groups, 2), dtype=torch.float64, device=dev)
    spart = torch.empty((max(1, int(lib().dance_conv3x3_16_partial_floats(b, c, h, w, groups))),), dtype=torch.float32,
                        device=dev)
    out = torch.empty((b, h, w, c), dtype=torch.float32, device=dev)
    words = partial = None
    if measure:
        words = torch.empty((b, ABSMAX_WORDS), dtype=torch.int32, device=dev)
        partial = _absmax_partial(ABSMAX_OF_GN_RELU, b, c, h * w, dev)
    with torch.cuda.device(dev):
        st = current_stream()
        to_p16 = lib().dance_nhwc_to_p16img if nhwc_in else lib().dance_nchw_to_p16img
        check(to_p16(ptr(features), ct, c_off, ptr(x16), b, c, h, w, st), "to_p16img")
        for i, (w16, scale, gamma, beta) in enumerate(layers):
            if w16.shape[2] * 8 != c or w16.shape[3] != c:
                raise ValueError("bottom_out layers are C -> C")
            check(lib().dance_conv3x3_16(ptr(x16), ptr(w16), 1.0 / scale, ptr(yb), ptr(stats), ptr(spart), b, c, c, h, w,
                                         groups, st), "conv3x3_16")
            last = i + 1 == len(layers)
            check(lib().dance_gn_relu(ptr(yb), ptr(stats), ptr(gamma), ptr(beta), eps, groups,
                                      None if last else ptr(x16), ptr(out) if last else None,
                                      ptr(words) if last else None, ptr(partial) if last else None, b, c, h, w, st),
                  "gn_relu")
    if measure:
        setattr(out, ABSMAX_ATTR, words)
    return out


def is_nhwc(t):
    """NCHW-shaped fp32 GPU tensor whose memory is dense NHWC (channels_last)."""
    return (t.dim() == 4 and t.is_cuda and t.dtype == torch.float32 and t.permute(0, 2, 3, 1).is_contiguous())


def upsample2x_add(src, acc=None, inplace=False):
    """acc + F.interpolate(src, scale_factor=2, mode="bilinear", align_corners=False) for
    channels_last tensors (B,C,H,W) / (B,C,2H,2W), in one launch; acc None: the upsampling
    alone; inplace: the sum is written into acc.  Returns a channels_last (B,C,2H,2W) tensor."""
    if not is_nhwc(src) or (acc is not None and not is_nhwc(acc)):
        raise RuntimeError("upsample2x_add: channels_last fp32 GPU tensors expected")
    b, c, h, w = src.shape
    if acc is not None and tuple(acc.shape) != (b, c, 2 * h, 2 * w):
        raise RuntimeError("upsample2x_add: acc must be (B, C, 2H, 2W)")
    out = acc if (inplace and acc is not None) else torch.empty((b, 2 * h, 2 * w, c), dtype=torch.float32,
                                                                device=src.device).permute(0, 3, 1, 2)
    with torch.cuda.device(src.device):
        check(lib().dance_upsample2x_add(ptr(src), ptr(acc), ptr(out), b, h, w, c, current_stream()),
              "upsample2x_add")
    return out


def conv1x1_nhwc(x, weight, bias=None):
    """nn.Conv2d(C, K, 1) on a channels_last fp32 GPU tensor (B,C,H,W), summed in a fixed order
    (dance_conv1x1_nhwc): -> (B,K,H,W) contiguous.  The edge predictor's last layer (edge_det.py:197)."""
    if not is_nhwc(x):
        raise RuntimeError("conv1x1_nhwc: a channels_last fp32 GPU tensor expected")
    b, c, h, w = x.shape
    k = weight.shape[0]
    wt = weight.detach().reshape(k, c).float().contiguous()
    bs = None if bias is None else bias.detach().float().contiguous()
    out = torch.empty((b, k, h, w), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().dance_conv1x1_nhwc(ptr(x), ptr(wt), ptr(bs), ptr(out), b, h, w, c, k, current_stream()),
              "conv1x1_nhwc")
    return out


# ---------------------------------------------------------------------------- attender
def attender(x, w1, gn1_w, gn1_b, w2_packed, gn2_w, gn2_b, w3, b3, eps=1e-5):
    """EdgeSnakeFPNHead.attender (edge_det.py:66-100) on x = 1 - pred_edge, (B,1,H,W) or (B,H,W)
    -> attention map of the same shape.  w2_packed = attender.1.weight.permute(1,2,3,0)."""
    _need_gpu(x, w1, w2_packed, w3)
    shape = x.shape
    if x.dim() == 4:
        if x.size(1) != 1:
            raise RuntimeError("attender: input must have one channel")
        x = x[:, 0]
    x = _f32c(x)
    b, h, w = x.shape
    att = torch.empty_like(x)
    ws_bytes = lib().dance_attender_workspace_bytes(b, h, w)
    ws = torch.empty((max(ws_bytes, 8) // 8,), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().dance_attender(ptr(x), ptr(w1), ptr(gn1_w), ptr(gn1_b), ptr(w2_packed), ptr(gn2_w),
                                   ptr(gn2_b), ptr(w3), ptr(b3), float(eps), ptr(att), ptr(ws), ws_bytes,
                                   b, h, w, current_stream()), "attender")
    return att.view(shape)
