"""ORACLE -- TEST INFRASTRUCTURE ONLY.

fp32 torch-CPU restatement of the reference's contour-deformation path, written from
scratch as plain functions over tensors and state-dicts (no nn.Module, no detectron2).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it, and
only as the checker / the timed CPU baseline; the product (dance_amd/) never does.

Parity status: pinned by tests/golden/*.npz, which oracle/gen_golden.py produced in the
build container by running the reference's own Python
(/root/reference/core/modeling/edge_snake/snake_head.py and
 .../dsnake_baseline/dsnake_head.py) on the same seeded inputs.

Each function cites the reference lines it follows.  "sd" arguments are flat dicts with
the reference's state-dict key names relative to one `_SnakeNet`
(head.conv.fc.weight, head.norm.running_mean, res3.conv.fc.bias, fusion.weight,
 prediction.0.weight, ...).
"""
import torch
import torch.nn.functional as F

from . import extreme_utils_ref as _ext

BN_EPS = 1e-5                       # nn.BatchNorm1d default (snake_head.py:146)
DILATIONS = (1, 1, 1, 2, 2, 4, 4)   # defaults.py:60-64, dsnake_head.py:102
STRIDE = 4                          # snake_head.py:1543-1544


# ---------------------------------------------------------------------------- resampler
def uniform_upsample(poly, p_num):
    """snake_head.py:764-794.  poly (1,N,E,2) -> ((N,p,2), edge_start_idx (N,E))."""
    if poly.size(1) == 0:
        return torch.zeros([0, p_num, 2]), None
    next_poly = torch.roll(poly, -1, 2)
    edge_len = (next_poly - poly).pow(2).sum(3).sqrt()
    edge_num = torch.round(edge_len * p_num / torch.sum(edge_len, dim=2)[..., None]).long()
    edge_num = torch.clamp(edge_num, min=1)
    edge_num_sum = torch.sum(edge_num, dim=2)
    edge_idx_sort = torch.argsort(edge_num, dim=2, descending=True, stable=True)
    _ext.calculate_edge_num(edge_num, edge_num_sum, edge_idx_sort, p_num)
    edge_num_sum = torch.sum(edge_num, dim=2)
    assert torch.all(edge_num_sum == p_num)
    edge_start_idx = torch.cumsum(edge_num, dim=2) - edge_num
    weight, ind = _ext.calculate_wnp(edge_num, edge_start_idx, p_num)
    poly1 = poly.gather(2, ind[..., 0:1].expand(ind.size(0), ind.size(1), ind.size(2), 2))
    poly2 = poly.gather(2, ind[..., 1:2].expand(ind.size(0), ind.size(1), ind.size(2), 2))
    poly = poly1 * (1 - weight) + poly2 * weight
    return poly[0], edge_start_idx[0]


def box_polygon(boxes):
    """snake_head.py:870-879 / 913-922: corners (xmax,ymin),(xmin,ymin),(xmin,ymax),(xmax,ymax)."""
    xmin, ymin, xmax, ymax = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    return torch.stack([xmax, ymin, xmin, ymin, xmin, ymax, xmax, ymax], dim=1).view(-1, 4, 2)


def quadrangle(boxes):
    """dsnake_head.py:501-517: edge mid-points top, left, bottom, right."""
    x_min, y_min, x_max, y_max = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    q = [(x_min + x_max) / 2.0, y_min, x_min, (y_min + y_max) / 2.0,
         (x_min + x_max) / 2.0, y_max, x_max, (y_min + y_max) / 2.0]
    return torch.stack(q, dim=1).view(-1, 4, 2)


def octagon_polygon(ex):
    """snake_head.py:796-830 / dsnake_head.py:373-409: (N,4,2) extreme points -> (N,12,2)."""
    w, h = ex[:, 3, 0] - ex[:, 1, 0], ex[:, 2, 1] - ex[:, 0, 1]
    t, l, b, r = ex[:, 0, 1], ex[:, 1, 0], ex[:, 2, 1], ex[:, 3, 0]
    x = 8.0
    o = [
        ex[:, 0, 0], ex[:, 0, 1],
        torch.max(ex[:, 0, 0] - w / x, l), ex[:, 0, 1],
        ex[:, 1, 0], torch.max(ex[:, 1, 1] - h / x, t),
        ex[:, 1, 0], ex[:, 1, 1],
        ex[:, 1, 0], torch.min(ex[:, 1, 1] + h / x, b),
        torch.max(ex[:, 2, 0] - w / x, l), ex[:, 2, 1],
        ex[:, 2, 0], ex[:, 2, 1],
        torch.min(ex[:, 2, 0] + w / x, r), ex[:, 2, 1],
        ex[:, 3, 0], torch.min(ex[:, 3, 1] + h / x, b),
        ex[:, 3, 0], ex[:, 3, 1],
        ex[:, 3, 0], torch.max(ex[:, 3, 1] - h / x, t),
        torch.min(ex[:, 0, 0] + w / x, r), ex[:, 0, 1],
    ]
    return torch.stack(o, dim=1).view(-1, 12, 2)


def get_octagon(ex, p_num):
    """snake_head.py:796-832."""
    if len(ex) == 0:
        return torch.zeros([0, p_num, 2]), None
    return uniform_upsample(octagon_polygon(ex)[None], p_num)


def fit_to_box(ext):
    """core/structures/pointset.py:132-142 (in place on a (N,4,2) tensor)."""
    box = torch.stack([ext[:, 1, 0], ext[:, 0, 1], ext[:, 3, 0], ext[:, 2, 1]], dim=1)
    n = box.size(0)
    lower = box.view(-1, 2, 2)[:, :1, :]
    upper = box.view(-1, 2, 2)[:, 1:, :]
    lo, hi = ext < lower, ext > upper
    if lo.any():
        ext[lo] = lower.expand(n, 4, 2)[lo]
    if hi.any():
        ext[hi] = upper.expand(n, 4, 2)[hi]
    return ext


def contour_upsample_2x(locations):
    """snake_head.py:1752-1761."""
    new = locations.new_empty((locations.size(0), 2 * locations.size(1), locations.size(2)))
    new[:, ::2, :] = locations
    shifted = torch.cat([locations[:, 1:, :], locations[:, :1, :]], dim=1)
    new[:, 1::2, :] = (locations + shifted) / 2
    return new


def bottom_out(sd_all, features, groups=32, eps=1e-5):
    """RefineNet.bottom_out, snake_head.py:562-615 / 1816-1820: Conv2d(3x3, padding 1, no bias)
    -> GroupNorm(32) -> ReLU per layer (detectron2's Conv2d wrapper: conv, norm, activation).
    sd_all holds bottom_out.{i}.weight, bottom_out.{i}.norm.{weight,bias}."""
    import torch.nn.functional as F
    x, i = features, 0
    while "bottom_out.%d.weight" % i in sd_all:
        x = F.conv2d(x, sd_all["bottom_out.%d.weight" % i], None, padding=1)
        x = F.group_norm(x, groups, sd_all["bottom_out.%d.norm.weight" % i], sd_all["bottom_out.%d.norm.bias" % i], eps)
        x = F.relu(x)
        i += 1
    return x


def attender(sd_all, x, prefix="attender.", eps=1e-5):
    """EdgeSnakeFPNHead.attender, edge_det.py:66-100, applied as at 202-204 to x = 1 - pred_edge
    (B,1,H,W): Conv2d(1->32, 3x3, pad 1, no bias) -> GroupNorm(4,32) -> ReLU, the same 32->32,
    then Conv2d(32->1, 3x3, pad 1, bias) -> sigmoid.  Returns the attention map (B,1,H,W)."""
    import torch.nn.functional as F
    for i in (0, 1):
        x = F.conv2d(x, sd_all[prefix + "%d.weight" % i], None, padding=1)
        x = F.group_norm(x, 4, sd_all[prefix + "%d.norm.weight" % i], sd_all[prefix + "%d.norm.bias" % i], eps)
        x = F.relu(x)
    x = F.conv2d(x, sd_all[prefix + "2.weight"], sd_all[prefix + "2.bias"], padding=1)
    return torch.sigmoid(x)


def edge_head_forward(sd, features, boxes, image_size, num_sampling, in_features=("p2", "p3", "p4", "p5"),
                      strides=(4, 8, 16, 32), common_stride=4):
    """EdgeSnakeFPNHead.forward, inference, one image, shipped configuration (edge_det.py:187-204,
    313-315, 387-395 -> RefineNet.single_test snake_head.py:1763-1869 with INITIAL "box"):
    semantic-FPN scale heads summed -> predictor -> sigmoid -> attender(1 - edge) -> bottom_out ->
    3 attentive deformation iterations.  sd = the head's state-dict; features = {"p2": ...};
    boxes (N,4).  Returns (att_map (1,1,Hf,Wf), [contours (N,V,2)] * 3)."""
    import math
    import torch.nn.functional as F

    def cna(x, prefix):
        x = F.conv2d(x, sd[prefix + "weight"], None, padding=1)
        return F.relu(F.group_norm(x, 32, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-5))

    x = None
    for f, stride in zip(in_features, strides):
        y, idx = features[f], 0
        for _ in range(max(1, int(math.log2(stride) - math.log2(common_stride)))):
            y = cna(y, "%s.%d." % (f, idx))
            idx += 1
            if stride != common_stride:
                y = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=False)
                idx += 1
        x = y if x is None else x + y
    logits = F.conv2d(cna(x, "predictor.0."), sd["predictor.1.weight"], sd["predictor.1.bias"])
    att_map = attender(sd, 1 - torch.sigmoid(logits))
    rsd = sub_state_dict(sd, "refine_head.")
    boxes = boxes.clone()
    h, w = image_size
    init, _ = uniform_upsample(box_polygon(boxes)[None], num_sampling)
    boxes[:, 0::2].clamp_(0, w)
    boxes[:, 1::2].clamp_(0, h)
    whs = torch.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], dim=1)
    feats = torch.cat([att_map, bottom_out(rsd, x)], dim=1)
    return att_map, dance_deform(rsd, feats, init, None, image_size, whs)


# ------------------------------------------------------------------------------- gather
def get_locations_feature(features, locations, image_idx, padding_mode):
    """snake_head.py:1542-1578 (padding_mode="reflection") and dsnake_head.py:670-695
    ("zeros").  features (B,C,Hf,Wf), locations (N,V,2) image px, image_idx (N,) float
    tensor or None -> (N,C,V)."""
    h = features.shape[2] * STRIDE
    w = features.shape[3] * STRIDE
    locations = locations.clone()
    locations[..., 0] = locations[..., 0] / (w / 2.0) - 1
    locations[..., 1] = locations[..., 1] / (h / 2.0) - 1
    out = torch.zeros([locations.size(0), features.size(1), locations.size(1)], dtype=features.dtype)
    for i in range(features.size(0)):
        loc_i = locations.unsqueeze(0) if image_idx is None else locations[image_idx == i].unsqueeze(0)
        f = F.grid_sample(features[i:i + 1], loc_i, padding_mode=padding_mode,
                          align_corners=False)[0].permute(1, 0, 2)
        if image_idx is None:
            out = f
        else:
            out[image_idx == i] = f
    return out


def bilinear_sample_closed_form(plane_chw, xy, padding_mode):
    """Closed form of F.grid_sample(bilinear, align_corners=False) in image pixels, used
    to state the kernel's arithmetic; checked against
    get_locations_feature in tests.  plane_chw (C,Hf,Wf), xy (P,2) image px -> (C,P)."""
    C, Hf, Wf = plane_chw.shape
    w, h = Wf * STRIDE, Hf * STRIDE
    gx = xy[:, 0] / (w / 2.0) - 1
    gy = xy[:, 1] / (h / 2.0) - 1
    fx = ((gx + 1) * Wf - 1) / 2
    fy = ((gy + 1) * Hf - 1) / 2

    def reflect(t, size):
        # reflect about -0.5 and size-0.5 (twice_low=-1, twice_high=2*size-1), then clip
        lo, span = -0.5, float(size)
        t = (t - lo).abs()
        extra = torch.fmod(t, span)
        flips = torch.floor(t / span)
        t = torch.where(flips % 2 == 0, extra + lo, span - extra + lo)
        return t.clamp(0, size - 1)

    if padding_mode == "reflection":
        fx, fy = reflect(fx, Wf), reflect(fy, Hf)
    x0, y0 = torch.floor(fx), torch.floor(fy)
    out = torch.zeros((C, xy.size(0)))
    for dy in (0, 1):
        for dx in (0, 1):
            xi, yi = x0 + dx, y0 + dy
            wx = (fx - x0) if dx else (x0 + 1 - fx)
            wy = (fy - y0) if dy else (y0 + 1 - fy)
            ok = (xi >= 0) & (xi <= Wf - 1) & (yi >= 0) & (yi <= Hf - 1)
            xi_c, yi_c = xi.clamp(0, Wf - 1).long(), yi.clamp(0, Hf - 1).long()
            out += plane_chw[:, yi_c, xi_c] * (wx * wy * ok)[None]
    return out


# -------------------------------------------------------------------------- de_location
def de_location(locations, kind):
    """snake_head.py:1580-1614 (derange / demean / demin); dsnake_head.py:697-708 = demin."""
    x_min = torch.min(locations[..., 0], dim=-1)[0]
    y_min = torch.min(locations[..., 1], dim=-1)[0]
    x_max = torch.max(locations[..., 0], dim=-1)[0]
    y_max = torch.max(locations[..., 1], dim=-1)[0]
    new = locations.clone()
    if kind == "derange":
        new[..., 0] = (new[..., 0] - x_min[..., None]) / (x_max[..., None] - x_min[..., None])
        new[..., 1] = (new[..., 1] - y_min[..., None]) / (y_max[..., None] - y_min[..., None])
    elif kind == "demean":
        new[..., 0] = 2.0 * (new[..., 0] - x_min[..., None]) / (x_max[..., None] - x_min[..., None]) - 1.0
        new[..., 1] = 2.0 * (new[..., 1] - y_min[..., None]) / (y_max[..., None] - y_min[..., None]) - 1.0
    elif kind == "demin":
        new[..., 0] = new[..., 0] - x_min[..., None]
        new[..., 1] = new[..., 1] - y_min[..., None]
    else:
        raise ValueError("Invalid operation!", kind)
    return new


# ---------------------------------------------------------------------------- snake net
def circ_conv(x, weight, bias, dilation, n_adj=4):
    """DilatedCircularConv, snake_head.py:92-116."""
    if n_adj != 0:
        x = torch.cat([x[..., -n_adj * dilation:], x, x[..., :n_adj * dilation]], dim=2)
    return F.conv1d(x, weight, bias, dilation=dilation)


def snake_block(x, sd, prefix, dilation):
    """SnakeBlock, snake_head.py:140-153: conv -> ReLU -> BatchNorm1d (eval statistics)."""
    x = circ_conv(x, sd[prefix + "conv.fc.weight"], sd[prefix + "conv.fc.bias"], dilation)
    x = F.relu(x)
    return F.batch_norm(x, sd[prefix + "norm.running_mean"], sd[prefix + "norm.running_var"],
                        sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], False, 0.0, BN_EPS)


def snake_net(x, sd, dilations=DILATIONS, return_states=False):
    """_SnakeNet.forward, snake_head.py:260-284 (= dsnake_head.py:119-136)."""
    states = []
    x = snake_block(x, sd, "head.", 1)
    states.append(x)
    for i, d in enumerate(dilations):
        x = snake_block(x, sd, "res%d." % i, d) + x
        states.append(x)
    state = torch.cat(states, dim=1)
    back_out = F.conv1d(state, sd["fusion.weight"], sd["fusion.bias"])
    global_state = torch.max(back_out, dim=2, keepdim=True)[0]
    global_state = global_state.expand(global_state.size(0), global_state.size(1), state.size(2))
    full = torch.cat([global_state, state], dim=1)
    y = F.relu(F.conv1d(full, sd["prediction.0.weight"], sd["prediction.0.bias"]))
    y = F.relu(F.conv1d(y, sd["prediction.2.weight"], sd["prediction.2.bias"]))
    y = F.conv1d(y, sd["prediction.4.weight"], sd["prediction.4.bias"])
    if return_states:
        return y, state
    return y


def sub_state_dict(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


# ---------------------------------------------------------------------- DANCE iteration
def clip_locations(pred_locs, image_idx, image_sizes):
    """snake_head.py:1715-1727, including its observable behaviour: with image_idx None
    only instance 0 is clamped; with a tensor image_idx the masked indexing makes a copy
    and nothing changes."""
    if image_idx is None:
        pred_locs[0, :, 0::2].clamp_(min=0, max=image_sizes[1] - 1)
        pred_locs[0, :, 1::2].clamp_(min=0, max=image_sizes[0] - 1)
    else:
        for i, size in enumerate(image_sizes):
            pred_locs[image_idx == i, :, 0::2].clamp_(min=0, max=size[1] - 1)
            pred_locs[image_idx == i, :, 1::2].clamp_(min=0, max=size[0] - 1)


def dance_evolve(sd, features, locations, image_idx, image_sizes, whs, att,
                 attention=True, individual_scale=True, de_loc_type="derange",
                 dilations=DILATIONS):
    """RefineNet.evolve, snake_head.py:1624-1713, for conv_type "ccn", no pre_offset."""
    sampled = get_locations_feature(features, locations, image_idx, "reflection")
    if attention:
        att_scores = sampled[:, :1, :]
        sampled = sampled[:, 1:, :]
    calibrated = de_location(locations, de_loc_type)
    x = torch.cat([sampled, calibrated.permute(0, 2, 1)], dim=1)
    off = snake_net(x, sd, dilations).permute(0, 2, 1)
    if individual_scale:
        off = torch.tanh(off) * whs[:, None, :]
    if att:
        off = off * att_scores.permute(0, 2, 1)
    pred = locations + off
    clip_locations(pred, image_idx, image_sizes)
    return pred


def dance_deform(sd_all, features, locations, image_idx, image_sizes, whs, num_iter=3,
                 attention=True, individual_scale=True, de_loc_type="derange",
                 dilations=None):
    """Iteration driver, snake_head.py:1822-1862 (single image) / 1969-1995 (batched):
    `features` is the (B,1+C,Hf,Wf) tensor after bottom_out with the attention map back
    in channel 0; iteration i uses deformer{i}; attention re-weighting from i>=1."""
    preds, cur = [], locations
    for i in range(num_iter):
        sd = sub_state_dict(sd_all, "deformer%d." % i)
        d = DILATIONS if dilations is None else dilations[i]
        cur = dance_evolve(sd, features, cur, image_idx, image_sizes, whs,
                           att=(attention and i > 0), attention=attention,
                           individual_scale=individual_scale, de_loc_type=de_loc_type,
                           dilations=d)
        preds.append(cur)
    return preds


# --------------------------------------------------------------------------- Dsnake path
def dsnake_init(sd_all, features, locations, image_idx):
    """DsnakeHead.init, dsnake_head.py:718-746 -> (N,4,2) predicted extreme points."""
    sampled = get_locations_feature(features, locations, image_idx, "zeros")
    center = (torch.min(locations, dim=1)[0] + torch.max(locations, dim=1)[0]) * 0.5
    center_feat = get_locations_feature(features, center[:, None], image_idx, "zeros")
    init_feat = torch.cat([sampled, center_feat.expand_as(sampled)], dim=1)
    init_feat = F.conv1d(init_feat, sd_all["fuse.weight"], sd_all["fuse.bias"])
    calibrated = de_location(locations, "demin")
    x = torch.cat([init_feat, calibrated.permute(0, 2, 1)], dim=1)
    off = snake_net(x, sub_state_dict(sd_all, "init_snake.")).permute(0, 2, 1)
    return (locations + off)[:, ::10]


def dsnake_evolve(sd, features, locations, image_idx):
    """DsnakeHead.evolve, dsnake_head.py:748-765."""
    sampled = get_locations_feature(features, locations, image_idx, "zeros")
    calibrated = de_location(locations, "demin")
    x = torch.cat([sampled, calibrated.permute(0, 2, 1)], dim=1)
    return locations + snake_net(x, sd).permute(0, 2, 1)


def dsnake_deform(sd_all, features, boxes_per_image, num_sampling=128, num_iter=3):
    """DsnakeHead.forward inference branch after bottom_out, dsnake_head.py:783-830.
    boxes_per_image: list of (n_i,4) xyxy tensors.  Returns dict with init contours,
    predicted extreme points (after clamping), octagon contours and per-iteration
    predictions."""
    init, idx = [], []
    for i, bx in enumerate(boxes_per_image):
        init.append(uniform_upsample(quadrangle(bx)[None], 40)[0])     # :438-451
        idx.append(torch.full((len(bx),), float(i)))
    init, image_idx = torch.cat(init), torch.cat(idx)
    exts = dsnake_init(sd_all, features, init, image_idx)
    h, w = features.shape[2] * STRIDE, features.shape[3] * STRIDE
    octs, exts_clamped = [], []
    for i in range(len(boxes_per_image)):
        e = exts[image_idx == i]
        e[..., 0] = torch.clamp(e[..., 0], min=0, max=w - 1)            # :800-805
        e[..., 1] = torch.clamp(e[..., 1], min=0, max=h - 1)
        exts_clamped.append(e)
        octs.append(get_octagon(e, num_sampling)[0])
    locations = torch.cat(octs)
    preds, cur = [], locations
    for i in range(num_iter):
        cur = dsnake_evolve(sub_state_dict(sd_all, "deformer%d." % i), features, cur, image_idx)
        preds.append(cur)
    return {"init": init, "image_idx": image_idx, "exts": torch.cat(exts_clamped),
            "octagons": locations, "preds": preds}
