"""Plain torch fp32 restatements of the fast mode's kernels on the GPU tensors they see, for the
kernel-level tests (the whole path is checked against the CPU oracle and the golden vectors
elsewhere), plus the conversions between true values and the scaled P16 blocks the kernels
exchange (csrc/p16.h "Activation scales": state slot l is stored as x * 2^(a_l), prediction[0]'s
output as x * 2^(a_p); `SnakeNetWeights.act_exps`)."""
import torch

from dance_amd import ops

F = torch.nn.functional


def slot_scales(net, device):
    """(8,) tensor 2^(a_l)"""
    return torch.tensor([2.0 ** a for a in net.act_exps["slots"]], dtype=torch.float32, device=device)


def scaled_state(net, x):
    """true state (n, 1024, v) fp32 -> the (n, 8, 2, 16, v, 8) P16 block the kernels read"""
    n, c, v = x.shape
    sc = slot_scales(net, x.device).repeat_interleave(128)[None, :, None]
    return ops.to_p16((x * sc).contiguous())


def true_state(net, p16):
    """inverse of scaled_state (also for a block whose leading slots only are filled)"""
    x = ops.from_p16(p16)
    sc = slot_scales(net, x.device).repeat_interleave(128)[None, :x.shape[1], None]
    return x / sc


def true_p1(net, p16):
    return ops.from_p16(p16) / (2.0 ** net.act_exps["pred0"])


def w32(sub_sd, device):
    return {k: t.to(device).float() for k, t in sub_sd.items() if t.is_floating_point()}


def head_ref(w, nhwc, att, loc, idx, pad, deloc):
    """gather (the exact-fp32 kernel, itself pinned by golden g2) -> circular Conv1d k = 9 -> ReLU ->
    BatchNorm (eval): (n, 128, v)"""
    sampled, att_s = ops.gather_features(nhwc, att, loc, idx, pad, deloc)
    x = torch.cat([sampled[:, :, -4:], sampled, sampled[:, :, :4]], dim=2)
    wt = w["head.conv.fc.weight"]
    y = torch.relu(F.conv1d(x, wt[:, :sampled.shape[1]], w["head.conv.fc.bias"]))
    y = F.batch_norm(y, w["head.norm.running_mean"], w["head.norm.running_var"], w["head.norm.weight"],
                     w["head.norm.bias"], False, 0.0, 1e-5)
    return y, att_s


def res_block_ref(w, i, x, dilation):
    """x + BN(relu(CircConv(x))) of residual block i"""
    pad = 4 * dilation
    xp = torch.cat([x[:, :, -pad:], x, x[:, :, :pad]], dim=2)
    y = torch.relu(F.conv1d(xp, w["res%d.conv.fc.weight" % i], w["res%d.conv.fc.bias" % i], dilation=dilation))
    y = F.batch_norm(y, w["res%d.norm.running_mean" % i], w["res%d.norm.running_var" % i],
                     w["res%d.norm.weight" % i], w["res%d.norm.bias" % i], False, 0.0, 1e-5)
    return y + x


def fuse_ref(w, xd):
    """(max over the ring of the fusion output (n, 256), prediction[0]'s state part (n, 256, v):
    no bias, no row constant, no ReLU)"""
    g = F.conv1d(xd, w["fusion.weight"], w["fusion.bias"])
    p = F.conv1d(xd, w["prediction.0.weight"][:, 256:])
    return g, p


def tail_ref(w, xd, loc, whs, att):
    """the net from the state on as the reference writes it, then evolve's tail: (n, v, 2)"""
    v = xd.shape[2]
    gstate = F.conv1d(xd, w["fusion.weight"], w["fusion.bias"]).max(dim=2, keepdim=True)[0]
    p = torch.cat([gstate.expand(-1, -1, v), xd], dim=1)
    p = torch.relu(F.conv1d(p, w["prediction.0.weight"], w["prediction.0.bias"]))
    p = torch.relu(F.conv1d(p, w["prediction.2.weight"], w["prediction.2.bias"]))
    off = F.conv1d(p, w["prediction.4.weight"], w["prediction.4.bias"]).permute(0, 2, 1)
    if whs is not None:
        off = torch.tanh(off) * whs[:, None, :]
    if att is not None:
        off = off * att[:, :, None]
    return loc + off, off
