"""dance_fuse16w_raw + dance_tail16 (fusion + ring max in parts, then row constant -> ReLU ->
prediction[2] -> prediction[4] -> offset update in one launch) against plain torch fp32 evaluated
from the same state -- the chain of k = 1 convolutions as the reference writes it: _SnakeNet.forward
snake_head.py:269-282, evolve tail 1702-1727."""
import ctypes

import pytest
import torch

from dance_amd import snake
from dance_amd._lib import PolicyStruct, check, current_stream, lib, ptr
from _fast_ref import scaled_state, tail_ref, true_state, w32
from _util import dance_state_dict

pytestmark = pytest.mark.gpu


def _fused(s16, state, loc, whs, att, n, v, terms, clip, policy=None):
    dev = state.device
    slot = 256 * v
    pol = None if policy is None else ctypes.byref(policy)
    parts = lib().dance_fuse16_parts(n, v, pol)
    gpart = torch.empty((n, parts, 256), device=dev)
    p1 = torch.zeros((n, 2, 2, 16, v, 8), dtype=torch.float16, device=dev)
    out = torch.empty((n, v, 2), device=dev)
    with torch.cuda.device(dev):
        st = current_stream()
        check(lib().dance_fuse16w_raw(ptr(state), state.stride(0), slot, s16.fuse_w, s16.fuse_bs, s16.fuse_winv,
                                      s16.pred0_winv, ptr(p1), p1.stride(0), slot, ptr(gpart), n, v, parts, terms,
                                      st), "fuse16w_raw")
        check(lib().dance_tail16(ptr(p1), p1.stride(0), slot, ptr(gpart), parts, s16.pred0_wg, s16.pred0_b,
                                 s16.pred1_w, s16.pred1_bs, s16.pred1_winv, s16.pred2_w, s16.pred2_b, ptr(loc),
                                 ptr(whs), ptr(att), ptr(out), n, v, int(clip), 300.0, 200.0, terms, pol, st),
              "tail16")
    return gpart.max(dim=1)[0], out


def _inputs(net, n, v, device, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n, 1024, v), generator=g)
    state = scaled_state(net, x.to(device))
    loc = (torch.rand((n, v, 2), generator=g) * 300).to(device)
    whs = (torch.rand((n, 2), generator=g) * 80 + 10).to(device)
    att = torch.rand((n, v), generator=g).to(device)
    return state, loc, whs, att


@pytest.mark.parametrize("v", [16, 40, 64, 68, 96, 100, 128, 132, 160, 164, 192, 196, 224, 228, 256])
def test_fuse_tail_vs_torch(device, v):
    cfg, sd, _ = dance_state_dict(31, 128)
    sub = snake.sub_state_dict(sd, "deformer1.")
    net = snake.SnakeNetWeights(sub, device)
    w = w32(sub, device)
    for n, terms, full in [(7, 3, True), (7, 1, True), (5, 3, False)]:
        state, loc, whs, att = _inputs(net, n, v, device, v + n)
        loc[0] += 1000.0                                    # contour 0 leaves the image: the clip shows
        if not full:
            whs = att = None
        _, out = _fused(net.struct16, state, loc, whs, att, n, v, terms, full)
        want, off = tail_ref(w, true_state(net, state), loc, whs, att)
        tol = 2e-4 if terms == 3 else 5e-2                  # of the offsets' scale (one term: fp16 operands)
        scale = float(off.abs().max())
        assert scale > 0 and bool(torch.isfinite(out).all())
        assert float(((out - loc)[1:] - off[1:]).abs().max()) <= tol * scale, (v, terms, full)
        if full:                                            # clip_locations: contour 0 only
            assert float(out[0, :, 0].max()) <= 300.0 and float(out[0, :, 1].max()) <= 200.0
            clipped = torch.stack([want[0, :, 0].clamp(0, 300.0), want[0, :, 1].clamp(0, 200.0)], dim=1)
            assert float((out[0] - clipped).abs().max()) <= tol * scale
        else:
            assert float((out[0] - want[0]).abs().max()) <= tol * scale


@pytest.mark.parametrize("v", [128, 196])
def test_fuse_tail_many_contours_vs_torch(device, v):
    """1100 contours through the 4-contours-per-workgroup tail kernel (policy.tail_group = 4)"""
    cfg, sd, _ = dance_state_dict(32, 128)
    sub = snake.sub_state_dict(sd, "deformer2.")
    net = snake.SnakeNetWeights(sub, device)
    n = 1100
    state, loc, whs, att = _inputs(net, n, v, device, v)
    _, out = _fused(net.struct16, state, loc, whs, att, n, v, 3, False, PolicyStruct(tail_group=4))
    want, off = tail_ref(w32(sub, device), true_state(net, state), loc, whs, att)
    got_off = out - loc
    assert float((got_off - off).abs().max()) <= 2e-4 * float(off.abs().max())
    assert float((out - want).abs().max()) <= 1e-4 * float(want.abs().max())


def test_tail16_group_sizes_agree(device):
    """Contours 4, 2 and 1 per workgroup (policy.tail_group; the default is 2 from 1024 contours on, 1
    below) give the same bits: the row constant's summation order does not depend on the grouping."""
    cfg, sd, _ = dance_state_dict(33, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer0."), device)
    v, n = 100, 1027                                        # 1027 = 256 groups of 4 + 3 = 513 pairs + 1
    state, loc, whs, att = _inputs(net, n, v, device, 5)
    outs = {}
    for g in (4, 2, 1):
        _, outs[g] = _fused(net.struct16, state, loc, whs, att, n, v, 3, False, PolicyStruct(tail_group=g))
    _, dflt = _fused(net.struct16, state, loc, whs, att, n, v, 3, False)
    assert torch.equal(outs[4], outs[1]) and torch.equal(outs[2], outs[1]) and torch.equal(dflt, outs[1])
    k = 300
    _, small = _fused(net.struct16, state[-k:].contiguous(), loc[-k:].contiguous(), whs[-k:].contiguous(),
                      att[-k:].contiguous(), k, v, 3, False)
    assert torch.equal(outs[4][-k:], small)


@pytest.mark.parametrize("v", [128, 196])
def test_fuse_tail_reruns_are_bit_identical_with_late_workgroups(device, v):
    """3200 contours = more workgroups than the chip holds at once, so many start while others
    are in their MFMA phase.  A GEMV the compiler had paired into v_pk_fma_f32 gave run-to-run
    different row constants in exactly those workgroups (tail16.hip, fma1); reruns must agree to
    the bit, and a contour's result must not depend on where in the launch it sits."""
    cfg, sd, _ = dance_state_dict(34, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer0."), device)
    n = 3200
    state, loc, whs, att = _inputs(net, n, v, device, 2)
    outs = [_fused(net.struct16, state, loc, whs, att, n, v, 3, False)[1] for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    k = 640                                                 # the last 640 contours alone, all started early
    _, alone = _fused(net.struct16, state[-k:].contiguous(), loc[-k:].contiguous(), whs[-k:].contiguous(),
                      att[-k:].contiguous(), k, v, 3, False)
    assert torch.equal(outs[0][-k:], alone)
