"""dance_fuse16_raw + dance_tail16 (fusion + ring max in parts, then row constant -> ReLU ->
prediction[2] -> prediction[4] -> offset update in one launch) against the sequence of separate
launches they replace (dance_conv16 x 3, dance_row_constant, dance_offset_update) and against
plain torch fp32 -- reference: _SnakeNet.forward snake_head.py:269-282, evolve tail 1702-1727."""
import ctypes

import pytest
import torch

from dance_amd import ops, snake
from dance_amd._lib import check, current_stream, lib, ptr
from _util import dance_state_dict

pytestmark = pytest.mark.gpu


def _separate(s16, state, loc, whs, att, n, v, terms, clip):
    """The launches the fused pair replaces (the path rings above 128 vertices took before)."""
    dev = state.device
    slot = 256 * v
    gmax = torch.empty((n, 256), device=dev)
    h = torch.empty((n, 256), device=dev)
    p1 = torch.zeros((n, 2, 2, 16, v, 8), dtype=torch.float16, device=dev)
    p2 = torch.empty((n, 64, v), device=dev)
    out = torch.empty((n, v, 2), device=dev)
    with torch.cuda.device(dev):
        st = current_stream()
        check(lib().dance_conv16(ptr(state), state.stride(0), slot, s16.fuse_w_split, s16.fuse_bs, s16.fuse_winv,
                                 None, None, None, 0, 0, 1, None, 0, 0, ptr(gmax), 0, n, v, 1024, 256, 1, 1,
                                 terms, st), "fusion")
        check(lib().dance_row_constant(ptr(gmax), s16.pred0_wg, s16.pred0_b, ptr(h), n, 256, 256, st), "rowc")
        check(lib().dance_conv16(ptr(state), state.stride(0), slot, s16.pred0_w_split, None, s16.fuse_winv,
                                 None, None, ptr(h), 0, 1, 0, ptr(p1), p1.stride(0), slot, None, 0, n, v,
                                 1024, 256, 1, 1, terms, st), "pred0")
        check(lib().dance_conv16(ptr(p1), p1.stride(0), slot, s16.pred1_w, s16.pred1_bs, s16.pred1_winv, None,
                                 None, None, 0, 1, 0, None, 0, 0, ptr(p2), 64 * v, n, v, 256, 64, 1, 1,
                                 terms, st), "pred1")
        check(lib().dance_offset_update(ptr(p2), 64 * v, s16.pred2_w, s16.pred2_b, ptr(loc), ptr(whs), ptr(att),
                                        ptr(out), n, v, 64, int(clip), 300.0, 200.0, st), "update")
    return gmax, out


def _fused(s16, state, loc, whs, att, n, v, terms, clip):
    dev = state.device
    slot = 256 * v
    parts = lib().dance_fuse16_parts(n, v)
    gpart = torch.empty((n, parts, 256), device=dev)
    p1 = torch.zeros((n, 2, 2, 16, v, 8), dtype=torch.float16, device=dev)
    out = torch.empty((n, v, 2), device=dev)
    with torch.cuda.device(dev):
        st = current_stream()
        check(lib().dance_fuse16_raw(ptr(state), state.stride(0), slot, s16.fuse_w, s16.fuse_bs, s16.fuse_winv,
                                     ptr(p1), p1.stride(0), slot, ptr(gpart), n, v, terms, st), "fuse16_raw")
        check(lib().dance_tail16(ptr(p1), p1.stride(0), slot, ptr(gpart), parts, s16.pred0_wg, s16.pred0_b,
                                 s16.pred1_w, s16.pred1_bs, s16.pred1_winv, s16.pred2_w, s16.pred2_b, ptr(loc),
                                 ptr(whs), ptr(att), ptr(out), n, v, int(clip), 300.0, 200.0, terms, st), "tail16")
    return gpart.max(dim=1)[0], out


def _inputs(n, v, device, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n, 1024, v), generator=g)
    state = ops.to_p16(x.to(device))
    loc = (torch.rand((n, v, 2), generator=g) * 300).to(device)
    whs = (torch.rand((n, 2), generator=g) * 80 + 10).to(device)
    att = torch.rand((n, v), generator=g).to(device)
    return x, state, loc, whs, att


@pytest.mark.parametrize("v", [16, 40, 64, 68, 96, 100, 128, 132, 160, 164, 192, 196, 224, 228, 256])
def test_fuse_raw_tail_equals_separate_launches(device, v):
    cfg, sd, _ = dance_state_dict(31, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer1."), device)
    s16 = net.struct16
    for n, terms, full in [(7, 3, True), (7, 1, True), (5, 3, False)]:
        x, state, loc, whs, att = _inputs(n, v, device, v + n)
        loc[0] += 1000.0                                    # contour 0 leaves the image: the clip shows
        if not full:
            whs = att = None
        gm_a, out_a = _fused(s16, state, loc, whs, att, n, v, terms, full)
        gm_b, out_b = _separate(s16, state, loc, whs, att, n, v, terms, full)
        tol = 1e-5 if terms == 3 else 5e-3                  # summation order (one term: fp16 operands)
        assert float((gm_a - gm_b).abs().max()) <= tol * float(gm_b.abs().max()), (v, terms)
        off_a, off_b = out_a[1:] - loc[1:], out_b[1:] - loc[1:]
        scale = float(off_b.abs().max())
        assert scale > 0 and bool(torch.isfinite(out_a).all())
        assert float((off_a - off_b).abs().max()) <= 10 * tol * scale, (v, terms, full)
        if full:                                            # clip_locations: contour 0 only
            assert float(out_a[0, :, 0].max()) <= 300.0 and float(out_a[0, :, 1].max()) <= 200.0
        assert torch.equal(out_a[0], out_b[0]) or float((out_a[0] - out_b[0]).abs().max()) <= 10 * tol * scale


@pytest.mark.parametrize("v", [128, 196])
def test_fuse_raw_tail_many_contours_vs_torch(device, v, monkeypatch):
    """1100 contours through the 4-contours-per-workgroup tail kernel (DANCE_TAIL16_GROUP=4); checked
    against torch fp32 evaluated from the same state (the chain of k = 1 convolutions as the
    reference writes it)."""
    monkeypatch.setenv("DANCE_TAIL16_GROUP", "4")
    cfg, sd, _ = dance_state_dict(32, 128)
    sub = snake.sub_state_dict(sd, "deformer2.")
    net = snake.SnakeNetWeights(sub, device)
    n = 1100
    x, state, loc, whs, att = _inputs(n, v, device, v)
    _, out = _fused(net.struct16, state, loc, whs, att, n, v, 3, False)
    xd = ops.from_p16(state)                                # what the kernels see (22-bit inputs)
    f = torch.nn.functional
    w = {k: t.to(device).float() for k, t in sub.items()}
    gstate = f.conv1d(xd, w["fusion.weight"], w["fusion.bias"]).max(dim=2, keepdim=True)[0]
    p = torch.cat([gstate.expand(-1, -1, v), xd], dim=1)
    p = torch.relu(f.conv1d(p, w["prediction.0.weight"], w["prediction.0.bias"]))
    p = torch.relu(f.conv1d(p, w["prediction.2.weight"], w["prediction.2.bias"]))
    p = f.conv1d(p, w["prediction.4.weight"], w["prediction.4.bias"])
    off = torch.tanh(p.permute(0, 2, 1)) * whs[:, None, :] * att[:, :, None]
    want = loc + off
    got_off = out - loc
    assert float((got_off - off).abs().max()) <= 2e-4 * float(off.abs().max())
    assert float((out - want).abs().max()) <= 1e-4 * float(want.abs().max())


def test_tail16_group_sizes_agree(device, monkeypatch):
    """Contours 4, 2 and 1 per workgroup (DANCE_TAIL16_GROUP; the default is 2 from 1024 contours on, 1
    below) give the same bits: the row constant's summation order does not depend on the grouping."""
    cfg, sd, _ = dance_state_dict(33, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer0."), device)
    v, n = 100, 1027                                        # 1027 = 256 groups of 4 + 3 = 513 pairs + 1
    x, state, loc, whs, att = _inputs(n, v, device, 5)
    outs = {}
    for g in ("4", "2", "1"):
        monkeypatch.setenv("DANCE_TAIL16_GROUP", g)
        _, outs[g] = _fused(net.struct16, state, loc, whs, att, n, v, 3, False)
    monkeypatch.delenv("DANCE_TAIL16_GROUP")
    _, dflt = _fused(net.struct16, state, loc, whs, att, n, v, 3, False)
    assert torch.equal(outs["4"], outs["1"]) and torch.equal(outs["2"], outs["1"]) and torch.equal(dflt, outs["1"])
    k = 300
    _, small = _fused(net.struct16, state[-k:].contiguous(), loc[-k:].contiguous(), whs[-k:].contiguous(),
                      att[-k:].contiguous(), k, v, 3, False)
    assert torch.equal(outs["4"][-k:], small)



@pytest.mark.parametrize("v", [128, 196])
def test_fuse_raw_tail_reruns_are_bit_identical_with_late_workgroups(device, v):
    """3200 contours = more workgroups than the chip holds at once, so many start while others
    are in their MFMA phase.  A GEMV the compiler had paired into v_pk_fma_f32 gave run-to-run
    different row constants in exactly those workgroups (tail16.hip, fma1); reruns must agree to
    the bit, and a contour's result must not depend on where in the launch it sits."""
    cfg, sd, _ = dance_state_dict(34, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer0."), device)
    n = 3200
    x, state, loc, whs, att = _inputs(n, v, device, 2)
    outs = [_fused(net.struct16, state, loc, whs, att, n, v, 3, False)[1] for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    k = 640                                                 # the last 640 contours alone, all started early
    _, alone = _fused(net.struct16, state[-k:].contiguous(), loc[-k:].contiguous(), whs[-k:].contiguous(),
                      att[-k:].contiguous(), k, v, 3, False)
    assert torch.equal(outs[0][-k:], alone)


@pytest.mark.parametrize("v", [40, 128, 196, 256])
def test_fuse_raw_four_wave_split_equals_eight_wave(device, v, monkeypatch):
    """DANCE_FUSE16_WAVES=4: the 512 outputs as two 4-wave workgroups (64-channel chunks staged by
    LDS-direct loads).  Same products in the same order per output element: identical bits."""
    cfg, sd, _ = dance_state_dict(35, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer2."), device)
    n = 300
    x, state, loc, whs, att = _inputs(n, v, device, 11)
    res = {}
    for waves in ("8", "4"):
        monkeypatch.setenv("DANCE_FUSE16_WAVES", waves)
        res[waves] = [_fused(net.struct16, state, loc, whs, att, n, v, terms, False) for terms in (3, 1)]
    for a, b in zip(res["8"], res["4"]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
