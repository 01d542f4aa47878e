"""dance_fuse16w_raw (fusion + ring max in parts, and prediction[0]'s state part, on 16-wide vertex
tiles -- csrc/fuse16w.hip) against plain torch fp32 from the same weights and state -- reference:
_SnakeNet.forward snake_head.py:269-278."""
import ctypes

import pytest
import torch

from dance_amd import ops, snake
from dance_amd._lib import PolicyStruct, check, current_stream, lib, ptr
from _fast_ref import fuse_ref, scaled_state, true_p1, true_state, w32
from _util import dance_state_dict

pytestmark = pytest.mark.gpu


def _run(s16, state, n, v, parts, terms):
    """-> (gpart (n, parts, 256) f32, the stored prediction[0] state part as a P16 block)"""
    dev = state.device
    slot = 256 * v
    gpart = torch.full((n, parts, 256), float("nan"), device=dev)
    p1 = torch.zeros((n, 2, 2, 16, v, 8), dtype=torch.float16, device=dev)
    with torch.cuda.device(dev):
        check(lib().dance_fuse16w_raw(ptr(state), state.stride(0), slot, s16.fuse_w, s16.fuse_bs, s16.fuse_winv,
                                      s16.pred0_winv, ptr(p1), p1.stride(0), slot, ptr(gpart), n, v, parts, terms,
                                      current_stream()), "dance_fuse16w_raw")
    return gpart, p1


@pytest.mark.parametrize("n,v", [(7, 100), (7, 128), (300, 112), (300, 128), (9, 196), (200, 208), (9, 244), (150, 256),
                                 (300, 16), (300, 40), (9, 48), (300, 64), (300, 68), (300, 80), (9, 96), (9, 132), (300, 144), (9, 160), (300, 176), (300, 236),
                                 (200, 164), (9, 192), (9, 212), (200, 224), (9, 228), (9, 240),
                                 # whole rounds of the CUs: one persistent workgroup per CU loops over its
                                 # contours (next contour's first chunk staged behind the current one)
                                 (512, 128), (768, 100), (512, 196), (512, 256)])
def test_fuse16w_vs_torch(device, n, v):
    cfg, sd, _ = dance_state_dict(41, 128)
    sub = snake.sub_state_dict(sd, "deformer1.")
    net = snake.SnakeNetWeights(sub, device)
    w = w32(sub, device)
    g = torch.Generator().manual_seed(v + n)
    state = scaled_state(net, torch.randn((n, 1024, v), generator=g).to(device))
    xd = true_state(net, state)                                # what the kernels see (22-bit inputs)
    parts = lib().dance_fuse16_parts(n, v, None)
    assert lib().dance_fuse16w_supported(parts, v) == 1
    gw, pw = fuse_ref(w, xd)
    for terms in (3, 1):
        ga, pa = _run(net.struct16, state, n, v, parts, terms)
        pa = true_p1(net, pa)
        tol = 1e-5 if terms == 3 else 5e-3
        assert bool(torch.isfinite(ga).all()) and bool(torch.isfinite(pa).all())
        assert float((ga.max(dim=1)[0] - gw.max(dim=2)[0]).abs().max()) <= tol * float(gw.abs().max()), (v, terms)
        assert float((pa - pw).abs().max()) <= tol * float(pw.abs().max()), (v, terms)
        if parts == 2:                                          # each part's max over ITS vertices
            pv = (v + 15) // 16
            pv = ((pv + 1) // 2 * 16) if v > 128 else 64
            assert float((ga[:, 0] - gw[:, :, :pv].max(dim=2)[0]).abs().max()) <= tol * float(gw.abs().max())


def test_fuse16w_support_table(device):
    f = lib().dance_fuse16w_supported
    assert [f(1, v) for v in (12, 16, 96, 128, 129)] == [0, 1, 1, 1, 0]
    assert [f(2, v) for v in (96, 97, 128, 160, 256, 260)] == [0, 1, 1, 1, 1, 0]
    assert f(3, 128) == 0
    p = lib().dance_fuse16_parts
    off = PolicyStruct(few_contour_forms=0)
    assert [p(100, 128, None), p(129, 128, None), p(100, 96, None), p(5000, 132, None)] == [2, 1, 1, 2]
    assert p(100, 128, ctypes.byref(off)) == 1 and p(100, 132, ctypes.byref(off)) == 2


@pytest.mark.parametrize("n,v,n_split", [(300, 128, 44), (300, 100, 44), (9, 112, 9), (40, 97, 1), (300, 128, 300)])
def test_fuse16w_split_tail_equals_whole(device, n, v, n_split):
    """dance_fuse16w_raw_split (the last n_split contours as two half-ring workgroups each, the rest
    whole; gpart (n, 2, 256) with whole rings' max in both slots) against dance_fuse16w_raw with one
    part: prediction[0] state part bit-identical, the max over the two part slots bit-identical."""
    cfg, sd, _ = dance_state_dict(43, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer2."), device)
    s16 = net.struct16
    g = torch.Generator().manual_seed(v + n)
    x = torch.randn((n, 1024, v), generator=g)
    x[1, :, 3] = float("nan")                                  # NaN travels through the max as in torch
    state = scaled_state(net, x.to(device))
    slot = 256 * v
    for terms in (3, 1):
        gw, pw = _run(s16, state, n, v, 1, terms)
        gs = torch.full((n, 2, 256), float("nan"), device=device)
        p1 = torch.zeros((n, 2, 2, 16, v, 8), dtype=torch.float16, device=device)
        with torch.cuda.device(device):
            check(lib().dance_fuse16w_raw_split(ptr(state), state.stride(0), slot, s16.fuse_w, s16.fuse_bs,
                                                s16.fuse_winv, s16.pred0_winv, ptr(p1), p1.stride(0), slot, ptr(gs),
                                                n, v, n_split, terms, current_stream()), "fuse16w_raw_split")
        assert torch.equal(p1.view(torch.int16), pw.view(torch.int16))
        whole = gs[:n - n_split]
        assert torch.equal(torch.nan_to_num(whole[:, 0], nan=7.0), torch.nan_to_num(whole[:, 1], nan=7.0))
        m = torch.maximum(gs[:, 0], gs[:, 1])                  # torch.maximum propagates NaN
        assert torch.equal(torch.nan_to_num(m, nan=7.0), torch.nan_to_num(gw[:, 0], nan=7.0))


def test_fuse16w_tail_split_rule(device):
    """n mod CUs contours when there is more than one round and they fill at most half the CUs"""
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    f = lib().dance_fuse16w_tail_split
    with torch.cuda.device(device):
        assert f(cus, 128, None) == 0 and f(cus - 1, 128, None) == 0 and f(6 * cus, 128, None) == 0
        assert f(6 * cus + 64, 128, None) == 64 and f(6 * cus + cus // 2, 100, None) == cus // 2
        assert f(6 * cus + cus // 2 + 1, 128, None) == 0
        assert f(6 * cus + 64, 96, None) == 0 and f(6 * cus + 64, 132, None) == 0
        assert f(6 * cus + 64, 128, ctypes.byref(PolicyStruct(fuse_tail_split=0))) == 0


def test_evolve_with_split_tail_is_bit_identical(device):
    """one iteration over CUs + 44 contours: the evolve sequence with the fusion kernel's last round as
    half rings (default) against policy.fuse_tail_split = 0"""
    from dance_amd import synthetic
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    n, v, b, hf, wf = cus + 44, 128, 2, 40, 56
    cfg, sd, _ = dance_state_dict(44, v)
    nets = [snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer%d." % i), device) for i in range(3)]
    d = snake.ContourDeformer(nets)
    feats = synthetic.make_features(b, 256, hf, wf, 3).to(device)
    f = d.prepare_features(feats)
    boxes = synthetic.make_boxes(n, hf * 4, wf * 4, 5)
    whs = torch.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1).to(device)
    init, _ = ops.uniform_upsample(boxes.to(device), v, kind=ops.POLY_BOX)
    idx = (torch.arange(n) % b).to(torch.int32).to(device)
    with torch.cuda.device(device):
        assert lib().dance_fuse16w_tail_split(n, v, ctypes.byref(d.policy)) == 44
    a = d.evolve(1, *f, init, idx, whs, use_att=True, clip_first=False, clip_max=(0, 0))
    d.policy.fuse_tail_split = 0
    b_ = d.evolve(1, *f, init, idx, whs, use_att=True, clip_first=False, clip_max=(0, 0))
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b_)
