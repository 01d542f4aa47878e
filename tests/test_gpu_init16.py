"""Dsnake's init stage as one kernel sequence of the fast precision mode (dance_snake_init16:
k_init_head16 = gather + centre feature + fuse Conv1d + de_location + head block, then stack /
fusion / tail) against the separate launches it replaces (two gathers, row constant, `fuse` as an
exact fp32 convolution, dance_xin_head16) and against plain torch fp32 -- reference:
DsnakeHead.init dsnake_head.py:718-746."""
import pytest
import torch

from dance_amd import modules, ops
from oracle import snake_ref
from _util import dsnake_state_dict

pytestmark = pytest.mark.gpu


def _head(device, seed=7):
    cfg, sd, _ = dsnake_state_dict(seed)
    head = modules.DsnakeHead(cfg)
    head.load_state_dict(sd)
    head = head.to(device).eval()
    return head, head._get_deformer()


def _rings(n, v, hw, device, seed):
    """n boxes as v-point quadrangle rings (uniform_upsample), some partly outside the image."""
    g = torch.Generator().manual_seed(seed)
    h, w = hw
    x0 = torch.rand(n, generator=g) * w * 0.9 - w * 0.05
    y0 = torch.rand(n, generator=g) * h * 0.9 - h * 0.05
    bw = torch.rand(n, generator=g) * w * 0.4 + 6
    bh = torch.rand(n, generator=g) * h * 0.4 + 6
    boxes = torch.stack([x0, y0, x0 + bw, y0 + bh], dim=1).to(device)
    rings, _ = ops.uniform_upsample(boxes, v, kind=ops.POLY_QUAD)
    return rings


@pytest.mark.parametrize("v", [16, 40, 64])
def test_one_launch_init_equals_separate_launches(device, v):
    head, d = _head(device)
    g = torch.Generator().manual_seed(v)
    b, hf, wf = 3, 24, 40
    feat = torch.randn((b, hf, wf, 256), generator=g).to(device)
    n = 37
    rings = _rings(n, v, (hf * 4, wf * 4), device, v)
    idx = torch.randint(0, b, (n,), generator=g).to(torch.int32).to(device)
    got = head.init_contours(d, feat, rings, idx, one_launch=True)
    want = head.init_contours(d, feat, rings, idx, one_launch=False)
    off_g, off_w = got - rings, want - rings
    scale = float(off_w.abs().max())
    assert scale > 0 and bool(torch.isfinite(got).all())
    assert float((off_g - off_w).abs().max()) <= 1e-4 * scale          # `fuse`: fp16x3 vs exact fp32
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_one_launch_init_vs_oracle(device):
    """The reference's op sequence restated on the CPU (oracle/snake_ref.py: grid_sample with zeros
    padding, cat with the expanded centre feature, fuse, de_location, init snake), whole ring."""
    cfg, sd, _ = dsnake_state_dict(9)
    head = modules.DsnakeHead(cfg)
    head.load_state_dict(sd)
    head = head.to(device).eval()
    d = head._get_deformer()
    g = torch.Generator().manual_seed(3)
    b, hf, wf, v, n = 2, 20, 28, 40, 25
    feat = torch.randn((b, hf, wf, 256), generator=g)
    rings = _rings(n, v, (hf * 4, wf * 4), device, 4)
    idx = (torch.arange(n) % b).to(torch.int32)
    got = head.init_contours(d, feat.to(device), rings, idx.to(device), one_launch=True).cpu()
    loc = rings.cpu()
    nchw = feat.permute(0, 3, 1, 2).contiguous()
    sampled = snake_ref.get_locations_feature(nchw, loc, idx, "zeros")
    center = (torch.min(loc, dim=1)[0] + torch.max(loc, dim=1)[0]) * 0.5
    cfeat = snake_ref.get_locations_feature(nchw, center[:, None], idx, "zeros")
    x = torch.nn.functional.conv1d(torch.cat([sampled, cfeat.expand_as(sampled)], dim=1), sd["fuse.weight"],
                                   sd["fuse.bias"])
    x = torch.cat([x, snake_ref.de_location(loc, "demin").permute(0, 2, 1)], dim=1)
    off = snake_ref.snake_net(x, snake_ref.sub_state_dict(sd, "init_snake.")).permute(0, 2, 1)
    want = loc + off
    assert float((got - loc - off).abs().max()) <= 2e-4 * float(off.abs().max())
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())


def test_one_launch_init_reruns_and_position_independence(device):
    """3200 contours (BASELINE config 4's init stage): more workgroups than the chip holds at once.
    Reruns agree to the bit, and a contour's result does not depend on where in the launch it sits
    (the row-constant GEMV runs beside other workgroups' MFMAs -- mfma16.h, fma1)."""
    head, d = _head(device, 11)
    g = torch.Generator().manual_seed(5)
    b, hf, wf, v, n = 32, 24, 32, 40, 3200
    feat = torch.randn((b, hf, wf, 256), generator=g).to(device)
    rings = _rings(n, v, (hf * 4, wf * 4), device, 6)
    idx = (torch.arange(n) // 100).to(torch.int32).to(device)
    outs = [head.init_contours(d, feat, rings, idx, one_launch=True).clone() for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    k = 500
    alone = head.init_contours(d, feat, rings[-k:].contiguous(), idx[-k:].contiguous(), one_launch=True)
    assert torch.equal(outs[0][-k:], alone)
    want = head.init_contours(d, feat, rings, idx, one_launch=False)
    assert float((outs[0] - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_one_launch_init_image_index_out_of_range(device):
    """A contour whose image index is outside the batch samples zeros (as the separate gather
    kernels do): both paths agree."""
    head, d = _head(device, 13)
    g = torch.Generator().manual_seed(8)
    feat = torch.randn((2, 16, 16, 256), generator=g).to(device)
    rings = _rings(6, 40, (64, 64), device, 9)
    idx = torch.tensor([0, 1, 5, -1, 1, 0], dtype=torch.int32, device=device)
    got = head.init_contours(d, feat, rings, idx, one_launch=True)
    want = head.init_contours(d, feat, rings, idx, one_launch=False)
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_init_evolve_rejects_unsupported_shapes_and_accepts_empty(device):
    """dance_snake_init16: rings above 64 vertices / other channel counts are refused by the wrapper
    (the module then takes the separate launches); an empty batch returns an empty result."""
    head, d = _head(device, 15)
    feat = torch.randn((1, 16, 16, 256)).to(device)
    assert not d.supports_init_evolve(256, 68) and not d.supports_init_evolve(128, 40)
    assert not d.supports_init_evolve(256, 42)
    assert d.supports_init_evolve(256, 40) == (d.precision == "fp16x3")    # DANCE_PRECISION=fp32: separate launches
    rings = _rings(3, 128, (64, 64), device, 2)
    idx = torch.zeros(3, dtype=torch.int32, device=device)
    with pytest.raises(RuntimeError):
        d.init_evolve(0, feat, rings, idx, head._fuse16)
    empty = d.init_evolve(0, feat, rings[:0, :40].contiguous(), idx[:0], head._fuse16)
    assert empty.shape == (0, 40, 2)
    # the C entry itself refuses what the wrapper filters (no launch happens)
    from dance_amd._lib import check, current_stream, lib, ptr
    import ctypes
    ws = torch.empty(lib().dance_snake_workspace_bytes(3, 128), dtype=torch.uint8, device=device)
    out = torch.empty((3, 128, 2), device=device)
    with torch.cuda.device(device), pytest.raises(RuntimeError):
        check(lib().dance_snake_init16(ctypes.byref(d.nets[0].struct16), ctypes.byref(head._fuse16), ptr(feat),
                                       ptr(rings), ptr(idx), ptr(out), ptr(ws), ws.numel(), 3, 128, 1, 16, 16,
                                       ops.PAD_ZEROS, ops.DELOC["demin"], 3, None, None, None, current_stream()), "init16")
