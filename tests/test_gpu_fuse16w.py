"""dance_fuse16w_raw (the raw fusion kernel on 16-wide vertex tiles, csrc/fuse16w.hip) against
dance_fuse16_raw (32-wide tiles): same gpart / prediction[0] state part up to summation order --
reference: _SnakeNet.forward snake_head.py:269-278."""
import pytest
import torch

from dance_amd import ops, snake
from dance_amd._lib import check, current_stream, lib, ptr
from _util import dance_state_dict

pytestmark = pytest.mark.gpu


def _run(fn, s16, state, n, v, parts, terms):
    dev = state.device
    slot = 256 * v
    gpart = torch.full((n, parts, 256), float("nan"), device=dev)
    p1 = torch.zeros((n, 2, 2, 16, v, 8), dtype=torch.float16, device=dev)
    extra = (parts,) if fn == "dance_fuse16w_raw" else ()
    with torch.cuda.device(dev):
        check(getattr(lib(), fn)(ptr(state), state.stride(0), slot, s16.fuse_w, s16.fuse_bs, s16.fuse_winv, ptr(p1),
                                 p1.stride(0), slot, ptr(gpart), n, v, *extra, terms, current_stream()), fn)
    return gpart, ops.from_p16(p1)


@pytest.mark.parametrize("n,v", [(7, 100), (7, 128), (300, 112), (300, 128), (9, 196), (200, 208), (9, 244), (150, 256),
                                 (300, 16), (300, 40), (9, 48), (300, 64), (300, 68), (9, 96), (9, 132), (9, 160),
                                 (200, 164), (9, 192), (9, 212), (200, 224), (9, 228), (9, 240)])
def test_fuse16w_equals_fuse16(device, n, v):
    cfg, sd, _ = dance_state_dict(41, 128)
    net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer1."), device)
    s16 = net.struct16
    g = torch.Generator().manual_seed(v + n)
    state = ops.to_p16(torch.randn((n, 1024, v), generator=g).to(device))
    parts = lib().dance_fuse16_parts(n, v)
    assert lib().dance_fuse16w_supported(parts, v) == 1
    for terms in (3, 1):
        ga, pa = _run("dance_fuse16w_raw", s16, state, n, v, parts, terms)
        gb, pb = _run("dance_fuse16_raw", s16, state, n, v, parts, terms)
        tol = 1e-5 if terms == 3 else 5e-3
        assert bool(torch.isfinite(ga).all()) and bool(torch.isfinite(pa).all())
        assert float((ga.max(dim=1)[0] - gb.max(dim=1)[0]).abs().max()) <= tol * float(gb.abs().max()), (v, terms)
        assert float((pa - pb).abs().max()) <= tol * float(pb.abs().max()), (v, terms)


def test_fuse16w_support_table(device):
    f = lib().dance_fuse16w_supported
    assert [f(1, v) for v in (12, 16, 96, 128, 129)] == [0, 1, 1, 1, 0]
    assert [f(2, v) for v in (96, 97, 128, 160, 256, 260)] == [0, 1, 1, 1, 1, 0]
    assert f(3, 128) == 0
