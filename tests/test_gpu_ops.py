"""Host-side behaviour of dance_amd.ops around the resampler's status assert (the reference's
`assert torch.all(edge_num_sum == p_num)`, snake_head.py:782) in its three modes."""
import pytest
import torch

from dance_amd import ops

pytestmark = pytest.mark.gpu


def test_lazy_status_check_raises_late(device, monkeypatch):
    """DANCE_SYNC_CHECKS=lazy: the resampler's edge-count assert (snake_head.py:782) is not waited for;
    flush_checks() -- or a later call once the status has arrived -- raises it.  "0" never raises."""
    monkeypatch.delenv("DANCE_SYNC_CHECKS", raising=False)
    ok = torch.tensor([[10.0, 10.0, 50.0, 60.0]], device=device)
    # a polygon of more edges than points to hand out cannot be resampled: status 1
    bad = (torch.rand((1, 40, 2), generator=torch.Generator().manual_seed(3)) * 100).to(device)
    with pytest.raises(AssertionError):
        ops.uniform_upsample(bad, 16, kind=ops.POLY_RAW)
    monkeypatch.setenv("DANCE_SYNC_CHECKS", "lazy")
    ops.flush_checks()
    ops.uniform_upsample(ok, 128, kind=ops.POLY_BOX)
    ops.flush_checks()                                        # nothing wrong: nothing raised
    ops.uniform_upsample(bad, 16, kind=ops.POLY_RAW)          # returns at once
    with pytest.raises(AssertionError):
        ops.flush_checks()
    ops.flush_checks()                                        # the list is empty again
    monkeypatch.setenv("DANCE_SYNC_CHECKS", "0")
    ops.uniform_upsample(bad, 16, kind=ops.POLY_RAW)
    ops.flush_checks()


def test_polygon_clip_finite_check_modes(device, monkeypatch):
    """PolygonPoints.clip's finite assert (pointset.py:247-251) in the three DANCE_SYNC_CHECKS modes."""
    from dance_amd.structures import PolygonPoints
    monkeypatch.delenv("DANCE_SYNC_CHECKS", raising=False)
    good = torch.rand((3, 128, 2), device=device) * 50
    bad = good.clone()
    bad[1, 7, 0] = float("nan")
    PolygonPoints(good.clone()).clip((40, 40))
    with pytest.raises(AssertionError):
        PolygonPoints(bad.clone()).clip((40, 40))
    monkeypatch.setenv("DANCE_SYNC_CHECKS", "lazy")
    ops.flush_checks()
    p = PolygonPoints(bad.clone())
    p.clip((40, 40))                                           # returns; the clamp has run
    assert float(p.tensor[0].max()) <= 40.0
    with pytest.raises(AssertionError, match="Polygon tensor"):
        ops.flush_checks()
    monkeypatch.setenv("DANCE_SYNC_CHECKS", "0")
    PolygonPoints(bad.clone()).clip((40, 40))
    ops.flush_checks()
