"""GPU: the one entry point that allocates nothing (`dance_nms_sorted_plain_static`, scratch in a
file-scope device array) against the entry point it is a variant of (`dance_nms_sorted_plain`,
scratch in a caller workspace), through the C ABI: two calls in a row, the second with a smaller
input, bit-exact (index outputs).  Named test_zz_* so that pytest collects it last."""
import pytest
import torch

pytestmark = pytest.mark.gpu

STATIC_MAX = 4096


def _bind():
    from dance_amd import _lib

    h = _lib.lib()
    return h, h.dance_nms_sorted_plain_static


def _sorted_rows(n, seed, dev):
    """(n,5) x1,y1,x2,y2,score in descending score order: clustered boxes so that many overlap."""
    g = torch.Generator().manual_seed(seed)
    centres = torch.rand((max(1, n // 8), 2), generator=g) * 600
    c = centres[torch.randint(0, centres.size(0), (n,), generator=g)] + torch.randn((n, 2), generator=g) * 6
    wh = 20 + torch.rand((n, 2), generator=g) * 60
    scores = torch.rand((n,), generator=g)
    scores[::7] = scores[0]                                   # ties
    rows = torch.cat([c - wh / 2, c + wh / 2, scores[:, None]], 1)
    return rows[torch.sort(scores, descending=True, stable=True)[1]].contiguous().to(dev)


def _with_workspace(h, rows, thresh):
    from dance_amd._lib import check, current_stream, ptr

    n = rows.size(0)
    ws = torch.empty(((n + 1) * ((n + 63) // 64),), dtype=torch.int64, device=rows.device)
    keep = torch.full((max(n, 1),), -1, dtype=torch.int64, device=rows.device)
    count = torch.full((1,), -1, dtype=torch.int32, device=rows.device)
    check(h.dance_nms_sorted_plain(ptr(rows), n, thresh, ptr(ws), ptr(keep), ptr(count), current_stream()), "nms")
    return keep, count


def _static(fn, rows, thresh):
    from dance_amd._lib import check, current_stream, ptr

    n = rows.size(0)
    keep = torch.full((max(n, 1),), -1, dtype=torch.int64, device=rows.device)
    count = torch.full((1,), -1, dtype=torch.int32, device=rows.device)
    check(fn(ptr(rows), n, thresh, ptr(keep), ptr(count), current_stream()), "nms_static")
    return keep, count


def test_static_scratch_nms_equals_workspace_nms(device):
    h, fn = _bind()
    # two calls in a row on the same GPU's array, the second smaller (it must not see the
    # first call's words), then the maximum size, one box and the empty list
    for n, seed in ((3000, 1), (700, 2), (STATIC_MAX, 3), (65, 4), (1, 5), (0, 6)):
        rows = _sorted_rows(n, seed, device) if n else torch.empty((0, 5), device=device)
        k0, c0 = _with_workspace(h, rows, 0.6)
        k1, c1 = _static(fn, rows, 0.6)
        torch.cuda.synchronize()
        m = int(c0.item())
        assert int(c1.item()) == m and (n == 0 or 0 < m <= n)
        assert torch.equal(k1[:m], k0[:m]), "static-scratch NMS differs at n=%d" % n
        if n >= 65:
            assert m < n                                      # the fixture does suppress boxes


def test_static_scratch_nms_rejects_lists_beyond_its_array(device):
    h, fn = _bind()
    rows = _sorted_rows(STATIC_MAX + 1, 9, device)
    keep = torch.empty((STATIC_MAX + 1,), dtype=torch.int64, device=device)
    count = torch.zeros((1,), dtype=torch.int32, device=device)
    rc = fn(rows.data_ptr(), STATIC_MAX + 1, 0.6, keep.data_ptr(), count.data_ptr(), None)
    assert rc == 1 and b"static scratch" in h.dance_last_error()


def test_static_scratch_nms_inside_a_hip_graph(device):
    """What the entry point is for: a caller under stream capture.  After one warm-up call outside the
    capture (the header's rule: the first call on a device looks the array's address up), the call is
    captured into a HIP graph -- nothing in it allocates or synchronises, or the capture would fail --
    and the replays, on new box coordinates written into the same input buffer and an empty list via the
    n = 0 path, give what the workspace form gives."""
    h, fn = _bind()
    n = 900
    rows = _sorted_rows(n, 11, device)
    _static(fn, rows, 0.6)                                    # warm-up outside the capture
    torch.cuda.synchronize()
    keep = torch.full((n,), -1, dtype=torch.int64, device=device)
    count = torch.full((1,), -1, dtype=torch.int32, device=device)
    from dance_amd._lib import check, current_stream, ptr
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        check(fn(ptr(rows), n, 0.6, ptr(keep), ptr(count), current_stream()), "nms_static (captured)")
    for seed in (11, 12, 13):
        rows.copy_(_sorted_rows(n, seed, device))             # same buffer, new boxes
        k0, c0 = _with_workspace(h, rows, 0.6)
        _static(fn, _sorted_rows(300, seed + 50, device), 0.6)   # an eager call in between: the array is shared
        g.replay()
        torch.cuda.synchronize()
        m = int(c0.item())
        assert int(count.item()) == m and torch.equal(keep[:m], k0[:m])
