"""Training-side target matching, NumPy half, on the GPU: one workgroup per
instance instead of the reference's per-instance Python / NumPy loop on the host
(`RefineNet.compute_targets_for_polys`, core/modeling/edge_snake/snake_head.py:982-1254).

Function names, argument meaning and results follow the reference's; every function takes a LIST
of instances (or a batch tensor) and returns device tensors.  Float results are float64, what
NumPy 2.2 promotes the reference's lines to, and bit-identical to the reference's on the same
inputs (tests/golden/g12_targets.npz).  The Shapely-dependent steps of the reference (orientation
test of the ground-truth polygon) are not part of this module.
"""
import torch

from ._lib import check, current_stream, lib, ptr

MAX_POINTS = 2048


def _ragged(arrays, dtype, device):
    """list of (k_i, 2) arrays -> (sum k, 2) tensor on the device + offsets (n + 1) int32."""
    ts = [torch.as_tensor(a, dtype=dtype).reshape(-1, 2) for a in arrays]
    off = torch.zeros(len(ts) + 1, dtype=torch.int32)
    if ts:
        off[1:] = torch.cumsum(torch.tensor([len(t) for t in ts]), 0)
    flat = torch.cat(ts) if ts else torch.zeros((0, 2), dtype=dtype)
    return flat.to(device).contiguous(), off.to(device), [len(t) for t in ts]


def _status(status, what):
    bad = (status == 2).nonzero()
    if len(bad):
        raise RuntimeError("%s: instance %d is outside the kernel's limits (%d points)"
                           % (what, int(bad[0]), MAX_POINTS))


def uniform_sample(polygons, newpnums, device="cuda"):
    """RefineNet.uniform_sample (snake_head.py:661-725) for a list of (P_i, 2) float32 polygons and
    target point counts.  Returns a list of (newpnum_i, 2) tensors: float32 rows of the polygon
    where P_i > newpnum_i (points removed), float64 otherwise."""
    dev = torch.device(device)
    flat, off, lens = _ragged(polygons, torch.float32, dev)
    n = len(lens)
    newp = torch.as_tensor(list(newpnums), dtype=torch.int32)
    ooff = torch.zeros(n + 1, dtype=torch.int64)
    ooff[1:] = torch.cumsum(newp.to(torch.int64), 0)
    out = torch.empty((int(ooff[-1]), 2), dtype=torch.float64, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    newp_d, ooff_d = newp.to(dev), ooff.to(dev)          # named: they must outlive the launch's enqueue
    with torch.cuda.device(dev):
        check(lib().dance_uniform_sample(ptr(flat), ptr(off), ptr(newp_d), ptr(ooff_d), ptr(out),
                                         ptr(status), n, current_stream()), "uniform_sample")
    _status(status, "uniform_sample")
    res = []
    for i in range(n):
        piece = out[int(ooff[i]):int(ooff[i + 1])]
        res.append(piece.float() if lens[i] > int(newp[i]) else piece)
    return res


def dense_targets(polygons, first_points, num_points, device="cuda"):
    """snake_head.py:1134-1144 for a list of ground-truth polygons: uniform_sample(cnt, len(cnt) *
    num_points), rolled to start at the point nearest the initial contour's first vertex, every
    len(cnt)-th point kept.  first_points (n, 2) float32.  Returns (n, num_points, 2) float64."""
    dev = torch.device(device)
    flat, off, lens = _ragged(polygons, torch.float32, dev)
    n = len(lens)
    first = torch.as_tensor(first_points, dtype=torch.float32).reshape(n, 2).to(dev).contiguous()
    out = torch.empty((n, num_points, 2), dtype=torch.float64, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib().dance_dense_targets(ptr(flat), ptr(off), ptr(first), num_points, ptr(out), ptr(status), n,
                                        current_stream()), "dense_targets")
    _status(status, "dense_targets")
    return out


def uniform_sample_1d(polylines, new_ns, device="cuda"):
    """RefineNet.uniform_sample_1d (snake_head.py:728-762) for a list of open poly-lines (n_i, 2)
    float64.  Returns (list of (new_n_i, 2) float64 tensors, status (n) int32: 1 where the
    reference raises -- a poly-line of zero length)."""
    dev = torch.device(device)
    flat, off, lens = _ragged(polylines, torch.float64, dev)
    n = len(lens)
    nn = torch.as_tensor(list(new_ns), dtype=torch.int32)
    ooff = torch.zeros(n + 1, dtype=torch.int64)
    ooff[1:] = torch.cumsum(nn.to(torch.int64), 0)
    out = torch.zeros((int(ooff[-1]), 2), dtype=torch.float64, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    nn_d, ooff_d = nn.to(dev), ooff.to(dev)
    with torch.cuda.device(dev):
        check(lib().dance_uniform_sample_1d(ptr(flat), ptr(off), ptr(nn_d), ptr(ooff_d), ptr(out),
                                            ptr(status), n, current_stream()), "uniform_sample_1d")
    _status(status, "uniform_sample_1d")
    return [out[int(ooff[i]):int(ooff[i + 1])] for i in range(n)], status


def get_aux_extreme_points(contours, cap=64):
    """get_aux_extreme_points (core/modeling/fcose/utils.py:94-189) for a batch of closed contours
    (n, N, 2) float64.  Returns (re_ordered (n, N, 2), list of sorted unique index tensors)."""
    pts = torch.as_tensor(contours)
    if not pts.is_cuda:
        raise RuntimeError("get_aux_extreme_points: CUDA tensor expected")
    pts = pts.to(torch.float64).contiguous()
    n, num, _ = pts.shape
    re = torch.empty_like(pts)
    ext = torch.zeros((n, cap), dtype=torch.int64, device=pts.device)
    count = torch.zeros(n, dtype=torch.int32, device=pts.device)
    status = torch.zeros(n, dtype=torch.int32, device=pts.device)
    with torch.cuda.device(pts.device):
        check(lib().dance_aux_extreme_points(ptr(pts), ptr(re), ptr(ext), ptr(count), ptr(status), n, num, cap,
                                             current_stream()), "get_aux_extreme_points")
    if bool((status != 0).any()):
        raise RuntimeError("get_aux_extreme_points: more than %d auxiliary extreme points" % cap)
    c = count.tolist()
    re.ext_idx_padded, re.ext_count = ext, count                 # the batch form the matching takes
    return re, [ext[i, :c[i]] for i in range(n)]


def single_uniform_segment_matching(dense_targets_, sampled_pts, edge_idx, up_rate):
    """RefineNet.single_uniform_segment_matching (snake_head.py:1265-1384, INITIAL = "octagon")
    for a batch: dense_targets (n, V * up_rate, 2) float64, sampled_pts (n, V, 2) float32,
    edge_idx (n, 12) int64.  Returns (targets (n, V, 2) float64, status (n): 1 where the reference
    takes its fall-back, reorder_perloss of dense_targets[::up_rate])."""
    dense = torch.as_tensor(dense_targets_)
    if not dense.is_cuda:
        raise RuntimeError("single_uniform_segment_matching: CUDA tensors expected")
    dense = dense.to(torch.float64).contiguous()
    sampled = torch.as_tensor(sampled_pts).to(dense.device, torch.float32).contiguous()
    eidx = torch.as_tensor(edge_idx).to(dense.device, torch.int64).contiguous()
    n, v, _ = sampled.shape
    if dense.shape != (n, v * up_rate, 2) or eidx.dim() != 2 or eidx.size(0) != n or eidx.size(1) < 10:
        raise RuntimeError("single_uniform_segment_matching: shapes do not match")
    out = torch.empty((n, v, 2), dtype=torch.float64, device=dense.device)
    status = torch.zeros(n, dtype=torch.int32, device=dense.device)
    with torch.cuda.device(dense.device):
        check(lib().dance_segment_matching(ptr(dense), ptr(sampled), ptr(eidx), ptr(out), ptr(status), n, v,
                                           up_rate, eidx.size(1), current_stream()), "segment_matching")
    return out, status


def single_uniform_multisegment_matching(dense_reordered, sampled_pts, ext_idx, ext_count, up_rate):
    """RefineNet.single_uniform_multisegment_matching (snake_head.py:1386-1498, the shipped
    INITIAL = "box" + NEW_MATCHING form) for a batch, including the roll of the initial contour
    that precedes it (1150-1155).  dense_reordered (n, V * up_rate, 2) float64 and ext_idx
    (n, cap) int64 / ext_count (n) int32 as get_aux_extreme_points leaves them (`.ext_idx_padded`,
    `.ext_count` of its first result); sampled_pts (n, V, 2) float32.
    Returns (rolled contour (n, V, 2) float32, targets (n, V, 2) float64, status (n))."""
    dense = torch.as_tensor(dense_reordered)
    if not dense.is_cuda:
        raise RuntimeError("single_uniform_multisegment_matching: CUDA tensors expected")
    dense = dense.to(torch.float64).contiguous()
    sampled = torch.as_tensor(sampled_pts).to(dense.device, torch.float32).contiguous()
    ext = torch.as_tensor(ext_idx).to(dense.device, torch.int64).contiguous()
    cnt = torch.as_tensor(ext_count).to(dense.device, torch.int32).contiguous()
    n, v, _ = sampled.shape
    if dense.shape != (n, v * up_rate, 2) or ext.dim() != 2 or ext.size(0) != n or cnt.shape != (n,):
        raise RuntimeError("single_uniform_multisegment_matching: shapes do not match")
    rolled = torch.empty_like(sampled)
    out = torch.empty((n, v, 2), dtype=torch.float64, device=dense.device)
    status = torch.zeros(n, dtype=torch.int32, device=dense.device)
    with torch.cuda.device(dense.device):
        check(lib().dance_multisegment_matching(ptr(dense), ptr(sampled), ptr(ext), ptr(cnt), ptr(rolled), ptr(out),
                                                ptr(status), n, v, up_rate, ext.size(1), current_stream()),
              "multisegment_matching")
    if bool((status == 2).any()):
        raise RuntimeError("single_uniform_multisegment_matching: an instance has fewer than 2 (the reference "
                           "fails there too) or more than 64 auxiliary extreme points")
    return rolled, out, status
