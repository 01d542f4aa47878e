"""The widened steps in sequence on one 800x1216 image (the reference's inference mode is one
image per GPU, dance.py:52): FCOSE head outputs -> FCOSEOutputs.predict_proposals (decode + NMS)
-> EdgeSnakeFPNHead (scale heads, edge predictor, attender, bottom_out, 3 snake iterations)
-> get_polygon_rles.  Wall time per stage, synchronised after each (so the sum is an upper
bound of the pipelined time).  Synthetic head outputs with ~100 planted objects per image."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from _util import edge_head_cfg  # noqa: E402
from dance_amd import modules, rle, synthetic  # noqa: E402
from dance_amd.fcose_outputs import FCOSEOutputs, compute_locations  # noqa: E402

dev = torch.device("cuda:0")
H, W, C, K = 800, 1216, 80, 100
SHAPES, STRIDES = [(100, 152), (50, 76), (25, 38), (13, 19), (7, 10)], [8, 16, 32, 64, 128]


def head_outputs(seed):
    g = torch.Generator().manual_seed(seed)
    logits = [torch.randn(1, C, h, w, generator=g) * 0.5 - 8.0 for h, w in SHAPES]
    ctr = [torch.randn(1, 1, h, w, generator=g) for h, w in SHAPES]
    reg = [torch.rand(1, 4, h, w, generator=g) + 0.5 for h, w in SHAPES]
    ext = [torch.randn(1, 4, h, w, generator=g) * 0.2 for h, w in SHAPES]
    h, w = SHAPES[0]
    ys = torch.randint(h // 8, h - h // 8, (K,), generator=g)
    xs = torch.randint(w // 8, w - w // 8, (K,), generator=g)
    cls = torch.randint(0, C, (K,), generator=g)
    logits[0][0, cls, ys, xs] = 3.0 + torch.rand(K, generator=g)
    ctr[0][0, 0, ys, xs] = 2.0
    reg[0][0, :, ys, xs] = (torch.rand(4, K, generator=g) * 80 + 20) / STRIDES[0]
    up = lambda lst: [x.to(dev) for x in lst]
    return up(logits), up(reg), up(ext), up(ctr)


class Images:
    image_sizes = [(H, W)]

    def __len__(self):
        return 1


cfg, shapes = edge_head_cfg(num_sampling=128, num_convs=2)
head = modules.EdgeSnakeFPNHead(cfg, shapes)
sd = {k: v.clone() for k, v in head.state_dict().items()}
synthetic.fill_state_dict_(sd, 1234)
head.load_state_dict(sd)
head = head.to(dev).eval()
feats = {k: synthetic.make_smooth_features(1, 256, H // s, W // s, 5 + i, attention=False, cell=4).to(dev)
         for i, (k, s) in enumerate((("p2", 4), ("p3", 8), ("p4", 16), ("p5", 32)))}
logits, reg, ext, ctr = head_outputs(3)
locations = compute_locations(logits, STRIDES)


def stage(fn, *a):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn(*a)
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3


def decode():
    return FCOSEOutputs(Images(), locations, logits, reg, ext, ctr, 0.25, 2.0, None, None, True, None, STRIDES, 1.5, C,
                        0.05, 1000, 0.6, 100, True).predict_proposals()


def contours(insts):
    with torch.no_grad():
        return head(feats, insts, None)[2]


def masks(res):
    polys = res[0].pred_polys.tensor.clone()
    polys[..., 0].clamp_(0, W)
    polys[..., 1].clamp_(0, H)
    return rle.get_polygon_rles(polys.reshape(polys.size(0), -1), (H, W))


acc = [0.0, 0.0, 0.0]
REPS = 10
for it in range(3 + REPS):
    insts, t0 = stage(decode)
    res, t1 = stage(contours, insts)
    m, t2 = stage(masks, res)
    if it >= 3:
        acc = [a + t for a, t in zip(acc, (t0, t1, t2))]
n = len(insts[0])
print("1 image 800x1216, %d detections -> %d contours x 128 vertices -> %d RLE masks" % (n, len(res[0].pred_polys.tensor), len(m)))
print("predict_proposals %.2f ms | EdgeSnakeFPNHead forward %.2f ms | get_polygon_rles (incl. D2H + strings) %.2f ms | total %.2f ms"
      % (acc[0] / REPS, acc[1] / REPS, acc[2] / REPS, sum(acc) / REPS))

# the same head as ONE HIP-graph replay (dance_amd/graph.py; detections padded to 100)
from dance_amd import graph  # noqa: E402

gh = graph.GraphedEdgeSnakeFPNHead(head, max_instances=100)
boxes = insts[0].pred_boxes.tensor.clone()
with torch.no_grad():
    want = head(feats, [type(insts[0])((H, W), pred_boxes=type(insts[0].pred_boxes)(boxes.clone()))], None)[2][0].pred_polys.tensor
for _ in range(3):
    att_g, polys_g = gh(feats, boxes, (H, W))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(REPS):
    att_g, polys_g = gh(feats, boxes, (H, W))
torch.cuda.synchronize()
tg = (time.perf_counter() - t0) / REPS * 1e3
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e = next(iter(gh._graphs.values()))
ev0.record()
for _ in range(REPS):
    e["graph"].replay()
ev1.record()
torch.cuda.synchronize()
# a backbone that writes its pyramid levels straight into the graph's input buffers saves the copy
static = gh.input_buffers(feats, (H, W))
for f in static:
    static[f].copy_(feats[f])
for _ in range(3):
    gh(static, boxes, (H, W))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(REPS):
    att_s, polys_s = gh(static, boxes, (H, W))
torch.cuda.synchronize()
ts = (time.perf_counter() - t0) / REPS * 1e3
assert torch.equal(polys_s, polys_g)
diff = float((torch.nan_to_num(polys_g) - torch.nan_to_num(want)).abs().max())
print("graphed head (one HIP-graph replay, detections padded to 100): %.2f ms per call incl. the 82 MB input copy, "
      "%.2f ms per replay alone; max |contour - eager| %.2e px | total with it %.2f ms"
      % (tg, ev0.elapsed_time(ev1) / REPS, diff, acc[0] / REPS + tg + acc[2] / REPS))
print("the same with the features produced in the graph's input buffers (no copy): %.2f ms per call | total with it %.2f ms"
      % (ts, acc[0] / REPS + ts + acc[2] / REPS))
