"""graph replay of deform(): what happens to the graph's absmax words when eager runs follow an input change (diagnostic)"""
import sys
import torch
sys.path.insert(0, "tests")
from dance_amd import ops, snake, synthetic
from _util import dance_state_dict

device = torch.device("cuda:0")
cfg, sd, _ = dance_state_dict(1234, 128)
nets = [snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer%d." % i), device) for i in range(3)]
feats = synthetic.make_smooth_features(2, 256, 50, 76, 21).to(device)
boxes = synthetic.make_boxes(24, 200, 304, 22, r_min=8, r_max=60)
loc0 = synthetic.make_circle_contours(boxes, 128).to(device)
whs = torch.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1).to(device)
idx = (torch.arange(24) % 2).float().to(device)
orig = ops.nchw_to_nhwc

for keep_what in ("nothing", "words", "feat+att"):
    loc = loc0.clone()
    keep, ptrs = [], []

    def spy(f, a):
        r = orig(f, a)
        w = getattr(r[0], ops.ABSMAX_ATTR)
        ptrs.append((w.data_ptr(), r[0].data_ptr(), r[1].data_ptr()))
        if keep_what == "words":
            keep.append(w)
        elif keep_what == "feat+att":
            keep.append(r)
        return r

    ops.nchw_to_nhwc = spy
    d = snake.ContourDeformer(nets)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            d.deform(feats, loc, idx, [(200, 304)] * 2, whs)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    del keep[:], ptrs[:]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = d.deform(feats, loc, idx, [(200, 304)] * 2, whs)
    gptr = ptrs[0]
    res = []
    for op in "R E A E R R".split():
        if op == "A":
            loc.add_(1.5)
            continue
        if op == "R":
            g.replay()
            torch.cuda.synchronize()
            e = snake.ContourDeformer(nets).deform(feats, loc, idx, [(200, 304)] * 2, whs)
            res.append(("R", all(torch.equal(a, b) for a, b in zip(out, e))))
        else:
            n0 = len(ptrs)
            d.deform(feats, loc, idx, [(200, 304)] * 2, whs)
            res.append(("E ptrs==graph's", [a == b for a, b in zip(ptrs[n0], gptr)]))
    print("keep", keep_what, res)
    ops.nchw_to_nhwc = orig
