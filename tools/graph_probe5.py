"""graph replay of ContourDeformer.deform vs eager: float vs int32 image index (diagnostic)"""
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

for kind in ("int32", "float", "float-keep"):
    idx = (torch.arange(24) % 2).to(device)
    idx = idx.to(torch.int32) if kind == "int32" else idx.float()
    loc = loc0.clone()
    d = snake.ContourDeformer(nets)
    keep = []
    if kind == "float-keep":
        orig = ops.nchw_to_nhwc
        ops.nchw_to_nhwc = lambda f, a: (keep.append(orig(f, a)), keep[-1])[1]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            d.deform(feats, loc, idx, [(200, 304)] * 2, whs)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = d.deform(feats, loc, idx, [(200, 304)] * 2, whs)
    res = []
    for delta in (0.0, 1.5, 0.25):
        loc.add_(delta)
        g.replay()
        torch.cuda.synchronize()
        e = snake.ContourDeformer(nets).deform(feats, loc, idx, [(200, 304)] * 2, whs)
        res.append([torch.equal(a, b) for a, b in zip(out, e)])
    print(kind, "->", res)
    if kind == "float-keep":
        ops.nchw_to_nhwc = orig
