"""graph replay vs eager: which nodes make the replay deviate after the inputs change? (diagnostic)"""
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
idx = (torch.arange(24) % 2).to(torch.int32).to(device)
pre = ops.nchw_to_nhwc(feats, 1)


def chain(d, loc, iters, prep_inside):
    f = d.prepare_features(feats) if prep_inside else pre
    outs, cur = [], loc
    for it in range(iters):
        cur = d.evolve(it, *f, cur, idx, whs, use_att=it > 0, clip_first=False, clip_max=(0.0, 0.0))
        outs.append(cur)
    return outs


for iters, prep_inside in ((1, True), (2, True), (3, True), (3, False), (2, False)):
    loc = loc0.clone()
    d = snake.ContourDeformer(nets)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            chain(d, loc, iters, prep_inside)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = chain(d, loc, iters, prep_inside)
    res = []
    for delta in (0.0, 1.5, 0.25):
        loc.add_(delta)
        g.replay()
        torch.cuda.synchronize()
        e = chain(snake.ContourDeformer(nets), loc, iters, prep_inside)
        res.append([torch.equal(a, b) for a, b in zip(out, e)])
    print("iterations", iters, "layout pass inside the graph", prep_inside, "->", res)
