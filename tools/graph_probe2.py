"""graph replay vs eager: look at the absmax words and the layout pass's outputs inside the graph (diagnostic)"""
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
loc = synthetic.make_circle_contours(boxes, 128).to(device)
whs = torch.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1).to(device)
idx = (torch.arange(24) % 2).float().to(device)
d = snake.ContourDeformer(nets)
stash = []
orig = ops.nchw_to_nhwc


def spy(features, att_channels):
    r = orig(features, att_channels)
    stash.append(r)
    return r


ops.nchw_to_nhwc = spy


def step():
    return d.deform(feats, loc, idx, [(200, 304)] * 2, whs)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
del stash[:]
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
gf, ga = stash[0]
gw = getattr(gf, ops.ABSMAX_ATTR)
for rnd in range(3):
    g.replay()
    torch.cuda.synchronize()
    e = step()
    ef, ea = stash[-1]
    ew = getattr(ef, ops.ABSMAX_ATTR)
    torch.cuda.synchronize()
    print("round", rnd, "feat equal", torch.equal(gf, ef), "att equal", torch.equal(ga, ea), "words equal", torch.equal(gw, ew),
          "out equal", [torch.equal(a, b) for a, b in zip(out, e)])
    print("   graph words", gw.view(torch.float32)[0].tolist())
    print("   eager words", ew.view(torch.float32)[0].tolist())
    loc.add_(1.5)
