"""which of eager / graph replay is off after the inputs change in place?  (diagnostic)"""
import sys
import torch
sys.path.insert(0, "tests")
from dance_amd import snake, synthetic
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
d32 = snake.ContourDeformer(nets, precision="fp32")


def step(dd=d):
    return dd.deform(feats, loc, idx, [(200, 304)] * 2, whs)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        eager = [t.clone() for t in step()]
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
g.replay()
torch.cuda.synchronize()
print("replay == eager (before change):", [torch.equal(a, b) for a, b in zip(out, eager)])
loc.add_(1.5)
torch.cuda.synchronize()
for rnd in range(2):
    want = [t.clone() for t in step()]
    ref = [t.clone() for t in step(d32)]
    g.replay()
    torch.cuda.synchronize()
    for it in range(3):
        de = (want[it] - ref[it]).abs().flatten(1).max(1)[0]
        dg = (out[it] - ref[it]).abs().flatten(1).max(1)[0]
        print("round", rnd, "it", it, "eager==replay", torch.equal(want[it], out[it]),
              "| eager vs fp32 max %.4f" % float(de.max()), "replay vs fp32 max %.4f" % float(dg.max()),
              "| contours off (eager, replay):", int((de > 0.1).sum()), int((dg > 0.1).sum()),
              "nan", bool(torch.isnan(want[it]).any()), bool(torch.isnan(out[it]).any()))


# ---- who deviates?
print("---- consistency")
e1 = [t.clone() for t in step()]
e2 = [t.clone() for t in step()]
g.replay(); torch.cuda.synchronize()
r1 = [t.clone() for t in out]
g.replay(); torch.cuda.synchronize()
r2 = [t.clone() for t in out]
dn = snake.ContourDeformer(nets)
f1 = [t.clone() for t in dn.deform(feats, loc, idx, [(200, 304)] * 2, whs)]
print("eager twice equal", [torch.equal(a, b) for a, b in zip(e1, e2)])
print("replay twice equal", [torch.equal(a, b) for a, b in zip(r1, r2)])
print("fresh deformer == eager", [torch.equal(a, b) for a, b in zip(f1, e1)])
print("fresh deformer == replay", [torch.equal(a, b) for a, b in zip(f1, r1)])
nd = (e1[0] != r1[0]).flatten(1).any(1)
print("it0 contours differing:", nd.nonzero().flatten().tolist())
print("it0 max abs diff eager-replay %.6f" % float((e1[0] - r1[0]).abs().max()))
# the absmax words eager sees
from dance_amd import ops
fe, ae = d.prepare_features(feats)
print("eager absmax words", getattr(fe, ops.ABSMAX_ATTR).view(torch.float32).tolist())
