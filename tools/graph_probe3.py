"""graph replay vs eager, order of runs and workspace sharing (diagnostic)"""
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


def run(shared):
    dg = snake.ContourDeformer(nets)
    de = dg if shared else snake.ContourDeformer(nets)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            dg.deform(feats, loc, idx, [(200, 304)] * 2, whs)
            de.deform(feats, loc, idx, [(200, 304)] * 2, whs)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = dg.deform(feats, loc, idx, [(200, 304)] * 2, whs)

    def base():
        return [t.clone() for t in snake.ContourDeformer(nets).deform(feats, loc, idx, [(200, 304)] * 2, whs)]

    for op in "R E A E R R E R A R R E".split():
        if op == "A":
            loc.add_(1.5)
            print("  -- loc += 1.5")
            continue
        if op == "R":
            g.replay()
            torch.cuda.synchronize()
            got = [t.clone() for t in out]
        else:
            got = [t.clone() for t in de.deform(feats, loc, idx, [(200, 304)] * 2, whs)]
        b = base()
        print("  ", "replay" if op == "R" else "eager ", [torch.equal(x, y) for x, y in zip(got, b)])


print("shared deformer (one workspace)")
run(True)
print("separate deformers")
run(False)
