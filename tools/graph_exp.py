"""Experiment: one HIP graph per deformation step vs eager launches (same workload as bench.py)."""
import sys, time, torch
sys.path.insert(0, '.')
from dance_amd import config, modules, ops, snake, synthetic
dev = torch.device('cuda')
import os
b, per, v, CH, HF, WF = int(os.environ.get("GB", "16")), 100, 128, 256, 200, 304
n = b * per
cfg = config.dance_r50(num_sampling=v, num_convs=0)
sd = {k: t.clone() for k, t in modules.RefineNet(cfg).state_dict().items()}
synthetic.fill_state_dict_(sd, 1234)
nets = [snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer%d." % i), dev) for i in range(3)]
d = snake.ContourDeformer(nets)
d.MAX_CHUNK = max(d.MAX_CHUNK, n)
feats = torch.randn((b, CH + 1, HF, WF), device=dev)
boxes = synthetic.make_boxes(n, HF * 4, WF * 4, 4321)
whs = torch.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1).to(dev)
init, _ = ops.uniform_upsample(boxes.to(dev), v, kind=ops.POLY_BOX)
idx = torch.arange(b).repeat_interleave(per).float().to(dev)
sizes = [(HF * 4, WF * 4)] * b
def step():
    return d.deform(feats, init, idx, sizes, whs)[-1]
for _ in range(3):
    ref = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print("eager  %.3f ms/step" % ((time.perf_counter() - t0) * 100))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = step()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print("graph  %.3f ms/step" % ((time.perf_counter() - t0) * 100))
print("same result:", bool(torch.equal(out, ref)))
