"""Experiment: the deformation of two halves of the contours on two HIP streams (kernel tails and
launch gaps of one stream overlap the other's kernels) vs one stream."""
import sys, time, torch
sys.path.insert(0, '.')
from dance_amd import config, modules, ops, snake, synthetic
dev = torch.device('cuda')
b, per, v, CH, HF, WF = 16, 100, 128, 256, 200, 304
n = b * per
cfg = config.dance_r50(num_sampling=v, num_convs=0)
sd = {k: t.clone() for k, t in modules.RefineNet(cfg).state_dict().items()}
synthetic.fill_state_dict_(sd, 1234)
nets = [snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer%d." % i), dev) for i in range(3)]
ds = [snake.ContourDeformer(nets) for _ in range(4)]
feats = torch.empty((b, CH + 1, HF, WF), device=dev)
for i in range(b):
    feats[i] = synthetic.make_features(1, CH, HF, WF, 1234 + i)[0].to(dev)
boxes = synthetic.make_boxes(n, HF * 4, WF * 4, 4321)
whs = torch.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1).to(dev)
init, _ = ops.uniform_upsample(boxes.to(dev), v, kind=ops.POLY_BOX)
idx = torch.arange(b).repeat_interleave(per).float().to(dev)
sizes = [(HF * 4, WF * 4)] * b
f = ds[0].prepare_features(feats)
def one():
    return ds[0].deform(f, init, idx, sizes, whs, layout="nhwc")[-1]
def multi(k):
    streams = [torch.cuda.Stream() for _ in range(k)]
    cuts = [n * i // k for i in range(k + 1)]
    def run():
        cur = torch.cuda.current_stream()
        outs = []
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                sl = slice(cuts[i], cuts[i + 1])
                outs.append(ds[i].deform(f, init[sl], idx[sl], sizes, whs[sl], layout="nhwc")[-1])
        for s in streams:
            cur.wait_stream(s)
        return torch.cat(outs)
    return run
def timed(fn, reps=10):
    for _ in range(3):
        r = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r
t1, r1 = timed(one)
print("1 stream : %.3f ms (excl. layout)" % t1)
for k in (2, 3, 4):
    tk, rk = timed(multi(k))
    print("%d streams: %.3f ms  same=%s" % (k, tk, bool(torch.equal(r1, rk))))
