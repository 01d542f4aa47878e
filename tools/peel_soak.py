"""Soak of deform()'s peeled remainder: random batch sizes and ring lengths, peeled against DANCE_PEEL=0,
bit for bit, many rounds back to back without synchronising in between (a missing stream dependency
would show as sporadic differences)."""
import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import dance_state_dict
from dance_amd import ops, snake, synthetic
dev = torch.device("cuda")
rnd = random.Random(7)
bad = 0
for v in (128, 196, 96):
    cfg, sd, _ = dance_state_dict(900 + v, v)
    nets = [snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer%d." % i), dev) for i in range(3)]
    d = snake.ContourDeformer(nets)
    b, hf, wf = 4, 100, 152
    feats = torch.randn((b, 257, hf, wf), device=dev)
    f = d.prepare_features(feats)
    for trial in range(12):
        n = rnd.choice([513, 600, 700, 777, 1000, 1100, 1300, 1599, 1600, 1601, 2000, 2047])
        boxes = synthetic.make_boxes(n, hf * 4, wf * 4, trial)
        whs = torch.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1).to(dev)
        init, _ = ops.uniform_upsample(boxes.to(dev), v, kind=ops.POLY_BOX)
        idx = torch.randint(0, b, (n,), device=dev).sort()[0].to(torch.int32)
        os.environ.pop("DANCE_PEEL", None)
        side = d.peel_count(n, dev)
        runs = [d.deform(f, init, idx, [(hf * 4, wf * 4)] * b, whs, layout="nhwc") for _ in range(4)]
        os.environ["DANCE_PEEL"] = "0"
        one = d.deform(f, init, idx, [(hf * 4, wf * 4)] * b, whs, layout="nhwc")
        torch.cuda.synchronize()
        ok = all(torch.equal(r[i], one[i]) for r in runs for i in range(3))
        bad += not ok
        print("v=%d n=%d side=%d %s" % (v, n, side, "ok" if ok else "DIFFERENT"))
os.environ.pop("DANCE_PEEL", None)
print("soak:", "all identical" if bad == 0 else "%d DIFFERENT" % bad)
sys.exit(1 if bad else 0)
