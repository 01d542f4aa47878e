"""Per-slot largest |activation| of the snake nets on the benchmark's synthetic data (torch fp32 on
the GPU, inputs from the exact-fp32 chain) against the pack-time estimates of
SnakeNetWeights._pack16 (act_exps): how far off are the estimates?  (diagnostic)"""
import math
import os
import sys
import torch
sys.path.insert(0, "tests")
os.environ["DANCE_PRECISION"] = "fp32"
from dance_amd import modules, ops, snake, synthetic
from _util import FakeBoxes, FakeInstances, dance_state_dict, dsnake_state_dict
from _fast_ref import fuse_ref, head_ref, res_block_ref, w32

dev = torch.device("cuda:0")
HF, WF = 200, 304
F = torch.nn.functional


def report(tag, net, w, nhwc, att, loc, idx, pad, deloc):
    x, _ = head_ref(w, nhwc, att, loc, idx, pad, deloc)
    mx = [float(x.abs().max())]
    states = [x]
    for i in range(7):
        x = res_block_ref(w, i, x, net.dilations[i])
        states.append(x)
        mx.append(float(x.abs().max()))
    xd = torch.cat(states, 1)
    g, p = fuse_ref(w, xd)
    gm = g.max(dim=2, keepdim=True)[0]
    h = F.conv1d(gm, w["prediction.0.weight"][:, :256], w["prediction.0.bias"])
    p1 = float((p + h).abs().max())
    b = net.act_exps["bounds"]
    print(tag, "finite", bool(torch.isfinite(xd).all()))
    print("   slot max      ", " ".join("%9.3g" % m for m in mx), " p1 %9.3g" % p1)
    print("   estimate      ", " ".join("%9.3g" % m for m in b))
    print("   log2(max/est) ", " ".join("%9.2f" % math.log2(m / e) for m, e in zip(mx, b)))
    print("   stored max at 2^a (limit 65504):", " ".join("%9.3g" % (m * 2.0 ** a) for m, a in zip(mx, net.act_exps["slots"])),
          " p1 %9.3g" % (p1 * 2.0 ** net.act_exps["pred0"]))


# DANCE, config 2-like
cfg, sd, _ = dance_state_dict(1234, 128)
subs = [snake.sub_state_dict(sd, "deformer%d." % i) for i in range(3)]
nets = [snake.SnakeNetWeights(s, dev) for s in subs]
for kind in ("randn", "smooth"):
    b, per = 2, 100
    mk = synthetic.make_features if kind == "randn" else synthetic.make_smooth_features
    feats = mk(b, 256, HF, WF, 1234).to(dev)
    boxes = synthetic.make_boxes(b * per, HF * 4, WF * 4, 77)
    whs = torch.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1).to(dev)
    init, _ = ops.uniform_upsample(boxes.to(dev), 128, kind=ops.POLY_BOX)
    idx = torch.arange(b).repeat_interleave(per).to(dev)
    d = snake.ContourDeformer(nets, precision="fp32")
    preds = d.deform(feats, init, idx.float(), [(HF * 4, WF * 4)] * b, whs)
    nhwc, att = ops.nchw_to_nhwc(feats, 1)
    prev = init
    for it in range(3):
        report("DANCE %s it%d" % (kind, it), nets[it], w32(subs[it], dev), nhwc, att, prev, idx.to(torch.int32),
               ops.PAD_REFLECTION, "derange")
        prev = preds[it]

# Dsnake, config 4-like
cfg, sd, _ = dsnake_state_dict(1234)
head = modules.DsnakeHead(cfg)
head.load_state_dict(sd)
head = head.to(dev).eval()
head.one_launch_init = False
b, per = 2, 100
feats = synthetic.make_smooth_features(b, 256, HF, WF, 900, attention=False).to(dev)
boxes = synthetic.make_boxes(b * per, HF * 4, WF * 4, 78).view(b, per, 4)
insts = [FakeInstances((HF * 4, WF * 4), pred_boxes=FakeBoxes(boxes[i].to(dev))) for i in range(b)]
with torch.no_grad():
    res, _ = head(feats, insts, None)
dd = head._get_deformer()
print("dsnake deformer precision", dd.precision)
st = head.last_stages
nhwc, _ = ops.nchw_to_nhwc(feats, 0)
idx = torch.arange(b).repeat_interleave(per).to(dev).to(torch.int32)
names = ["init_snake"] + ["deformer%d" % i for i in range(3)]
prev = st["octagons"]
for it in range(3):
    sub = {k[len(names[1 + it]) + 1:]: v for k, v in sd.items() if k.startswith(names[1 + it] + ".")}
    print("   input extent max", float((prev.max(1)[0] - prev.min(1)[0]).max()))
    report("Dsnake it%d" % it, dd.nets[1 + it], w32(sub, dev), nhwc, None, prev, idx, ops.PAD_ZEROS, "demin")
    prev = st["preds"][it]
