"""which stage of EdgeSnakeFPNHead.forward differs between two eager runs? (diagnostic)"""
import sys
import torch
sys.path.insert(0, "tests")
from _util import edge_head_cfg
from dance_amd import modules, ops, synthetic

device = torch.device("cuda:0")
h, w = 192, 256
cfg, shapes = edge_head_cfg(num_sampling=128, num_convs=2)
head = modules.EdgeSnakeFPNHead(cfg, shapes)
sd = {k: v.clone() for k, v in head.state_dict().items()}
synthetic.fill_state_dict_(sd, 1234)
head.load_state_dict(sd)
head = head.to(device).eval()
feats = {k: synthetic.make_smooth_features(1, 256, h // s, w // s, 1 + i, attention=False, cell=4).to(device)
         for i, (k, s) in enumerate((("p2", 4), ("p3", 8), ("p4", 16), ("p5", 32)))}


def stages():
    out = {}
    with torch.no_grad():
        x = None
        for f, hd in zip(head.in_features, head.scale_heads):
            mods = list(hd)
            y = feats[f]
            for i, m in enumerate(mods):
                y = m(y)
                out["%s.%d %s" % (f, i, type(m).__name__)] = y.clone()
            x = y if x is None else x + y
            out["sum after " + f] = x.clone()
        x2 = None
        for f, hd in zip(head.in_features, head.scale_heads):
            x2 = head._scale_head_join(x2, *head._scale_head_front(hd, feats[f]))
        out["fused sum"] = x2.clone()
        p0 = head.predictor[0](x2)
        out["predictor.0"] = p0.clone()
        p1 = head.predictor[1](p0)
        out["predictor.1 (1x1)"] = p1.clone()
        e = p1.sigmoid()
        out["att"] = head.attender(1 - e).clone()
    return out


runs = [stages() for _ in range(4)]
for k in runs[0]:
    print("%-28s" % k, [torch.equal(runs[0][k], r[k]) for r in runs[1:]], [float((runs[0][k].float() - r[k].float()).abs().max()) for r in runs[1:]])
