"""Dsnake module forward at 32 x 100 with one and with two HIP streams: which stages / contours differ."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from _util import FakeBoxes, FakeInstances
from dance_amd import config, modules, synthetic
dev = torch.device('cuda')
b, per, HF, WF = 32, 100, 200, 304
cfg = config.dsnake_r50(num_convs=0)
head = modules.DsnakeHead(cfg)
sd = {k: v.clone() for k, v in head.state_dict().items()}
synthetic.fill_state_dict_(sd, 1234)
head.load_state_dict(sd)
head = head.to(dev).eval()
feats = torch.randn((b, 256, HF, WF), device=dev)
boxes = synthetic.make_boxes(b * per, HF * 4, WF * 4, 78).view(b, per, 4).to(dev)


def run(streams):
    os.environ["DANCE_STREAMS"] = str(streams)
    insts = [FakeInstances((HF * 4, WF * 4), pred_boxes=FakeBoxes(boxes[i])) for i in range(b)]
    with torch.no_grad():
        head(feats, insts, None)
    torch.cuda.synchronize()
    st = head.last_stages
    return {"exts_raw": st["exts_raw"].clone(), "exts": st["exts"].clone(), "init": st["init"].clone(), "octagons": st["octagons"].clone(),
            **{"pred%d" % i: p.clone() for i, p in enumerate(st["preds"])}}


ref = run(1)
print("library:", os.environ.get("DANCE_HIP_LIB", "in-tree"), flush=True)
for trial in range(3):
    for k in (1, 2):
        got = run(k)
        for name in ref:
            d = (got[name] - ref[name]).abs().reshape(len(ref[name]), -1).max(dim=1)[0]
            bad = (d != 0).nonzero().flatten()
            if len(bad):
                print("trial %d streams %d %s: %d contours differ, max %.3e, first %s" % (
                    trial, k, name, len(bad), float(d.max()), bad[:8].tolist()))
                if got[name].dim() == 3 and got[name].shape[1] % 64 == 0:
                    # which vertices: (pass = vertex // 64, lane quarter = vertex % 64 // 16) -> count
                    vd = ((got[name] - ref[name]).abs().amax(dim=2) != 0).nonzero()[:, 1]
                    hist = {}
                    for v in vd.tolist():
                        key = (v // 64, v % 64 // 16)
                        hist[key] = hist.get(key, 0) + 1
                    print("    differing vertices by (pass, lane quarter):", sorted(hist.items()))
                    if name == "octagons":
                        ne = (got[name] != ref[name])
                        print("    differing x / y coordinates: %d / %d" % (int(ne[..., 0].sum()), int(ne[..., 1].sum())))
                        i, v = [int(t) for t in ne[..., 0].nonzero()[0]] if int(ne[..., 0].sum()) else (int(bad[0]), 0)
                        print("    e.g. contour %d vertex %d: got %s want %s, neighbours want %s %s" % (
                            i, v, got[name][i, v].tolist(), ref[name][i, v].tolist(), ref[name][i, v - 1].tolist(),
                            ref[name][i, (v + 1) % ref[name].shape[1]].tolist()))
                if name == "octagons" and trial == 0:
                    i = int(bad[0])
                    print(" exts ref", ref["exts"][i].flatten().tolist())
                    print(" exts got", got["exts"][i].flatten().tolist())
                    print(" oct ref", ref["octagons"][i, :4].flatten().tolist())
                    print(" oct got", got["octagons"][i, :4].flatten().tolist())
                    dd = (got["octagons"][i] - ref["octagons"][i]).abs().max(dim=1)[0]
                    print(" vertices differing:", (dd != 0).nonzero().flatten().tolist()[:40])
print("done")
