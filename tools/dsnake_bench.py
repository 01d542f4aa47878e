"""BASELINE config 4 (Dsnake R50, 32 images x 100 instances) through the drop-in module:
init (V=40) -> extreme points -> octagon -> 3 evolves; wall time per forward."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from _util import FakeBoxes, FakeInstances
from dance_amd import config, modules, synthetic
dev = torch.device('cuda')
b, per, HF, WF = 32, 100, 200, 304
only = os.environ.get("DSNAKE_BENCH_ONLY")
for nc in ((0, 2) if only is None else (int(only),)):
    cfg = config.dsnake_r50(num_convs=nc)
    head = modules.DsnakeHead(cfg)
    sd = {k: v.clone() for k, v in head.state_dict().items()}
    synthetic.fill_state_dict_(sd, 1234)
    head.load_state_dict(sd)
    head = head.to(dev).eval()
    feats = torch.randn((b, 256, HF, WF), device=dev)
    boxes = synthetic.make_boxes(b * per, HF * 4, WF * 4, 78).view(b, per, 4).to(dev)
    def run():
        insts = [FakeInstances((HF * 4, WF * 4), pred_boxes=FakeBoxes(boxes[i])) for i in range(b)]
        with torch.no_grad():
            return head(feats, insts, None)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = int(os.environ.get("DSNAKE_BENCH_REPS", "5"))
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("DsnakeHead NUM_CONVS=%d: %.2f ms per forward, %.0f contours/s" % (nc, ms, b * per / ms * 1e3))
