"""EdgeSnakeFPNHead drop-in on one 800x1216 image (the reference's inference mode is one image per
GPU, dance.py:52): wall time per forward with the conv3x3 + GroupNorm + ReLU layers on the HIP
feature-prep kernels vs stock torch.nn (DANCE_FEATPREP=torch)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from _util import FakeBoxes, FakeInstances, edge_head_cfg  # noqa: E402
from dance_amd import modules, synthetic, timer  # noqa: E402

dev = torch.device("cuda:0")
cfg, shapes = edge_head_cfg(num_sampling=128, num_convs=2)
head = modules.EdgeSnakeFPNHead(cfg, shapes)
sd = {k: v.clone() for k, v in head.state_dict().items()}
synthetic.fill_state_dict_(sd, 1234)
head.load_state_dict(sd)
head = head.to(dev).eval()
H, W = 800, 1216
feats = {k: torch.randn((1, 256, H // s, W // s), device=dev) for k, s in (("p2", 4), ("p3", 8), ("p4", 16), ("p5", 32))}
boxes = synthetic.make_boxes(100, H, W, 7).to(dev)


def run():
    inst = FakeInstances((H, W), pred_boxes=FakeBoxes(boxes.clone()))
    with torch.no_grad():
        return head(feats, [inst], None)


for mode in ("hip", "torch"):
    os.environ["DANCE_FEATPREP"] = mode
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = run()
    torch.cuda.synchronize()
    print("EdgeSnakeFPNHead, 1 image, 100 instances, DANCE_FEATPREP=%s: %.2f ms per forward" % (
        mode, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
