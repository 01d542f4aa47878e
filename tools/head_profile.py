"""torch.profiler table of the EdgeSnakeFPNHead drop-in's forward on one 800x1216 image with 100
instances: which kernels (this library's and stock torch's) the one-image inference spends its
GPU time in.  python3 tools/head_profile.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _util import FakeBoxes, FakeInstances, edge_head_cfg
from dance_amd import modules, synthetic
dev = torch.device("cuda:0")
cfg, shapes = edge_head_cfg(num_sampling=128, num_convs=2)
head = modules.EdgeSnakeFPNHead(cfg, shapes)
sd = {k: v.clone() for k, v in head.state_dict().items()}
synthetic.fill_state_dict_(sd, 1234); head.load_state_dict(sd); head = head.to(dev).eval()
H, W = 800, 1216
feats = {k: torch.randn((1, 256, H // s, W // s), device=dev) for k, s in (("p2", 4), ("p3", 8), ("p4", 16), ("p5", 32))}
boxes = synthetic.make_boxes(100, H, W, 7).to(dev)
def run():
    inst = FakeInstances((H, W), pred_boxes=FakeBoxes(boxes.clone()))
    with torch.no_grad():
        return head(feats, [inst], None)
for _ in range(5): run()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): run()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
