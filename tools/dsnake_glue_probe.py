"""Which torch ops (the glue around the HIP kernels) a DsnakeHead forward at 32 x 100 issues:
torch.profiler table by launch count, NUM_CONVS=0."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from _util import FakeBoxes, FakeInstances
from dance_amd import config, modules, synthetic
from torch.profiler import profile, ProfilerActivity
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
def run():
    insts = [FakeInstances((HF * 4, WF * 4), pred_boxes=FakeBoxes(boxes[i])) for i in range(b)]
    with torch.no_grad():
        return head(feats, insts, None)
for _ in range(3):
    run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=40, max_name_column_width=60))
