"""Attender kernels on config-2 maps (16,1,200,304): for rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dance_amd import modules, synthetic  # noqa: E402

dev = torch.device("cuda:0")
att = modules.Attender()
sd = {k: t.clone() for k, t in att.state_dict().items()}
synthetic.fill_state_dict_(sd, 1234)
att.load_state_dict(sd)
att = att.to(dev).eval()
x = torch.rand((16, 1, 200, 304), device=dev)
for _ in range(12):
    y = att(x)
torch.cuda.synchronize()
print(float(y.mean()))
