"""Time dance_res_stack16 alone (n contours x v vertices, random state and the synthetic weights):
python tools/stack_probe.py [n] [v] [reps] [coop 0|1].  Used for A/B runs of library variants (DANCE_HIP_LIB)
and of the kernel forms (coop = the two-workgroups-per-contour form for few contours)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import dance_state_dict  # noqa: E402
from dance_amd import snake  # noqa: E402
from dance_amd._lib import PolicyStruct, check, current_stream, lib, ptr  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
v = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
coop = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda")
cfg, sd, _ = dance_state_dict(1234, 128)
net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer0."), dev)
state = torch.zeros((n, 8, 2, 16, v, 8), dtype=torch.float16, device=dev)
state[:, 0, 0] = torch.randn((n, 16, v, 8), device=dev).half()
state[:, 0, 1] = (torch.randn((n, 16, v, 8), device=dev) * 1e-4).half()
slot = 256 * v


scratch = torch.zeros((16 + 2 * n,), dtype=torch.int32, device=dev)
policy = PolicyStruct(coop_stack=coop)


def run():
    check(lib().dance_res_stack16(ctypes.byref(net.struct16), ptr(state), state.stride(0), slot, n, v, 3,
                                  ctypes.byref(policy), ptr(scratch), current_stream()), "stack")


for _ in range(5):
    run()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
ev[0].record()
for i in range(reps):
    run()
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
fin = bool(torch.isfinite(state.float()).all())
print("stack n=%d v=%d coop=%d: median %.4f ms, min %.4f ms, finite=%s, status=%d, lib=%s" % (
    n, v, coop, ts[reps // 2], ts[0], fin, int(scratch[0]), os.environ.get("DANCE_HIP_LIB", "default")))
