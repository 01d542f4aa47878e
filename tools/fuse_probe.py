"""Time dance_fuse16w_raw alone (n contours x v vertices, random state and the synthetic weights):
python tools/fuse_probe.py [n] [v] [reps].  Used for A/B runs of library variants (DANCE_HIP_LIB)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import dance_state_dict  # noqa: E402
from dance_amd import snake  # noqa: E402
from dance_amd._lib import check, current_stream, lib, ptr  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
v = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device("cuda")
cfg, sd, _ = dance_state_dict(1234, 128)
net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer0."), dev)
s16 = net.struct16
state = torch.zeros((n, 8, 2, 16, v, 8), dtype=torch.float16, device=dev)
state[:, :, 0] = torch.randn((n, 8, 16, v, 8), device=dev).half()
state[:, :, 1] = (torch.randn((n, 8, 16, v, 8), device=dev) * 1e-4).half()
slot = 256 * v
parts = lib().dance_fuse16_parts(n, v)
gpart = torch.empty((n, parts, 256), device=dev)
p1 = torch.zeros((n, 2, 2, 16, v, 8), dtype=torch.float16, device=dev)


def run():
    check(lib().dance_fuse16w_raw(ptr(state), state.stride(0), slot, s16.fuse_w, s16.fuse_bs, s16.fuse_winv,
                                  ptr(p1), p1.stride(0), slot, ptr(gpart), n, v, parts, 3, current_stream()), "fuse16w_raw")


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
print("fuse n=%d v=%d: median %.4f ms, min %.4f ms, lib=%s" % (n, v, ts[reps // 2], ts[0],
                                                                os.path.basename(os.environ.get("DANCE_HIP_LIB", "default"))))
