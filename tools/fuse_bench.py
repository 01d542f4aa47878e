"""dance_fuse16 alone (fusion + max + row constant + prediction[0]) on 1600 contours x 128 vertices:
launch time via HIP events.  DANCE_HIP_LIB selects another build (A/B of kernel variants)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from _util import dance_state_dict  # noqa: E402
from dance_amd import ops, snake  # noqa: E402
from dance_amd._lib import check, current_stream, lib, ptr  # noqa: E402

dev = torch.device("cuda:0")
n, v = int(os.environ.get("FB_N", "1600")), 128
cfg, sd, _ = dance_state_dict(1234, v)
net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer0."), dev)
s16 = net.struct16
x = ops.to_p16(torch.randn((n, 1024, v), device=dev))
y = torch.zeros((n, 2, 2, 16, v, 8), dtype=torch.float16, device=dev)
slot = 256 * v


def f():
    check(lib().dance_fuse16(ptr(x), x.stride(0), slot, s16.fuse_w, s16.fuse_bs, s16.fuse_winv, s16.pred0_wg,
                             s16.pred0_b, ptr(y), y.stride(0), slot, n, v, 3, current_stream()), "fuse16")


for _ in range(3):
    f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f()
e1.record()
torch.cuda.synchronize()
print("%s: dance_fuse16 n=%d  %.1f us per launch" % (os.path.basename(os.environ.get("DANCE_HIP_LIB", "default")),
                                                  n, e0.elapsed_time(e1) / 20 * 1e3))
