"""Determinism probe of dance_fuse16w_raw + dance_tail16 at sizes where workgroups start late
(more workgroups than the CUs hold at once): reruns must be bit-identical.  PYTHONPATH=. """
import sys
import torch
sys.path.insert(0, "tests")
from dance_amd import snake
from _util import dance_state_dict
import test_gpu_tail as T

device = torch.device("cuda:0")
cfg, sd, _ = dance_state_dict(1234, 128)
net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer0."), device)
for v in (128, 196):
    n = 3200
    state, loc, whs, att = T._inputs(net, n, v, device, 1)
    outs = [T._fused(net.struct16, state, loc, whs, att, n, v, 3, False)[1] for _ in range(4)]
    torch.cuda.synchronize()
    print(v, "reruns identical:", all(torch.equal(outs[0], o) for o in outs[1:]))
