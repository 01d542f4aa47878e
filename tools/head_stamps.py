"""Phase timeline of k_gather_head16w at BASELINE config 2 (1536 contours x 128 vertices, 16 images of
(200, 304, 256) features) from an experiment build with time stamps (tools/build_variant.py stamps head16.hip
-DDANCE_EXP_STAMPS=1):  DANCE_HIP_LIB=dance_amd/_variants/lib_stamps.so python3 tools/head_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from _util import dance_state_dict  # noqa: E402
from dance_amd import ops, snake, synthetic  # noqa: E402
from dance_amd._lib import check, current_stream, lib, ptr  # noqa: E402

dev = torch.device("cuda:0")
N, V, B, HF, WF = int(os.environ.get("HS_N", "1536")), 128, 16, 200, 304
cfg, sd, _ = dance_state_dict(77, V)
net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer0."), dev)
s16 = net.struct16
feats = torch.cat([synthetic.make_features(1, 257, HF, WF, 1234 + i) for i in range(B)]).to(dev)
nhwc, att = ops.nchw_to_nhwc(feats, 1)
del feats
boxes = synthetic.make_boxes(N, HF * 4, WF * 4, 4321)
loc = synthetic.make_circle_contours(boxes, V).to(dev)
idx = (torch.arange(N) // (N // B)).clamp(max=B - 1).to(torch.int32).to(dev)
y = torch.zeros((N, 8, 2, 16, V, 8), dtype=torch.float16, device=dev)
att_o = torch.zeros((N, V), device=dev)
words = ops.absmax_of(nhwc)


def call():
    check(lib().dance_gather_head16w(ptr(nhwc), ptr(att), ptr(loc), ptr(idx), s16.head_w, s16.head_wd, s16.head_bs,
                                     s16.head_winv, s16.head_wd_winv, s16.head_scale, s16.head_shift, ptr(y), y.stride(0),
                                     ptr(att_o), N, V, B, HF, WF, 256, ops.PAD_REFLECTION, ops.DELOC["derange"], 3,
                                     ptr(words), None, current_stream()), "head")


for _ in range(5):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    call()
e1.record()
torch.cuda.synchronize()
print("head kernel: %.4f ms per launch" % (e0.elapsed_time(e1) / 20))
if not hasattr(lib(), "dance_exp_head_stamps"):
    sys.exit(0)
buf = np.zeros((2048, 20), dtype=np.uint64)
lib().dance_exp_head_stamps.argtypes = [ctypes.c_void_p]
assert lib().dance_exp_head_stamps(buf.ctypes.data) == 0
st = buf[:N].astype(np.int64)
t0 = st[:, 0].min()
us = (st[:, :17] - t0) / 100.0                     # 100 MHz -> us
hw, xcc = st[:, 17], st[:, 18]
cu = ((xcc & 15) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
tg = (hw >> 16) & 15
print("kernel span %.1f us; distinct CUs %d; TG_ID values %s" % (us[:, 16].max(), len(set(cu.tolist())), sorted(set(tg.tolist()))))
first = us[:, 0] < 5.0
print("workgroups starting in the first 5 us: %d" % first.sum())


def stats(name, x):
    print("%-38s mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f us" % (name, x.mean(), *np.percentile(x, [10, 50, 90])))


for sel, tag in ((first, "first round"), (~first, "later rounds")):
    if sel.sum() == 0:
        continue
    u = us[sel]
    print("--", tag, int(sel.sum()), "workgroups")
    stats("setup (bounds, taps, de_location)", u[:, 2] - u[:, 0])
    for c in range(4):
        stats("chunk %d gather (wave 0)" % c, u[:, 3 + 3 * c] - u[:, 2 + 3 * c])
        stats("chunk %d wait for the other waves" % c, u[:, 4 + 3 * c] - u[:, 3 + 3 * c])
        nxt = u[:, 2 + 3 * (c + 1)] if c < 3 else u[:, 16]
        stats("chunk %d MFMA (+ barrier / epilogue)" % c, nxt - u[:, 4 + 3 * c])
    stats("whole workgroup", u[:, 16] - u[:, 0])
# how synchronised are the phases chip-wide: at sampled times, the share of live workgroups inside a gather phase
ts = np.arange(0.0, us[:, 16].max(), 2.0)
share = []
for t in ts:
    live = (us[:, 0] <= t) & (us[:, 16] > t)
    ing = np.zeros(N, dtype=bool)
    for c in range(4):
        ing |= (us[:, 2 + 3 * c] <= t) & (us[:, 4 + 3 * c] > t)
    share.append((int(live.sum()), int((ing & live).sum())))
print("t(us): live workgroups / of them gathering")
print(" ".join("%d:%d/%d" % (t, a, b) for t, (a, b) in zip(ts, share)))
# per-CU: order of one CU's workgroups
one = cu == cu[0]
print("workgroups of CU %#x: " % cu[0], [(int(i), int(tg[i]), round(float(us[i, 0]), 1), round(float(us[i, 16]), 1)) for i in np.nonzero(one)[0]])
