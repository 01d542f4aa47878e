"""Micro-benchmark of the HBM-bound stages: layout pass and stand-alone gather (config 2 shapes).
Algorithmic bytes per SURVEY 8d: gather read 4 taps x 257 ch x 4 B per vertex (+ write (C+2) x 4 B)."""
import sys, torch
sys.path.insert(0, '.')
from dance_amd import ops, synthetic
dev = torch.device('cuda')
b, per, v, C, HF, WF = 16, 100, 128, 256, 200, 304
n = b * per
feats = torch.randn((b, C + 1, HF, WF), device=dev)
boxes = synthetic.make_boxes(n, HF * 4, WF * 4, 4321)
loc, _ = ops.uniform_upsample(boxes.to(dev), v, kind=ops.POLY_BOX)
idx = torch.arange(b).repeat_interleave(per).float().to(dev)
def timed(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timed(lambda: ops.nchw_to_nhwc(feats, True))
byt = 2.0 * feats.numel() * 4
print("k_nchw_to_nhwc : %.1f us  %.0f GB/s (read+write %.2f GB)  %.2f of 8 TB/s" % (ms * 1e3, byt / ms / 1e6, byt / 1e9, byt / ms / 1e6 / 8000))
f, att = ops.nchw_to_nhwc(feats, True)
ms = timed(lambda: ops.gather_features(f, att, loc, idx, ops.PAD_REFLECTION, "derange"))
rd, wr = 4.0 * (C + 1) * 4 * v * n, (C + 2.0) * 4 * v * n
print("k_gather       : %.1f us  read %.0f GB/s algorithmic (%.2f of 8 TB/s), read+write %.0f GB/s" % (
    ms * 1e3, rd / ms / 1e6, rd / ms / 1e6 / 8000, (rd + wr) / ms / 1e6))
