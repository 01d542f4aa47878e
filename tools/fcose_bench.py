"""Times the FCOSE decode (dance_amd.fcose_outputs.decode_candidates) on BASELINE-size head
outputs: 16 images of 800 x 1216, five levels, 80 classes.  Used under rocprofv3 for
profiles/*fcose*:  python3 tools/fcose_bench.py [mean_logit]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from dance_amd import fcose_outputs


def main():
    mean = float(sys.argv[1]) if len(sys.argv) > 1 else -6.5
    dev = torch.device("cuda:0")
    shapes, strides, n, c, th = [(100, 152), (50, 76), (25, 38), (13, 19), (7, 10)], [8, 16, 32, 64, 128], 16, 80, 0.05
    g = torch.Generator(device=dev).manual_seed(77)
    logits = [torch.randn(n, c, h, w, generator=g, device=dev) * 1.5 + mean for h, w in shapes]
    ctr = [torch.randn(n, 1, h, w, generator=g, device=dev) for h, w in shapes]
    reg = [torch.rand(n, 4, h, w, generator=g, device=dev) * 6 + 0.25 for h, w in shapes]
    ext = [torch.randn(n, 4, h, w, generator=g, device=dev) * 0.3 for h, w in shapes]
    loc = fcose_outputs.compute_locations(logits, strides)
    run = lambda: fcose_outputs.decode_candidates(loc, logits, reg, ext, ctr, strides, th, True)
    for _ in range(3):
        fields, _ = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 20
    nbytes = sum(n * (c + 1) * h * w * 4 for h, w in shapes)
    print("candidates %d  wall %.3f ms  logits+ctr %.1f MB per pass" % (fields["scores"].numel(), ms, nbytes / 1e6))

    class Images:
        image_sizes = [(800, 1216)] * n

        def __len__(self):
            return n

    out = fcose_outputs.FCOSEOutputs(Images(), loc, logits, reg, ext, ctr, 0.25, 2.0, None, None, True, None, strides,
                                     1.5, c, th, 1000, 0.6, 100, True)
    res = out.predict_proposals()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        res = out.predict_proposals()
    torch.cuda.synchronize()
    print("predict_proposals (decode + top-k + ml_nms + cut, %d images): %.3f ms, %s detections"
          % (n, (time.perf_counter() - t0) * 1e3 / 5, [len(r) for r in res][:4]))


if __name__ == "__main__":
    main()
