"""Micro-benchmark of dance_circconv launches (HIP events on torch's current stream)."""
import os, sys, torch
sys.path.insert(0, '.')
from dance_amd import ops
dev = torch.device('cuda')
def run(n, v, cin, cout, taps, dil, reps=20, residual=True, maxv=False):
    x = torch.randn((n, cin, v), device=dev)
    w = ops.pack_conv_weight(torch.randn((cout, cin, taps), device=dev) * 0.03)
    b = torch.randn((cout,), device=dev)
    sc = torch.rand((cout,), device=dev) + 0.5
    sh = torch.randn((cout,), device=dev)
    res = x if (residual and cin == cout) else None
    out = torch.empty((n, cout) if maxv else (n, cout, v), device=dev)
    for _ in range(3):
        ops.circconv(x, w, b, cin=cin, dilation=dil, scale=sc, shift=sh, residual=res, relu=True, maxv=maxv, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.circconv(x, w, b, cin=cin, dilation=dil, scale=sc, shift=sh, residual=res, relu=True, maxv=maxv, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * n * v * cin * cout * taps
    print("n=%5d v=%3d cin=%4d cout=%3d taps=%d d=%d ck=%s : %8.1f us  %6.1f TFLOP/s" % (n, v, cin, cout, taps, dil, os.environ.get("DANCE_DEBUG_CK", "auto"), ms * 1e3, fl / ms / 1e9))
if __name__ == "__main__":
    for n in (256, 512, 1024, 1536, 1600, 2048, 3200):
        run(n, 128, 128, 128, 9, 1)
    run(1600, 128, 128, 128, 1, 1)
    run(1600, 128, 1024, 256, 1, 1, residual=False)
    run(1600, 128, 1024, 256, 1, 1, residual=False, maxv=True)
    run(1600, 128, 256, 64, 1, 1, residual=False)
    run(1600, 256, 128, 128, 9, 4)
    run(1600, 196, 128, 128, 9, 2)
    run(1600, 40, 128, 128, 9, 1)
