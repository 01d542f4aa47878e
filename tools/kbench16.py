"""Micro-benchmark of dance_conv16 launches (HIP events on torch's current stream)."""
import os, sys, torch
sys.path.insert(0, '.')
from dance_amd import ops
from dance_amd._lib import lib, check, ptr, current_stream
dev = torch.device('cuda')
def run(n, v, cin, cout, taps, dil, reps=3 if os.environ.get('DANCE_DEBUG_STAMPS') else 20, residual=True, tag=""):
    x = ops.to_p16(torch.randn((n, cin, v), device=dev) * (0.0 if os.environ.get('KB_ZERO') else 1.0))
    w, s = ops.pack_conv_weight16(torch.randn((cout, cin, taps), device=dev) * 0.03)
    b = (torch.randn((cout,), device=dev) * s).contiguous()
    sc = torch.rand((cout,), device=dev) + 0.5
    sh = torch.randn((cout,), device=dev)
    res = residual and cin == cout
    slot = 256 * v
    y = torch.zeros((n, max(1, cout // 128), 2, 16, v, 8), dtype=torch.float16, device=dev)
    dbgbuf = torch.zeros(64, dtype=torch.int64, device=dev)
    def f():
        check(lib().dance_conv16(ptr(x), x.stride(0), slot, ptr(w), ptr(b), 1.0 / s, ptr(sc), ptr(sh), None,
                                 int(res), 1, 0, ptr(y), y.stride(0), slot, None, 0, n, v, cin, cout, taps, dil,
                                 int(os.environ.get('KB_TERMS', '3')), current_stream()), "conv16")
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    e0.record()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    host_us = (time.perf_counter() - t0) / reps * 1e6
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    if int(os.environ.get('DANCE_DEBUG_FLAGS', '0')) & 64:
        tt = dbgbuf.cpu().tolist(); tt = [x for x in tt if x]
        print('stamps (cycles):', [b - a for a, b in zip(tt, tt[1:])])
    fl = 2.0 * n * v * cin * cout * taps
    print("%-10s n=%5d v=%3d cin=%4d cout=%3d taps=%d d=%d flags=%s slots=%s : %8.1f us  %6.1f TFLOP/s (x3 = %.0f f16) host %.1f us/call" % (
        tag, n, v, cin, cout, taps, dil, os.environ.get("DANCE_DEBUG_FLAGS", "0"), os.environ.get("DANCE_DEBUG_SLOTS", "auto"),
        ms * 1e3, fl / ms / 1e9, 3 * fl / ms / 1e9, host_us), flush=True)
if __name__ == "__main__":
    ns = [int(a) for a in sys.argv[1:]] or [1536, 1600]
    for n in ns:
        run(n, 128, 128, 128, 9, 1)
    if os.environ.get("KB_ALL"):
        run(1600, 128, 256, 64, 1, 1, residual=False)
        run(1600, 128, 1024, 256, 1, 1, residual=False)
        run(1600, 196, 128, 128, 9, 2)
        run(1600, 256, 128, 128, 9, 4)
        run(1600, 40, 128, 128, 9, 1)
