"""Feature prep (RefineNet.bottom_out: 2 x conv3x3 256->256 + GN(32) + ReLU) on config-2 maps:
HIP fast-mode path vs the stock torch.nn modules (MIOpen fp32, channels_last)."""
import sys, torch
sys.path.insert(0, '.')
from dance_amd import config, modules, ops, synthetic
dev = torch.device('cuda')
b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = config.dance_r50(num_sampling=128, num_convs=2)
net = modules.RefineNet(cfg)
sd = {k: v.clone() for k, v in net.state_dict().items()}
synthetic.fill_state_dict_(sd, 1234)
net.load_state_dict(sd)
net = net.to(dev).eval()
feats = torch.randn((b, 257, 200, 304), device=dev)
def timed(f, reps=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
layers = net._bottom_hip_layers(dev)
flops = 2 * 2.0 * 256 * 256 * 9 * 200 * 304 * b
with torch.no_grad():
    ms = timed(lambda: ops.bottom_out16(feats, 1, layers))
    print("HIP  bottom_out16 : %8.3f ms  %.0f TFLOP/s algorithmic (x3 = %.0f f16)" % (ms, flops / ms / 1e9, 3 * flops / ms / 1e9))
    def stock():
        modules._ConvNormAct.use_hip = False                 # the plain torch.nn ops
        try:
            x = feats[:, 1:].contiguous(memory_format=torch.channels_last)
            for conv in net.bottom_out:
                x = conv(x)
            return x.permute(0, 2, 3, 1).contiguous()
        finally:
            modules._ConvNormAct.use_hip = True
    ms2 = timed(stock, reps=3)
    print("torch.nn (MIOpen) : %8.3f ms  %.0f TFLOP/s   speed-up %.2fx" % (ms2, flops / ms2 / 1e9, ms2 / ms))
    a, c = ops.bottom_out16(feats, 1, layers), stock()
    print("max |diff| / max |ref| = %.3g" % float((a - c).abs().max() / c.abs().max()))
