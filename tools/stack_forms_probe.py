"""8-wave vs 4-wave residual stack: where do the bits differ?  (diagnostic)"""
import ctypes
import sys
import torch
sys.path.insert(0, "tests")
from dance_amd import ops, snake
from dance_amd._lib import PolicyStruct, check, current_stream, lib, ptr
from _util import dance_state_dict
from _fast_ref import slot_scales

dev = torch.device("cuda:0")
cfg, sd, _ = dance_state_dict(99, 128)
net = snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer1."), dev)
s16 = net.struct16
print("act exps", net.act_exps, "rmul", list(s16.res_rmul), "winv", list(s16.res_winv))
for v in (128, 100):
    slot = 256 * v
    sc0 = float(slot_scales(net, dev)[0])
    x = torch.randn((37, 128, v), generator=torch.Generator().manual_seed(v + 1)) * 2
    c = torch.zeros((37, 8, 2, 16, v, 8), dtype=torch.float16, device=dev)
    c[:, 0] = ops.to_p16((x * sc0).to(dev))[:, 0]
    outs = []
    for pol in (None, PolicyStruct(few_contour_forms=0), None, PolicyStruct(few_contour_forms=0)):
        e = c.clone()
        check(lib().dance_res_stack16(ctypes.byref(s16), ptr(e), e.stride(0), slot, 37, v, 3,
                                      None if pol is None else ctypes.byref(pol), current_stream()), "stack")
        torch.cuda.synchronize()
        outs.append(e)
    print("v", v, "8w==8w", torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16)),
          "4w==4w", torch.equal(outs[1].view(torch.int16), outs[3].view(torch.int16)))
    a, b = outs[0], outs[1]
    d = (a.view(torch.int16) != b.view(torch.int16))
    print("  differing halfs", int(d.sum()), "of", d.numel())
    for s in range(8):
        ds = d[:, s]
        print("  slot", s, "diff", int(ds.sum()), "hi plane", int(ds[:, 0].sum()), "lo plane", int(ds[:, 1].sum()))
    fa, fb = ops.from_p16(a), ops.from_p16(b)
    print("  max abs diff of values", float((fa - fb).abs().max()), "max", float(fb.abs().max()))
    idx = d.nonzero()[:10]
    print("  first diffs (n, slot, plane, cb, v, e):", idx.tolist())
    for i in idx[:5].tolist():
        print("   ", a[tuple(i)].item(), b[tuple(i)].item())
