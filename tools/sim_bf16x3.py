"""CPU simulation: error of a bf16 3-term split (hi*hi + hi*lo + lo*hi, fp32-ish accumulate)
for every conv of the snake, vs an fp64 evaluation, next to plain fp32's own error."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch.nn.functional as F
from test_oracle_golden import dance_case_inputs
from oracle import snake_ref

def split(t, dt=torch.bfloat16, scale=1.0):
    t = t.float() * scale
    h = t.to(dt).float()
    l = (t - h).to(dt).float()
    return h.double() / scale, l.double() / scale

MODE = {"m": "fp64"}
orig_conv1d = F.conv1d
def conv1d(x, w, b=None, **kw):
    if MODE["m"] == "fp64" or MODE["m"] == "fp32":
        return orig_conv1d(x, w, b, **kw)
    if MODE["m"].startswith("fp16"):
        ws = 2.0 ** torch.floor(torch.log2(1024.0 / w.abs().max())).item()
        xh, xl = split(x, torch.float16); wh, wl = split(w, torch.float16, ws)
        if MODE["m"] == "fp16x3s":      # activations only exist as hi+lo (22-bit storage)
            pass
    else:
        xh, xl = split(x); wh, wl = split(w)
    y = orig_conv1d(xh, wh, None, **kw) + orig_conv1d(xh, wl, None, **kw) + orig_conv1d(xl, wh, None, **kw)
    if MODE["m"] == "bf16x4":
        y = y + orig_conv1d(xl, wl, None, **kw)
    if b is not None:
        y = y + b.double()[None, :, None]
    y = y.float()
    if MODE["m"] == "fp16x3s":             # stored as an fp16 hi/lo pair
        h = y.half().float(); l = (y - h).half().float(); y = h + l
    return y.double()
snake_ref.F.conv1d = conv1d

for tag in ("single_box_v196", "batched_box_v128", "smooth_batched_box_v196"):
    g, cfg, sd, feats, image_idx, sizes = dance_case_inputs(tag)
    sdd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    args = (feats.double(), g["init"].double(), image_idx, sizes, g["whs"].double())
    kw = dict(attention=cfg.attention, individual_scale=cfg.individual_scale, de_loc_type=cfg.de_loc_type, num_iter=1)
    MODE["m"] = "fp64"; t64 = snake_ref.dance_deform(sdd, *args, **kw)[0]
    res = {}
    for m in ("bf16x3", "fp16x3", "fp16x3s"):
        MODE["m"] = m
        res[m] = snake_ref.dance_deform(sdd, *args, **kw)[0]
    ref32 = g["pred0"].double()
    sc = t64.abs().max()
    print(tag, "coord scale %.0f | abs err px vs fp64: ref-fp32 %.5f  bf16x3 %.5f  fp16x3 %.5f  fp16x3+split-storage %.5f" % (
        sc, (ref32 - t64).abs().max(), (res["bf16x3"] - t64).abs().max(), (res["fp16x3"] - t64).abs().max(),
        (res["fp16x3s"] - t64).abs().max()))
