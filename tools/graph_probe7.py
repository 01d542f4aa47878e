"""the failing protocol of test_deformation_replays_from_a_hip_graph with the graph's layout outputs kept (diagnostic)"""
import sys
import torch
sys.path.insert(0, "tests")
from dance_amd import ops, snake, synthetic
from _util import dance_state_dict

device = torch.device("cuda:0")
cfg, sd, _ = dance_state_dict(1234, 128)
nets = [snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer%d." % i), device) for i in range(3)]
feats = synthetic.make_smooth_features(2, 256, 50, 76, 21).to(device)
boxes = synthetic.make_boxes(24, 200, 304, 22, r_min=8, r_max=60)
loc = synthetic.make_circle_contours(boxes, 128).to(device)
whs = torch.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1).to(device)
idx = (torch.arange(24) % 2).float().to(device)
print("ptrs: feats %x loc %x whs %x idx %x" % (feats.data_ptr(), loc.data_ptr(), whs.data_ptr(), idx.data_ptr()))
orig = ops.nchw_to_nhwc
keep = []


def spy(f, a):
    r = orig(f, a)
    keep.append((getattr(r[0], ops.ABSMAX_ATTR), r[0], r[1]))
    return r


ops.nchw_to_nhwc = spy
d = snake.ContourDeformer(nets)


def step():
    return d.deform(feats, loc, idx, [(200, 304)] * 2, whs)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        eager = step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
del keep[:]
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
gw, gf, ga = keep[0]
del keep[:]
g.replay()
torch.cuda.synchronize()
print("before change: equal", [torch.equal(a, b) for a, b in zip(out, eager)])
def show(tag):
    torch.cuda.synchronize()
    print(tag, [hex(x & 0xffffffff) for x in gw.flatten().tolist()])


show("graph words after first replay")
loc.add_(1.5)
show("after loc.add_")
want = [t.clone() for t in step()]
show("after eager step")
ew, ef, ea = keep[-1]
g.replay()
torch.cuda.synchronize()
show("after replay")
gw.zero_()
g.replay()
show("after zero + replay")
print("after change: equal", [torch.equal(a, b) for a, b in zip(out, want)])
print("feat equal", torch.equal(gf, ef), "att equal", torch.equal(ga, ea), "words equal", torch.equal(gw, ew))
print("graph words", gw.view(torch.float32).tolist())
print("eager words", ew.view(torch.float32).tolist())
print("ptrs graph: words %x feat %x att %x out0 %x" % (gw.data_ptr(), gf.data_ptr(), ga.data_ptr(), out[0].data_ptr()))
print("ptrs eager: words %x feat %x att %x want0 %x ws %x" % (ew.data_ptr(), ef.data_ptr(), ea.data_ptr(), want[0].data_ptr(), d._ws.data_ptr()))
