import torch, sys
sys.path.insert(0, '.')
from dance_amd import ops, synthetic
from oracle import snake_ref, extreme_utils_ref as cref
boxes = synthetic.make_boxes(5, 800, 1216, seed=5)
print(boxes[:2])
dev = torch.device('cuda')
for kind, poly, p in ((ops.POLY_BOX, snake_ref.box_polygon(boxes), 16), (ops.POLY_QUAD, snake_ref.quadrangle(boxes), 16)):
    want, ws, _ = cref.uniform_upsample_fused(poly, p)
    got, gs = ops.uniform_upsample(boxes.to(dev), p, kind=kind)
    print(kind, 'want', want[0], ws[0]); print('got', got[0].cpu(), gs[0].cpu())
    got2, gs2 = ops.uniform_upsample(poly.to(dev), p)
    print('raw', torch.equal(got2.cpu(), want))
