import torch, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from _util import golden
from test_oracle_golden import dance_case_inputs, DANCE_CASES
from dance_amd import snake
from oracle import snake_ref
dev = torch.device('cuda')
for tag in DANCE_CASES:
    g, cfg, sd, feats, image_idx, sizes = dance_case_inputs(tag)
    nets = [snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer%d." % i), dev) for i in range(3)]
    d = snake.ContourDeformer(nets, padding="reflection", de_loc_type=cfg.de_loc_type, individual_scale=cfg.individual_scale, attention=cfg.attention)
    idx = None if image_idx is None else image_idx.to(dev)
    preds = d.deform(feats.to(dev), g["init"].to(dev), idx, sizes, g["whs"].to(dev))
    # fp64 truth for iteration 0
    sdd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    p64 = snake_ref.dance_deform(sdd, feats.double(), g["init"].double(), image_idx, sizes, g["whs"].double(), attention=cfg.attention, individual_scale=cfg.individual_scale, de_loc_type=cfg.de_loc_type, num_iter=1)[0]
    a = torch.nan_to_num(preds[0].cpu().double()); r = torch.nan_to_num(g["pred0"].double()); t = torch.nan_to_num(p64)
    print(tag, "iter0 abs err px: gpu-vs-ref %.5f  gpu-vs-fp64 %.5f  ref-vs-fp64 %.5f | coord max %.1f" % ((a-r).abs().max(), (a-t).abs().max(), (r-t).abs().max(), r.abs().max()))
    for i in (1, 2):
        a = torch.nan_to_num(preds[i].cpu().double()); r = torch.nan_to_num(g["pred%d" % i].double())
        print("   chain iter%d abs err px gpu-vs-ref %.5f" % (i, (a-r).abs().max()))
